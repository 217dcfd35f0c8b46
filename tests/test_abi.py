"""Host-side checks that need no GPU: libabrk.so loads and exports every symbol
include/abrk.h declares, the arm registry and tables are consistent, the Python surface
mirrors the reference's, and compute calls FAIL LOUDLY without a device."""
import ctypes as C
import os
import re
import subprocess
import sys

import numpy as np
import pytest

from abr_control_amd import _abi
from tests.conftest import REPO


@pytest.fixture(scope="module")
def L():
    from abr_control_amd._lib import lib

    return lib()


def test_every_declared_symbol_is_exported(L):
    import glob

    hdr = "".join(open(f).read() for f in sorted(glob.glob(os.path.join(REPO, "include", "*.h"))))
    names = sorted(set(re.findall(r"\b(abrk_[a-z0-9_]+)\s*\(", hdr)))
    assert len(names) >= 45 and {"abrk_plan_begin", "abrk_osc_generate_full_batch", "abrk_osc_generate_sharded"} <= set(names)
    for nm in names:
        assert hasattr(L, nm), f"libabrk.so does not export {nm}"
    assert L.abrk_version() == 100


def test_struct_layouts_match_header():
    """ctypes mirrors vs the C structs (sizes computed by compiling a probe with the real header)"""
    src = r'''
#include <stdio.h>
#include "abrk.h"
int main(){printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(abrk_arm_desc), sizeof(abrk_dyn_out),
  sizeof(abrk_null_ctrl), sizeof(abrk_osc_params), sizeof(abrk_sliding_params), offsetof(abrk_osc_params, null_ctrl),
  sizeof(abrk_limits_params), sizeof(abrk_obstacles_params), offsetof(abrk_obstacles_params, obstacles),
  sizeof(abrk_scratch_info), offsetof(abrk_scratch_info, device_free_bytes), sizeof(abrk_shard_cut),
  offsetof(abrk_shard_cut, rows), offsetof(abrk_shard_cut, streams));return 0;}'''
    exe = "/tmp/abrk_layout_probe"
    subprocess.run(["gcc", "-x", "c", "-", "-I", os.path.join(REPO, "include"), "-o", exe], input=src.encode(),
                   check=True)
    sizes = [int(v) for v in subprocess.run([exe], capture_output=True, check=True).stdout.split()]
    assert sizes == [C.sizeof(_abi.ArmDesc), C.sizeof(_abi.DynOut), C.sizeof(_abi.NullCtrl),
                     C.sizeof(_abi.OSCParams), C.sizeof(_abi.SlidingParams), _abi.OSCParams.null_ctrl.offset,
                     C.sizeof(_abi.LimitsParams), C.sizeof(_abi.ObstaclesParams), _abi.ObstaclesParams.obstacles.offset,
                     C.sizeof(_abi.ScratchInfo), _abi.ScratchInfo.device_free_bytes.offset, C.sizeof(_abi.ShardCut),
                     _abi.ShardCut.rows.offset, _abi.ShardCut.streams.offset]


def test_resident_entry_points_validate_and_fail_loudly_without_a_device(L):
    """abrk_*_resident / abrk_shards_sync / abrk_plans_launch (round 6): argument checks that need no device, and
    ABRK_ENODEV - never a CPU path - where one is needed"""
    import abr_control_amd as a

    p = _abi.make_osc_params(6, kp=200)
    ur5 = L.abrk_arm_builtin(b"ur5")
    dev = (C.c_int32 * 2)(0, 0)
    rows = (C.c_int64 * 2)(4, 4)
    cut = _abi.ShardCut(2, dev, rows, None)
    tab = (C.c_void_p * 2)(None, None)
    assert L.abrk_osc_generate_resident(ur5, _abi.F64, C.byref(p), None, tab, tab, tab, None, None, None, tab, None) == -1
    bad = _abi.ShardCut(0, dev, rows, None)
    assert L.abrk_osc_generate_resident(ur5, _abi.F64, C.byref(p), C.byref(bad), tab, tab, tab, None, None, None, tab, None) < 0
    assert L.abrk_shards_sync(None) < 0
    assert L.abrk_plans_launch(None, 0, 1, 0) < 0
    ids = (C.c_int * 1)(12345)
    assert L.abrk_plans_launch(ids, 1, 1, 7) < 0 and L.abrk_plans_launch(ids, 1, 1, 0) < 0  # bad mode; unknown plan
    if a.device_count() == 0:
        rc = L.abrk_osc_generate_resident(ur5, _abi.F64, C.byref(p), C.byref(cut), tab, tab, tab, None, None, None, tab, None)
        assert rc == -2 and b"no HIP device" in L.abrk_last_error()  # ABRK_ENODEV
        assert L.abrk_shards_sync(C.byref(cut)) == -2
        assert not L.abrk_shard_stream(0, 0)


def test_sharded_array_cut_matches_shard_range():
    from abr_control_amd.sharding import ShardedArray, shard_range

    for B in (0, 1, 7, 8, 9, 4096, 10007):
        for G in (1, 2, 3, 8, 13):
            rows = ShardedArray.cut(B, G)
            assert sum(rows) == B and max(rows) - min(rows) <= 1
            assert rows == [shard_range(B, g, G)[1] - shard_range(B, g, G)[0] for g in range(G)]


def test_builtin_arm_registry_matches_tables(L):
    for name in _abi.BUILTIN_ARMS:
        aid = L.abrk_arm_builtin(name.encode())
        assert aid >= 0
        d = _abi.ArmDesc()
        assert L.abrk_arm_get_desc(aid, C.byref(d)) == 0
        tab, got = _abi.load_table(name), _abi.table_from_desc(d)
        assert got["n_joints"] == tab["n_joints"] and got["n_links_dyn"] == tab["n_links_dyn"]
        assert got["has_ee"] == tab["has_ee"]
        assert np.array_equal(np.array(got["A0"]), np.array(tab["A0"]))
        assert np.array_equal(np.array(got["AJ"]), np.array(tab["AJ"]))
        assert np.array_equal(np.array(got["B"]), np.array(tab["B"]))
        n = tab["n_joints"]
        assert np.array_equal(np.array(got["mdiag"]), np.array((tab["mdiag"] + [[0.0] * 6] * 8)[: n + 1]))
    assert L.abrk_arm_builtin(b"nonexistent") == -4
    assert b"nonexistent" in L.abrk_last_error()


def test_generated_header_in_sync_with_tables():
    rc = subprocess.run([sys.executable, os.path.join(REPO, "tools", "gen_builtin_arms.py"), "--check"]).returncode
    assert rc == 0, "abr_control_amd/csrc/abrk_arms_builtin.h is stale: run tools/gen_builtin_arms.py"


def test_user_arm_create_and_validation(L):
    tab = _abi.load_table("ur5")
    d = _abi.desc_from_table(tab)
    aid = L.abrk_arm_create(C.byref(d))
    assert aid >= 5
    back = _abi.ArmDesc()
    assert L.abrk_arm_get_desc(aid, C.byref(back)) == 0 and back.n_joints == 6
    assert L.abrk_arm_destroy(aid) == 0
    assert L.abrk_arm_get_desc(aid, C.byref(back)) == -4
    assert L.abrk_arm_destroy(0) == -4  # built-ins cannot be destroyed
    d.n_joints = 9
    assert L.abrk_arm_create(C.byref(d)) == -1


def test_user_arm_slots_are_reused(L):
    """a long-running process may register and drop any number of arms: destroyed slots are handed out again;
    only the number of arms alive at once is bounded"""
    d = _abi.desc_from_table(_abi.load_table("twojoint"))
    seen, ids = set(), set()
    for _ in range(6000):
        aid = L.abrk_arm_create(C.byref(d))
        assert aid >= 5
        seen.add(aid & 4095)  # an id is slot | generation << 12
        ids.add(aid)
        assert L.abrk_arm_destroy(aid) == 0
    assert len(seen) <= 4  # the same slot(s) over and over (other tests may hold a few)
    assert len(ids) == 6000  # ... but never the same id twice: a stale handle cannot reach the slot's next tenant
    back = _abi.ArmDesc()
    assert L.abrk_arm_get_desc(aid, C.byref(back)) == -4 and L.abrk_arm_destroy(aid) == -4
    live = L.abrk_arm_create(C.byref(d))
    assert live != aid and L.abrk_arm_get_desc(aid, C.byref(back)) == -4  # same slot, other generation
    assert L.abrk_arm_get_desc(live, C.byref(back)) == 0 and L.abrk_arm_destroy(live) == 0
    held = [L.abrk_arm_create(C.byref(d)) for _ in range(50)]
    assert len(set(held)) == 50 and min(held) >= 5
    for aid in held:
        assert L.abrk_arm_destroy(aid) == 0


def test_frame_ids_and_invalid_names():
    assert _abi.frame_id("link0", 6) == 0 and _abi.frame_id("joint0", 6) == 1
    assert _abi.frame_id("link6", 6) == 12 and _abi.frame_id("EE", 6) == 13
    for bad in ("link7", "joint6", "ee", "hand", "link-1", 3):
        with pytest.raises(Exception, match="Invalid transformation name"):
            _abi.frame_id(bad, 6)


def test_python_surface_mirrors_reference():
    from abr_control_amd.arms import jaco2, onejoint, threejoint, twojoint, ur5
    from abr_control_amd.controllers import OSC, Damping, Joint, RestingConfig, Sliding

    rc = ur5.Config(use_cython=True)
    assert (rc.N_JOINTS, rc.N_LINKS, rc.ROBOT_NAME) == (6, 7, "ur5")
    assert rc.L.shape == (13, 3) and len(rc._M_LINKS) == 7 and rc.START_ANGLES.dtype == np.float32
    assert jaco2.Config().L.shape == (14, 3) and jaco2.Config().N_JOINTS == 6
    assert twojoint.Config().L.shape == (6, 3) and threejoint.Config().L.shape == (8, 3)
    assert onejoint.Config().N_LINKS == 1
    for m in ("g", "dJ", "J", "M", "R", "quaternion", "C", "T", "Tx", "T_inv"):
        assert callable(getattr(rc, m))
    with pytest.raises(TypeError):
        ur5.Config(hand_attached=True)  # base_config.py:78 rejects unknown kwargs the same way
    c = OSC(rc, kp=200)
    assert c.ko == 200 and np.isclose(c.kv, np.sqrt(400)) and list(c.ctrlr_dof) == [1, 1, 1, 0, 0, 0]
    c = OSC(rc, kp=10, ko=8, kv=4, vmax=[1, 1], ki=0.1)
    assert np.isclose(c.sat_gain_xyz, 1 / 10 * 4) and np.isclose(c.sat_gain_abg, 1 / 8 * 4)
    assert c.integrated_error.shape == (6,)
    assert Sliding(rc).kd == 160.0 and Sliding(rc).lamb == 30.0
    assert np.isclose(Joint(rc, kp=16).kv, 4.0)
    r = RestingConfig(rc, [None, 1.0, None, None, None, None], kp=4)
    assert r.rest_indices == [False, True, False, False, False, False] and not r.account_for_gravity
    assert Damping(rc, 10).kv == 10

    # the remaining secondary controllers (avoid_joint_limits.py:35-81, floating.py:22-26, avoid_obstacles.py:26-36)
    from abr_control_amd.controllers import AvoidJointLimits, AvoidObstacles, Floating

    a = AvoidJointLimits(rc, [np.pi / 5, None, 5.5, np.nan, 0.1, 0.2], [np.pi / 2, 3.0, 0.8, np.nan, 3.0, 3.0],
                         cross_zero=[False, False, True, False, False, False])
    assert np.allclose(a.min_joint_angles[[0, 2]], [np.pi / 5 - np.pi, 0.8 - np.pi])  # shifted, swapped
    assert np.isclose(a.max_joint_angles[2], 5.5 - np.pi) and list(a.no_limits_min) == [0, 1, 0, 1, 0, 0]
    assert list(a.max_torque) == [1.0] * 6 and not a.gradient.any()
    with pytest.raises(Exception, match="incorrect size"):
        AvoidJointLimits(rc, [0.0] * 5, [1.0] * 5)
    f = Floating(rc)
    assert f.dynamic is False and f.task_space is False
    o = AvoidObstacles(rc)
    assert (o.threshold, o.gain, o.maximum) == (0.2, 1, 500) and o.obstacles.shape == (0,)
    o.set_obstacles([[0.1, 0.2, 0.3, 0.05]])
    assert o.obstacles.shape == (1, 4) and o._params().n_obstacles == 1
    with pytest.raises(ValueError):
        AvoidObstacles(rc, obstacles=[[0, 0, 0, 0.1]] * 17)._params()
    c = OSC(rc, kp=10, null_controllers=[a, Damping(rc, 10), o])
    assert len(c._device) == 2 and len(c._fused) == 1 and not c._foreign

    class Foreign:  # any duck-typed robot_config (e.g. the reference's MujocoConfig) is accepted by OSC:
        N_JOINTS = 6  # its J/M/g/Tx come from its own code, the law runs on the GPU (abrk_osc_law_batch)

    assert OSC(Foreign(), kp=10)._fused_config is False
    with pytest.raises(TypeError, match="no CPU fallback"):
        Sliding(Foreign())


def test_compute_fails_loudly_without_gpu(L):
    from abr_control_amd import AbrkError, device_count
    from abr_control_amd.arms import ur5
    from abr_control_amd.controllers import OSC

    if device_count() > 0:
        pytest.skip("a GPU is present")
    rc = ur5.Config()
    with pytest.raises(AbrkError, match="ENODEV"):
        rc.M(np.zeros(6))
    with pytest.raises(AbrkError, match="ENODEV"):
        OSC(rc, kp=200).generate(np.zeros(6), np.zeros(6), np.zeros(6))


def test_argument_validation_before_device(L):
    """bad arguments are rejected with EINVAL/EFRAME even without a device"""
    from abr_control_amd import AbrkError, engine

    p = _abi.make_osc_params(6, kp=1)
    q = np.zeros((2, 6))
    with pytest.raises(AbrkError, match="ENOARM"):
        engine.osc_generate(999, 6, p, q, q, q)
    p.ref_frame = 99
    with pytest.raises(AbrkError, match="EFRAME"):
        engine.osc_generate(0, 6, p, q, q, q)
    p = _abi.make_osc_params(6, kp=1, ctrlr_dof=[0] * 6)
    with pytest.raises(AbrkError, match="EINVAL"):
        engine.osc_generate(0, 6, p, q, q, q)
    with pytest.raises(ValueError):
        engine.osc_generate(0, 6, _abi.make_osc_params(6), q, q, np.zeros((2, 5)))
    with pytest.raises(TypeError):
        engine.osc_generate(0, 6, _abi.make_osc_params(6), q, q, q, dtype=np.float16)
    with pytest.raises(AbrkError, match="EINVAL"):
        engine.avoid_joint_limits_generate(9, _abi.make_limits_params(6, [0.0] * 6, [1.0] * 6), np.zeros((2, 9)))
    with pytest.raises(AbrkError, match="ENOARM"):
        engine.floating_generate(999, 6, False, False, q)
    with pytest.raises(AbrkError, match="EINVAL"):
        engine.avoid_obstacles_generate(0, 6, _abi.make_obstacles_params([[0, 0, 0, 1]], threshold=0), q)
    assert engine.floating_generate(0, 6, True, True, np.zeros((0, 6)), np.zeros((0, 6))).shape == (0, 6)
    # empty batch is a no-op even without a device
    u = engine.osc_generate(0, 6, _abi.make_osc_params(6), np.zeros((0, 6)), np.zeros((0, 6)), np.zeros((0, 6)))
    assert u.shape == (0, 6)


def test_config_copies_take_their_own_registry_handle(L):
    """copy / deepcopy / pickle of a user-arm config must not share its arm id: the first copy collected would hand
    the slot back under the survivor"""
    import copy
    import pickle

    from abr_control_amd.arms.base_config import BatchedConfig

    rc = BatchedConfig(_abi.load_table("threejoint"), compiled=False)
    first = rc.arm_id
    assert first >= 5
    back = _abi.ArmDesc()
    for dup in (copy.copy(rc), copy.deepcopy(rc), pickle.loads(pickle.dumps(rc))):
        assert dup._arm_id is None and dup.N_JOINTS == 3 and np.array_equal(dup.L, rc.L)
        other = dup.arm_id
        assert other != first and L.abrk_arm_get_desc(other, C.byref(back)) == 0
        dup.close()
        assert L.abrk_arm_get_desc(other, C.byref(back)) == -4  # the copy's handle is gone ...
        assert L.abrk_arm_get_desc(first, C.byref(back)) == 0 and back.n_joints == 3  # ... the original's is not
    rc.close()
    assert L.abrk_arm_get_desc(first, C.byref(back)) == -4


def test_error_codes_match_the_header_and_singular_is_a_linalg_error():
    """include/abrk_types.h's error enum against the Python table; ABRK_ESINGULAR surfaces as numpy.linalg.LinAlgError -
    what the reference's np.linalg.inv(M) raises (controllers/osc.py:136) - and still as an AbrkError"""
    from abr_control_amd._lib import AbrkError, SingularMatrixError

    txt = open(os.path.join(REPO, "include", "abrk_types.h")).read()
    codes = {m.group(1): int(m.group(2)) for m in re.finditer(r"ABRK_(E[A-Z]+)\s*=\s*(-\d+)", txt)}
    assert codes == {v: k for k, v in _abi.ERRORS.items()}
    assert codes["ESINGULAR"] == _abi.ESINGULAR == -6
    e = SingularMatrixError(_abi.ESINGULAR, "Singular matrix")
    assert isinstance(e, np.linalg.LinAlgError) and isinstance(e, AbrkError) and e.code == -6 and "ESINGULAR" in str(e)
