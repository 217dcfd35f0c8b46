"""Compiled user arms (abr_control_amd/specialize.py, csrc/abrk_plugin.h, include/abrk.h abrk_arm_create_compiled):
the counterpart of the reference's generate-and-cache step (base_config.py:125-191).  The CPU half checks the cache keys,
the generated source and every refusal of the loader; the GPU half (test_gpu_parity.py) checks the kernels."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from abr_control_amd import _abi, specialize
from abr_control_amd._lib import check, lib
from tests import compiled_arms


@pytest.fixture(scope="module")
def L():
    return lib()


def test_plugins_of_the_test_arms_are_built():
    """__graft_entry__.build() leaves them in the in-tree cache, keyed by the CURRENT kernel headers"""
    assert specialize.plugin_abi() == specialize.plugin_abi(from_sources=True), "libabrk.so is older than its headers"
    for name, tab in compiled_arms.test_arms().items():
        p = specialize.find_compiled(tab)
        assert p and p.startswith(specialize.IN_TREE), f"{name}: no plugin for the current headers - run build()"
        src = open(os.path.join(os.path.dirname(p), "arm.hip")).read()
        assert src == specialize.plugin_source(tab, specialize.arm_key(tab))


def test_key_depends_on_values_not_on_the_name():
    tab = _abi.load_table("ur5")
    k = specialize.arm_key(tab)
    t2 = dict(tab, name="something_else")
    assert specialize.arm_key(t2) == k
    t3 = dict(tab)
    t3["mdiag"] = [list(r) for r in tab["mdiag"]]
    t3["mdiag"][3][0] = np.nextafter(t3["mdiag"][3][0], 10.0)  # one ulp of one mass
    assert specialize.arm_key(t3) != k
    assert specialize.arm_key(tab, abi="other-headers") != k
    assert len(k) == 16 and int(k, 16) >= 0


def test_generated_table_reads_back_exactly():
    """the literals of the generated struct are the table's doubles (repr round-trips): parse them back"""
    import re

    tab = compiled_arms.test_arms()["synthetic4"]
    src = _abi.render_tab_struct(tab, "Tab_x")
    m = re.search(r"AJ\[4\]\[12\] = \{(.*?)\};", src, re.S)
    vals = [float(v) for v in re.findall(r"[-+0-9.e]+(?:inf|nan)?", m.group(1).replace("{", " ").replace("}", " "))]
    assert np.array_equal(np.array(vals).reshape(4, 3, 4), np.array(tab["AJ"], dtype=float))
    assert "static constexpr int N = 4;" in src and "kHasEE = true" in src


def test_compiled_arm_registers_and_recycles_slots(L):
    tab = compiled_arms.test_arms()["synthetic4"]
    path = specialize.find_compiled(tab)
    d = _abi.desc_from_table(tab)
    ids = [check(L.abrk_arm_create_compiled(C.byref(d), path.encode())) for _ in range(3)]
    assert len(set(ids)) == 3 and min(ids) >= 5
    back = _abi.ArmDesc()
    assert L.abrk_arm_get_desc(ids[0], C.byref(back)) == 0 and back.n_joints == 4
    for i in ids:
        assert L.abrk_arm_destroy(i) == 0
    again = check(L.abrk_arm_create_compiled(C.byref(d), path.encode()))
    assert again not in ids and (again & 4095) in {i & 4095 for i in ids}  # the slot is reused under a new generation
    assert L.abrk_arm_destroy(again) == 0


def test_loader_refuses_what_it_should(L, tmp_path):
    arms = compiled_arms.test_arms()
    tab = arms["synthetic4"]
    path = specialize.find_compiled(tab).encode()
    d = _abi.desc_from_table(tab)
    # a different table (one ulp in one frame offset) than the plugin was compiled for
    d2 = _abi.desc_from_table(tab)
    d2.B[1][3] = np.nextafter(d2.B[1][3], 10.0)
    assert L.abrk_arm_create_compiled(C.byref(d2), path) == -1 and b"different arm table" in L.abrk_last_error()
    # another arm's plugin
    other = specialize.find_compiled(arms["threejoint_user"]).encode()
    assert L.abrk_arm_create_compiled(C.byref(d), other) == -1
    # not a file / not a plugin / NULL
    assert L.abrk_arm_create_compiled(C.byref(d), b"/nonexistent/arm.so") == -1 and b"cannot load" in L.abrk_last_error()
    assert L.abrk_arm_create_compiled(C.byref(d), lib()._name.encode()) == -1 and b"not an arm plugin" in L.abrk_last_error()
    assert L.abrk_arm_create_compiled(None, path) == -1 and L.abrk_arm_create_compiled(C.byref(d), None) == -1
    d3 = _abi.desc_from_table(tab)
    d3.n_joints = 9
    assert L.abrk_arm_create_compiled(C.byref(d3), path) == -1
    # a plugin built from other kernel headers: same entry points, another tag (a stand-in compiled with gcc)
    src = tmp_path / "stale.c"
    src.write_text('const void* abrk_plugin_ops(void) { return 0; }\n'
                   'const char* abrk_plugin_abi_tag(void) { return "0000000000000000-00000000"; }\n'
                   'void abrk_plugin_desc(void* d) { (void)d; }\n')
    so = tmp_path / "stale.so"
    subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)], check=True)
    assert L.abrk_arm_create_compiled(C.byref(d), str(so).encode()) == -1
    assert b"rebuild it" in L.abrk_last_error() and specialize.plugin_abi().encode() in L.abrk_last_error()


def test_config_picks_up_a_cached_plugin_and_can_be_told_not_to():
    from abr_control_amd import arms

    tab = compiled_arms.test_arms()["synthetic4"]
    rc = arms.from_table(tab)
    assert rc.arm_id >= 5 and rc.plugin_path == specialize.find_compiled(tab)
    rc2 = arms.from_table(tab, compiled=False)
    assert rc2.arm_id >= 5 and rc2.plugin_path is None
    # an arm nobody compiled runs the runtime-table kernels; compiled=True would build (not exercised here: minutes)
    t2 = dict(tab)
    t2["mdiag"] = [list(r) for r in tab["mdiag"]]
    t2["mdiag"][1][0] += 0.125
    rc3 = arms.from_table(t2)
    assert rc3.arm_id >= 5 and rc3.plugin_path is None
    for r in (rc, rc2, rc3):
        r.close()


def test_compile_arm_fails_loudly_without_a_compiler(monkeypatch, tmp_path):
    tab = dict(compiled_arms.test_arms()["synthetic4"])
    tab["mdiag"] = [list(r) for r in tab["mdiag"]]
    tab["mdiag"][2][1] += 0.5  # not cached anywhere
    monkeypatch.setenv("HIPCC", "/nonexistent/hipcc")
    with pytest.raises(RuntimeError, match="hipcc not found"):
        specialize.compile_arm(tab, cache_dir=str(tmp_path))
    assert not any(f.endswith(".so") for _, _, fs in os.walk(tmp_path) for f in fs)


def test_prune_drops_only_plugins_of_other_headers(tmp_path):
    cur, old, other = tmp_path / "aaaa", tmp_path / "bbbb", tmp_path / "not_a_plugin"
    for d, tag in ((cur, specialize.plugin_abi()), (old, "0123456789abcdef-00000000")):
        d.mkdir()
        (d / "arm.hip").write_text("// x\n")
        (d / "arm.so").write_bytes(b"")
        (d / "abi.txt").write_text(tag + "\n")
    other.mkdir()
    (other / "keep.txt").write_text("x")
    gone = specialize.prune(str(tmp_path))
    assert gone == [str(old)] and cur.exists() and other.exists() and not old.exists()
    assert specialize.prune(str(tmp_path / "missing")) == []
