"""CPU checks of the measurement tooling: the sustained-rate statistic bench.py quotes as `roofline.frac`, the contract
fields of its workloads table, and tools/summarize_profiles.py's mapping of a grid-stride kernel's launch back to the
rows it processed (`traffic.json` was keyed by grid x 64 and made bench_cfg5.json print 2.00 x the
algorithmic bytes for a kernel that moves exactly the algorithmic bytes)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from tests.conftest import REPO


def test_sustained_stats_is_the_mean_of_the_last_second():
    sys.path.insert(0, REPO)
    import bench

    # 100 launches per event; 0.4 ms per launch for the first 1.2 s (30 chunks of 40 ms), then 0.5 ms (25 chunks of 50 ms)
    per = [0.4] * 30 + [0.5] * 25
    st = bench.sustained_stats(per, 100)
    assert abs(st["seconds"] - 2.45) < 1e-9 and st["launches"] == 5500
    assert st["us_per_launch_last_second"] == 500.0 and st["us_per_launch_first_second"] == 400.0
    assert st["us_per_launch_min_chunk"] == 400.0 and st["us_per_launch_max_chunk"] == 500.0
    # a run shorter than a second is averaged whole
    assert bench.sustained_stats([0.3, 0.5], 10)["us_per_launch_last_second"] == 400.0


def test_algorithmic_bytes_follow_the_survey():
    import bench

    assert bench.algorithmic_bytes(6, 8, "osc") == 192 and bench.algorithmic_bytes(2, 8, "osc") == 96  # SURVEY 8d Mode U
    assert bench.algorithmic_bytes(3, 4, "sliding") == 48
    assert bench.algorithmic_bytes(6, 8, "osc_full") == 840                                            # Mode F
    assert bench.algorithmic_bytes(6, 8, "osc_full", ("Tx", "J", "M", "g", "C")) == 840 + 288
    assert bench.algorithmic_bytes(6, 8, "osc_full", ("Tx", "J", "M", "g", "C", "dJ")) == 840 + 576
    assert bench.algorithmic_bytes(6, 8, "dyn", ("Tx", "J", "M", "g")) == 696


def test_summarize_profiles_keys_grid_stride_kernels_by_rows(tmp_path):
    src, dst = tmp_path / "run", tmp_path / "out"
    (src / "pmc_FETCH_SIZE").mkdir(parents=True)
    (src / "pmc_WRITE_SIZE").mkdir()
    kn = "void abrk::sliding_kernel<abrk::StaticArm<abrk::Tab_threejoint>, float>(abrk::StaticArm<abrk::Tab_threejoint>, ...)"
    rows, grid = 8388608, 256 * 32 * 4 * 64  # the launcher caps the grid at kSlidingMaxBlocks: 2 097 152 threads
    hdr = "Kernel_Name,Grid_Size,Counter_Name,Counter_Value,VGPR_Count,Accum_VGPR_Count,Scratch_Size,LDS_Block_Size\n"
    # 36 B read + 12 B written per row = algorithmic (FETCH_SIZE counts 64 B per 128-B request on gfx950, in KiB)
    (src / "pmc_FETCH_SIZE" / "bench_counter_collection.csv").write_text(
        hdr + f'"{kn}",{grid},FETCH_SIZE,{rows * 36 / 2 / 1024},68,0,0,1024\n')
    (src / "pmc_WRITE_SIZE" / "bench_counter_collection.csv").write_text(
        hdr + f'"{kn}",{grid},WRITE_SIZE,{rows * 12 / 1024},68,0,0,1024\n')
    leg = {"kernel": "sliding_kernel<abrk::StaticArm<abrk::Tab_threejoint>, float>", "grid_threads": grid, "batch": rows,
           "bytes_per_eval": 48, "us_per_launch": 100.0, "frac": 0.5}
    (src / "pmc_FETCH_SIZE.log").write_text("noise\n" + json.dumps({"roofline": leg}) + "\n")
    p = subprocess.run([sys.executable, os.path.join(REPO, "tools", "summarize_profiles.py"), str(src), str(dst)],
                       capture_output=True, text=True, env=dict(os.environ, ABRK_PROFILE_COMMIT="test"))
    assert p.returncode == 0, p.stderr[-2000:]
    t = json.load(open(dst / "traffic.json"))
    key = f"sliding_kernel<abrk::StaticArm<abrk::Tab_threejoint>,float>:{rows}"
    assert key in t and t[key]["grid_threads"] == grid
    assert np.isclose(t[key]["read_bytes"] + t[key]["write_bytes"], 48 * rows)
    # what bench.py then prints for that leg: traffic / algorithmic = 1.00
    sys.path.insert(0, REPO)
    import bench

    old = bench.PROFILE_DIRS
    try:
        os.makedirs(os.path.join(REPO, "profiles", "_test_tmp"), exist_ok=True)
        json.dump(t, open(os.path.join(REPO, "profiles", "_test_tmp", "traffic.json"), "w"))
        bench.PROFILE_DIRS = ("_test_tmp",)
        traffic, _, commit = bench.profiled_traffic(leg["kernel"], rows)
        # (the stamp: the profile's commit + whether its kernel sources are this tree's - round 6)
        assert commit.startswith("test") and "identical to this tree's" in commit and np.isclose(traffic, 48 * rows)
        t["_sources"] = "0" * 16
        json.dump(t, open(os.path.join(REPO, "profiles", "_test_tmp", "traffic.json"), "w"))
        assert "DIFFERENT from this tree's" in bench.profiled_traffic(leg["kernel"], rows)[2]
    finally:
        bench.PROFILE_DIRS = old
        import shutil

        shutil.rmtree(os.path.join(REPO, "profiles", "_test_tmp"), ignore_errors=True)


def test_timing_plots_replica_settings_have_a_staged_reference_workload():
    """tests/timing_plots_replica.py times the reference's published benchmark on both sides: every setting names a
    workload oracle/time_reference.py knows (so that oracle/stage_reference.py staged its functions), for an arm this
    package ships, with the keyword arguments of examples/timing_plots.py:34-39"""
    import importlib.util

    from oracle import time_reference

    spec = importlib.util.spec_from_file_location("tp_replica", os.path.join(REPO, "tests", "timing_plots_replica.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert [s[0] for s in mod.SETTINGS] == ["Two joint", "UR5", "Jaco2"]
    for label, arm, kw, wl in mod.SETTINGS:
        ref_arm, factory, nt = time_reference.WORKLOADS[wl]
        assert ref_arm == arm and nt == 6
        assert os.path.isdir(os.path.join(REPO, "abr_control_amd", "arms", arm))
        want = "OSC(rc)" if not kw else "OSC(rc, ctrlr_dof=" + ("[True] * 6" if all(kw["ctrlr_dof"]) else "[True] * 5 + [False]") + ")"
        assert factory == want


def _bench(*argv, timeout=120):
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    return subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *argv], capture_output=True, text=True,
                          timeout=timeout, env=env, cwd=REPO)


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without a launcher around it starts N ranks of itself, they meet through the HostGroup
    (barrier, max, gather), and exactly ONE JSON line comes out of rank 0 (--dry-run: no device is touched)"""
    p = _bench("--gpus", "4", "--steps", "20", "--warmup", "5", "--dry-run")
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    d = json.loads(lines[0])
    assert d["dry_run"] and d["n_gpus"] == 4 and d["n_ranks_seen"] == 4 and d["steps"] == 20 and d["warmup"] == 5
    assert [r["rank"] for r in d["ranks"]] == [0, 1, 2, 3] == [r["local_rank"] for r in d["ranks"]]
    assert len({r["pid"] for r in d["ranks"]}) == 4 and d["wall_max"] == 4e-3  # MAX over the ranks


def test_bench_self_launch_fails_when_a_rank_fails():
    """a rank that dies takes the run down: non-zero exit code, no contract line"""
    p = _bench("--gpus", "3", "--dry-run", "--dry-run-fail-rank", "2")
    assert p.returncode != 0
    assert not [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert "the run is void" in p.stderr


def test_bench_self_launch_refuses_to_double_up_on_devices():
    """fewer devices than ranks: ONE JSON error line with devices_seen, rc 1, nothing started (this container has no
    device at all; on a one-GPU box the same line says devices_seen = 1)"""
    import abr_control_amd as a

    if a.device_count() >= 2:
        pytest.skip("needs a machine with fewer than 2 HIP devices")
    p = _bench("--gpus", "2", "--steps", "20", "--warmup", "5")
    assert p.returncode == 1
    d = json.loads(p.stdout.strip())
    assert d["devices_seen"] == a.device_count() and d["n_gpus"] == 2 and "error" in d
