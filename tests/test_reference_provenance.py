"""Provenance of the committed reference-derived files, re-checked against the reference itself.

Build container only (needs /root/reference, which does not exist on the GPU box -> skipped there):
  * tools/extract_arm_table.py run on the reference's arm configs must reproduce
    abr_control_amd/arms/tables/<arm>.json byte for byte (arms/*/config.py `_calc_T`);
  * oracle/gen_golden.py run on one arm must reproduce the committed tests/golden/<arm>.npz bit for bit
    (the fixtures every parity test is pinned to really are outputs of the reference).
The arm regenerated is `onejoint` (the cheapest: ~1 min including the reference's SymPy code generation when its
function cache is cold); set ABRK_PROVENANCE_ARMS="twojoint,ur5" to regenerate others (minutes each)."""
import os
import shutil
import subprocess
import sys
import warnings

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
SCRATCH = "/tmp/abrk_ref_scratch_prov"

pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "abr_control")),
                                reason="needs the reference checkout (build container only)")


@pytest.fixture(scope="module")
def ref_env():
    # never import the reference from where it lies: imports write __pycache__ / pyximport artefacts (hazard 1)
    if os.path.isdir(SCRATCH):
        shutil.rmtree(SCRATCH)
    shutil.copytree(REF, SCRATCH)
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1", PYTHONPATH=SCRATCH)
    yield env
    shutil.rmtree(SCRATCH, ignore_errors=True)


@pytest.mark.parametrize("arm", ["onejoint", "twojoint", "threejoint", "ur5", "jaco2"])
def test_extractor_reproduces_committed_table(arm, ref_env, tmp_path):
    out = tmp_path / f"{arm}.json"
    subprocess.run([sys.executable, os.path.join(REPO, "tools", "extract_arm_table.py"), arm, str(out)], env=ref_env,
                   check=True, capture_output=True)
    committed = os.path.join(REPO, "abr_control_amd", "arms", "tables", f"{arm}.json")
    assert out.read_bytes() == open(committed, "rb").read(), f"{arm}: extractor output differs from the committed table"


def test_extractor_keeps_non_literal_constants():
    """a static rotation that is not axis aligned (cos(pi/4)) is kept as evaluated instead of being refused"""
    sys.path.insert(0, os.path.join(REPO, "tools"))
    try:
        import extract_arm_table as x
    finally:
        sys.path.pop(0)
    assert x._snap(0.25 + 3e-17) == 0.25
    assert x._snap(float(np.float32(0.3))) == float(np.float32(0.3))
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        v = float(np.cos(np.pi / 4)) + 1e-16
        assert x._snap(v) == v
    assert len(w) == 1


def test_gen_golden_reproduces_committed_fixture(tmp_path):
    arms = [a for a in os.environ.get("ABRK_PROVENANCE_ARMS", "onejoint").split(",") if a]
    out = tmp_path / "golden"
    subprocess.run([sys.executable, os.path.join(REPO, "oracle", "gen_golden.py"), "--out", str(out), *arms], check=True,
                   capture_output=True, timeout=3000)
    for arm in arms:
        new, old = np.load(out / f"{arm}.npz"), np.load(os.path.join(REPO, "tests", "golden", f"{arm}.npz"))
        assert sorted(new.files) == sorted(old.files)
        for k in old.files:
            a, b = new[k], old[k]
            assert a.shape == b.shape and a.dtype == b.dtype, k
            if a.dtype.kind == "f":
                # bit-identical when the reference's Cython functions are loaded from its cache in both runs; its
                # first-use fallback (lambdify, base_config.py:144) differs from them by rounding only
                assert np.array_equal(a, b) or np.max(np.abs(a - b)) <= 1e-12 * max(1.0, np.max(np.abs(b))), k
            else:
                assert np.array_equal(a, b), k
