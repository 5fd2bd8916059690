"""GPU: the alternative kernel paths of the BA solver are selected by problem shape (or, for A/B measurements, by
environment variables read once per process).  The parity suite of test_ba_gpu.py runs again in child processes with
each path forced, so that small test scenes also go through the point-major Schur pass and the fallbacks."""
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


@pytest.mark.parametrize("env", [
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1"},  # schur_fused (linearisation + matrix-core Schur products) + schur_sum for every scene, however small
    {"SNK_BA_NO_SCHUR_SET": "1"},         # block-major schur_pass everywhere
    {"SNK_BA_NO_POINT_WAVE": "1"},        # thread-per-point linearisation + schur_pass with activity lookups
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1", "SNK_BA_NO_GRAPH": "1"},
    {"SNK_BA_GRAPH_FIRST": "1"},
    {"SNK_BA_PCG_GENERAL": "1"},          # the 256-thread PCG loop instead of the replicated four-wavefront one
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1", "SNK_BA_FUSED_K10": "1"},       # schur_fused<4> also where points have 9-10 free observations
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1", "SNK_BA_NO_SCHUR_FUSED": "1"},  # point_wave + schur_mfma (W through HBM)
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1", "SNK_BA_NO_SCHUR_MFMA": "1"},   # point_wave + the vector-ALU schur_set          # explicitly built hipGraph already for the first solve of every scene
])
def test_ba_parity_suite_with_forced_path(env):
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_ba_gpu.py"), "-m", "gpu", "-x", "-q", "-p",
                        "no:cacheprovider"], env=dict(os.environ, **env), capture_output=True, text=True, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, (env, r.stdout[-2000:], r.stderr[-1000:])
    assert " passed" in r.stdout
