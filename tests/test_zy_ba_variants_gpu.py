"""GPU: the alternative kernel paths of the BA solver are selected by problem shape (or, for A/B measurements, by
environment variables read once per process).  The parity suite of test_ba_gpu.py runs again in child processes with
each path forced, so that small test scenes also go through the point-major Schur pass and the fallbacks.

The file sorts after every default-path parity file (test_zy_...): with `pytest -x` a red non-default switch must not hide the
default-path tests of the other subsystems (round 3: it hid 169 of 192).  A failing child's report is forwarded whole: `-rf
--tb=short`, last 6000 characters, so the record names the inner test, the scene and the assertion."""
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
    {"SNK_BA_CHECK_LISTS": "1", "SNK_BA_NO_SCHUR_SET": "1"},            # every scene hand-over compares the device-built lists (camera records, block entries) with the host builder's
    {"SNK_BA_HOST_ENTRIES": "1", "SNK_BA_NO_SCHUR_SET": "1"},           # block entries by the host builder (what scenes with > 512 free cameras use)
    {"SNK_BA_SCHUR_SET_MIN_ITEMS": "1", "SNK_BA_CAM_SUMS": "1"},       # schur_fused<3, true>'s per-item camera sums + cam_sum instead of cam_pass (measured slower: not the default)
    {"SNK_BA_PCG_LDS": "1"},                                            # S in LDS (pcg_solve<true>) instead of registers (pcg_small) for local-BA sized systems
    {"SNK_BA_NO_BIG_ITEMS": "1"},                                       # batches of >= 256 problems keep work items of <= 64 points
    {"SNK_BA_LOCAL_SYNC": "1"},                                         # snk_ba_solve_local_scene decides about the extra iteration on the host (count read back)
    {"SNK_BA_BECNT_BUDGET": "1", "SNK_BA_CHECK_LISTS": "1"},            # the counter-memory budget of the device-built block entries exceeded: host builder takes over
    {"SNK_BA_HOST_THREADS": "4", "SNK_BA_CHECK_LISTS": "1", "SNK_BA_SCHUR_SET_MIN_ITEMS": "1"},  # the threaded list builder of a batch hand-over forced on every batch of >= 2 scenes, its lists checked
    {"SNK_BA_HOST_THREADS": "4", "SNK_BA_NO_HOST_POOL": "1", "SNK_BA_CHECK_LISTS": "1"},  # ... with threads created and joined per pass (rounds 4-6) instead of the handle's parked pool
    {"SNK_BA_HOST_THREADS": "3", "SNK_BA_NO_SCHUR_SET": "1"},
    {"SNK_BA_NO_SCHUR_WIDE": "1", "SNK_BA_NO_SCHUR_SET": "1"},          # block-major schur_pass with one wavefront per block also for single windows
    {"SNK_BA_PCGL_LAUNCHES": "1"},                                      # global scenes: the multi-launch PCG (pcgl_matvec / combine / update / direction / latch) instead of the one cooperative launch (pcgl_persist)
    {"SNK_BA_PERSIST_STREAM": "1"},                                     # global scenes: the one-barrier PCG streaming its rows of S in every iteration (pcgl_persist1) instead of holding them in registers (pcgl_persist_reg)
    {"SNK_BA_PERSIST_REG_ROWS": "16"},                                  # global scenes: 16 rows of S per workgroup in pcgl_persist_reg also below 512 unknowns (default there: 8)
    {"SNK_BA_PERSIST_TWO_BARRIERS": "1"},                               # global scenes: the two-barrier persistent PCG of round 5 (pcgl_persist) instead of the one-barrier form (pcgl_persist1)
    {"SNK_BA_FLAT_BARRIER": "1"},                                       # global scenes: the flat grid barrier of round 5 (every workgroup polls every flag) instead of the two-level one
    {"SNK_BA_GRAPH_CHAINS": "2", "SNK_BA_GRAPH_FIRST": "1"},            # batches of >= 128 windows recorded as two graph branches over disjoint window ranges (measured +1 % with 2, -6 % with 4: not the default)
    {"SNK_BA_PERSIST_FAIL": "1"},                                       # global scenes: the runtime refuses the cooperative launch -> the handle falls back to the multi-launch PCG inside the same solve (round-5 advisor)
])
def test_ba_parity_suite_with_forced_path(env):
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_ba_gpu.py"), "-m", "gpu", "-x", "-q", "-rf", "--tb=short",
                        "-p", "no:cacheprovider",
                        # 14 s of scipy on the CPU per run and independent of the path switches that only change launch shapes: default path only
                        "-k", "not matches_scipy"], env=dict(os.environ, **env), capture_output=True, text=True, cwd=str(ROOT), timeout=600)
    assert r.returncode == 0, f"{env}\n--- child stdout (tail) ---\n{r.stdout[-6000:]}\n--- child stderr (tail) ---\n{r.stderr[-1500:]}"
    assert " passed" in r.stdout


@pytest.mark.parametrize("pair", [
    (("device", {"SNK_BA_CHECK_LISTS": "1"}), ("host", {"SNK_BA_HOST_ENTRIES": "1"})),
    (("registers", {}), ("lds", {"SNK_BA_PCG_LDS": "1"})),
], ids=["block-entries-device-vs-host", "pcg-registers-vs-lds"])
def test_equivalent_paths_give_bit_identical_solutions(pair):
    """Pairs of paths that run the same arithmetic in the same order, in two child processes, must agree bit for bit:
    * the camera records and camera-pair block entries are built by kernels (gather_cam_records, block_entries_*) when a scene has
      <= 512 free cameras and no camera twice on a point.  With SNK_BA_CHECK_LISTS=1 snk_ba_set_problems compares them with the
      host builder element by element (and fails on a difference); SNK_BA_HOST_ENTRIES=1 uses the host builder's lists;
    * pcg_small keeps its quarter of S in registers, pcg_solve<true> reads it from LDS -- same summation order."""
    code = r"""
import numpy as np, os, sys
from snake_slam_amd import synth
from snake_slam_amd.ba import BARec, lba_options
rng = np.random.default_rng(5)
scenes = []
for seed, (kf, npt, opp) in enumerate([(20, 2000, 8), (3, 10, 2), (8, 300, 5), (64, 500, 12), (66, 400, 6), (5, 1, 5), (12, 700, 12), (131, 900, 9),
                                       (520, 1600, 7)]):  # 130 free cameras: three mask words on the device; 519: the host builder
    sc, _ = synth.ba_scene(n_kf=kf, n_pt=npt, obs_per_pt=opp, seed=100 + seed, outlier_frac=0.02, n_fixed=1 + seed % 3)
    sc["pt_const"][: npt // 7] = 1
    sc["obs_img"] = sc["obs_img"].copy(); sc["obs_pt"] = sc["obs_pt"].copy()
    if len(sc["obs_img"]) > 20:
        sc["obs_img"][3] = -1; sc["obs_pt"][7] = 10**6
    scenes.append(sc)
dup, _ = synth.ba_scene(n_kf=6, n_pt=80, obs_per_pt=4, seed=77)    # one camera twice on a point: host builder
dup["obs_img"] = dup["obs_img"].copy(); dup["obs_img"][1] = dup["obs_img"][0]
allc, _ = synth.ba_scene(n_kf=4, n_pt=50, obs_per_pt=3, seed=78)   # every camera constant: no blocks at all
allc["img_const"][:] = 1
empty = dict(scenes[1], obs_img=scenes[1]["obs_img"][:0], obs_pt=scenes[1]["obs_pt"][:0], obs_uv=scenes[1]["obs_uv"][:0],
             obs_depth=scenes[1]["obs_depth"][:0], obs_weight=scenes[1]["obs_weight"][:0])
out = []
ba = BARec(lba_options())
for group in [[s] for s in scenes] + [[dup], [allc], [empty], scenes[:3] + [allc, empty] + scenes[3:4], scenes[:2] + [dup]]:
    ba.create(group)
    ba.initAndSolve()
    for i in range(len(group)):
        pose, pt, _ = ba.state(i)
        out.append(pose); out.append(pt)
ba.close()
np.savez(sys.argv[1], *out)
print("ok")
"""
    import numpy as np
    import tempfile

    with tempfile.TemporaryDirectory() as d:
        res = {}
        for name, env in pair:
            f = os.path.join(d, name + ".npz")
            r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, PYTHONPATH=str(ROOT), **env), capture_output=True, text=True,
                               cwd=str(ROOT), timeout=600)
            assert r.returncode == 0 and "ok" in r.stdout, (name, r.stdout[-1500:], r.stderr[-1500:])
            res[name] = np.load(f)
        ra, rb = res[pair[0][0]], res[pair[1][0]]
        assert len(ra.files) == len(rb.files) > 0
        for k in ra.files:
            assert np.array_equal(ra[k], rb[k]), k
