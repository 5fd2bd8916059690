"""The seeded generators of bench.py's inputs: the forked batch form returns exactly what the one-at-a-time functions do, and it
steps aside (serial loop) where forking is not safe -- under a profiler, or once the HIP library is loaded."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent


def test_batch_generators_equal_the_serial_functions():
    # in a child process: the test session itself may already have loaded libsnake_hip.so (then the pool is skipped)
    code = (
        "import numpy as np\n"
        "from snake_slam_amd import synth\n"
        "assert synth._fork_is_safe()\n"
        "fr = synth.stereo_frames([3, 1, 7, 2], 320, 240, n_rects=60, workers=3)\n"
        "for k, i in enumerate([3, 1, 7, 2]):\n"
        "    l, r = synth.stereo_frame(i, 320, 240, n_rects=60)\n"
        "    assert np.array_equal(l, fr[k][0]) and np.array_equal(r, fr[k][1])\n"
        "sc = synth.ba_scenes([11, 12, 13], workers=2, n_kf=5, n_pt=60, obs_per_pt=3)\n"
        "for k, sd in enumerate([11, 12, 13]):\n"
        "    w = synth.ba_scene(n_kf=5, n_pt=60, obs_per_pt=3, seed=sd)[0]\n"
        "    assert all(np.array_equal(np.asarray(w[key]), np.asarray(sc[k][key])) for key in w)\n"
        "assert synth.stereo_frames([], 64, 64) == [] and synth.ba_scenes([]) == []\n"
        "print('ok')\n"
    )
    env = {k: v for k, v in os.environ.items() if not k.startswith(("ROCP_", "ROCPROF"))}
    r = subprocess.run([sys.executable, "-c", code], cwd=str(ROOT), env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout, r.stderr[-2000:])


def test_no_fork_under_a_profiler(monkeypatch):
    from snake_slam_amd import synth

    monkeypatch.setenv("ROCP_TOOL_LIBRARIES", "librocprofiler-sdk-tool.so")
    assert not synth._fork_is_safe()
    a = synth.stereo_frames([5], 160, 120, n_rects=20, workers=4)  # serial path, same data
    l, r = synth.stereo_frame(5, 160, 120, n_rects=20)
    assert np.array_equal(a[0][0], l) and np.array_equal(a[0][1], r)


def test_textured_scene_has_a_realistic_stereo_yield():
    """Round 2's flat rectangles gave ~12 % stereo matches per keypoint (nearly every corner an occlusion corner); the textured
    scene (far wall + a few textured objects) must give >= 35 %, so that the accept paths of StereoMatching / filterMatches carry
    load in bench.py.  Measured with the oracle (CPU)."""
    from oracle import oracle as orc
    from snake_slam_amd import synth

    W, H = 752, 480
    p = orc.orb_params(1000, 1.2, 4, 20, 7)
    ls = (np.float32(1.2) ** np.arange(4)).astype(np.float32)
    rect = orc.rectification((1.0, 1.0, 0.0, 0.0))

    def yields(texture):
        l, r = synth.stereo_frame(1, W, H, texture=texture)
        kl, dl = orc.orb_detect(p, l, threads=4)
        kr, dr = orc.orb_detect(p, r, threads=4)
        rl, _ = orc.rectify(rect, kl)
        rr, _ = orc.rectify(rect, kr)
        perm, _, _, _ = orc.feature_grid(rl, (0.0, 0.0, float(W), float(H)))
        g, gd = np.zeros_like(rl), np.zeros_like(dl)
        g[perm], gd[perm] = rl, dl
        ns, _, _ = orc.stereo_match(g, gd, rr, dr, 47.9 * 2.5, ls, True)
        pairs = orc.bf_filter(orc.bf_knn2(gd, dr, threads=4), 60, 0.8)
        return ns / len(kl), len(pairs) / len(kl), len(kl)

    st, bf, n = yields(None)
    assert n >= 990 and st >= 0.35 and bf >= 0.2, (st, bf, n)
    st0, bf0, n0 = yields(0.0)
    assert n0 >= 990 and st0 < 0.2, (st0, bf0, n0)   # the legacy scene, still available (bench.py --scene flat)
