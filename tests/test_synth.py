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
