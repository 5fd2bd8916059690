"""GPU: one stereo frame through the front-end in ONE call (snk_frontend_process = FeatureDetector::Detect x 2 + Preprocess::Process,
reference Snake/Preprocess/FeatureDetector.cpp:116-156, Snake/Preprocess/Preprocess.cpp:35-53) against (a) the oracle chain and (b) the
call-by-call path through the host entry points -- bit for bit, on the first frame (plain launches), the second (recorded as a
hipGraph) and later ones (replayed), across an image-size change and with the graph switched off."""
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

E_K, E_D = (458.654, 457.296, 367.215, 248.375), (-0.28340811, 0.07395907, 0.00019359, 1.76187114e-05, 0.0, 0.0, 0.0, 0.0)
E_K2, E_D2 = (457.587, 456.134, 379.999, 255.238), (-0.28368365, 0.07451284, -0.00010473, -3.55590700e-05, 0.0, 0.0, 0.0, 0.0)


def oracle_frame(orc, orb, rl, rr, bounds, bf, ls, left, right):
    p = orc.orb_params(*orb)
    kl, dl = orc.orb_detect(p, left)
    out = dict(N=len(kl))
    ul, nl = orc.rectify(rl, kl)
    perm, cell_start, cols, rows = orc.feature_grid(ul, bounds)
    perm = np.asarray(perm)
    g, gd, gk, gn = np.zeros_like(ul), np.zeros_like(dl), np.zeros_like(kl), np.zeros_like(nl)
    g[perm], gd[perm], gk[perm], gn[perm] = ul, dl, kl, nl
    out.update(keypoints=gk, descriptors=gd, undistorted_keypoints=g, normalized_points=gn, permutation=perm.astype(np.int32),
               cell_start=np.asarray(cell_start, np.int32), cols=cols, rows=rows)
    if right is not None:
        kr, dr = orc.orb_detect(p, right)
        ur, _ = orc.rectify(rr, kr)
        n_st, rp, depth = orc.stereo_match(g, gd, ur, dr, bf, ls, True)
        out.update(keypoints_right=kr, descriptors_right=dr, n_right=len(kr), n_stereo=int(n_st), right_points=rp, depth=depth)
    return out


def assert_same(got, want, what):
    for k, w in want.items():
        g = got[k]
        if isinstance(w, np.ndarray) and w.dtype.names:
            for f in w.dtype.names:
                assert np.array_equal(g[f], w[f]), f"{what}: {k}.{f}"
        else:
            assert np.array_equal(np.asarray(g), np.asarray(w)), f"{what}: {k}"


def test_one_call_frame_equals_oracle_and_call_by_call(orc):
    from snake_slam_amd import synth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Preprocess, Rectification
    from snake_slam_amd.orb import ORBExtractor
    from snake_slam_amd.tracking import FeatureGrid

    orb = (1000, 1.2, 4, 20, 7)
    bounds, bf = (-120.0, -60.0, 880.0, 540.0), 47.9
    fe = Frontend(orb, Rectification.make(E_K, E_D), Rectification.make(E_K2, E_D2), bounds, bf)
    ls = fe.level_scale
    orl, orr = orc.rectification(E_K, E_D), orc.rectification(E_K2, E_D2)
    ext, pre, grid = ORBExtractor(*orb), Preprocess(0), FeatureGrid(0)
    try:
        for t in range(4):  # frame 0 plain, frame 1 captured, frames 2-3 replayed
            left, right = synth.stereo_frame(20 + t, 752, 480)
            got = fe.Process(left, right)
            want = oracle_frame(orc, orb, orl, orr, bounds, bf, ls, left, right)
            assert got["N"] > 900 and got["n_stereo"] > 100
            assert_same(got, want, f"frame {t} vs oracle")
            # call by call through the host entry points (five synchronisations)
            kl, dl = ext.Detect(left)
            kr, dr = ext.Detect(right)
            ul, nl = pre.rectify(Rectification.make(E_K, E_D), kl)
            ur, _ = pre.rectify(Rectification.make(E_K2, E_D2), kr)
            perm = np.asarray(grid.create(bounds, ul)[0])
            g, gd = np.zeros_like(ul), np.zeros_like(dl)
            g[perm], gd[perm] = ul, dl
            n_st, rp, depth = pre.StereoMatching(g, gd, ur, dr, bf, ls, True)
            assert n_st == got["n_stereo"] and np.array_equal(rp, got["right_points"]) and np.array_equal(depth, got["depth"])
            assert np.array_equal(gd, got["descriptors"]) and np.array_equal(dr, got["descriptors_right"])
        # another image size through the same handle (reconfigures, drops the graph), then back
        for (w, h), seed in (((640, 400), 7), ((752, 480), 8), ((752, 480), 9)):
            left, right = synth.stereo_frame(seed, w, h)
            got = fe.Process(left, right)
            assert_same(got, oracle_frame(orc, orb, orl, orr, bounds, bf, ls, left, right), f"size {w}x{h}")
        # a featureless pair: no keypoints, nothing matched, nothing written beyond the counts
        flat = np.full((480, 752), 90, np.uint8)
        got = fe.Process(flat, flat)
        assert got["N"] == 0 and got["n_right"] == 0 and got["n_stereo"] == 0 and not got["cell_start"].any()
    finally:
        for hnd in (fe, ext, pre, grid):
            hnd.close()


def test_mono_frame_and_kitti_size(orc):
    """stereo = 0 (settings.inputType == Mono: left image only, no StereoMatching) and the KITTI configuration (1241x376, 2000
    features, 7 levels; reference configs/kitti.ini:30-34)."""
    from snake_slam_amd import synth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Rectification

    orb = (2000, 1.2, 7, 20, 7)
    bounds = (0.0, 0.0, 1241.0, 376.0)
    k = (718.856, 718.856, 607.1928, 185.2157)
    for stereo in (False, True):
        fe = Frontend(orb, Rectification.make(k), None, bounds, 386.1448, stereo=stereo)
        try:
            for t in range(3):
                left, right = synth.stereo_frame(40 + t, 1241, 376)
                got = fe.Process(left, right if stereo else None)
                want = oracle_frame(orc, orb, orc.rectification(k), orc.rectification(k), bounds, 386.1448, fe.level_scale, left,
                                    right if stereo else None)
                assert got["N"] > 1500
                assert_same(got, want, f"kitti stereo={stereo} frame {t}")
                if not stereo:
                    assert got["n_right"] == 0 and got["n_stereo"] == 0 and (got["right_points"] == -1000).all() and (got["depth"] == -1000).all()
        finally:
            fe.close()


def test_without_graph_and_under_a_definition_change(orc, tmp_path):
    """SNK_FRONTEND_NO_GRAPH=1 (plain launches for every frame) gives the same frames; and a change of the "iround.mode" definition
    between frames -- a kernel argument the recorded graph holds -- is honoured (the graph is rebuilt)."""
    code = r"""
import numpy as np, sys
from snake_slam_amd import synth, _lib
from snake_slam_amd.frontend import Frontend
fe = Frontend()
out = []
for t in range(4):
    if t == 3:
        _lib.set_definition("iround.mode", 2)
    l, r = synth.stereo_frame(60 + t, 752, 480)
    f = fe.Process(l, r)
    out += [f["descriptors"], f["right_points"], f["depth"], f["permutation"], np.array([f["n_stereo"]])]
fe.close()
np.savez(sys.argv[1], *out)
print("ok")
"""
    res = {}
    for name, env in (("graph", {}), ("plain", {"SNK_FRONTEND_NO_GRAPH": "1"})):
        f = str(tmp_path / (name + ".npz"))
        r = subprocess.run([sys.executable, "-c", code, f], env=dict(os.environ, PYTHONPATH=str(ROOT), **env), capture_output=True, text=True,
                           cwd=str(ROOT), timeout=600)
        assert r.returncode == 0 and "ok" in r.stdout, (name, r.stdout[-1500:], r.stderr[-1500:])
        res[name] = np.load(f)
    for k in res["graph"].files:
        assert np.array_equal(res["graph"][k], res["plain"][k]), k
    # the last frame ran under iround.mode = 2: the oracle with that definition agrees with it
    from snake_slam_amd import synth
    from snake_slam_amd.frontend import Frontend  # noqa: F401  (level scales)

    l, r = synth.stereo_frame(63, 752, 480)
    ls = np.cumprod(np.array([1.0, 1.2, 1.2, 1.2], np.float32), dtype=np.float32)
    rect = orc.rectification((1.0, 1.0, 0.0, 0.0))
    orc.set_definition("iround.mode", 2)
    try:
        want = oracle_frame(orc, (1000, 1.2, 4, 20, 7), rect, rect, (0.0, 0.0, 752.0, 480.0), 47.9, ls, l, r)
    finally:
        orc.set_definition("iround.mode", 0)
    files = res["graph"].files
    assert np.array_equal(res["graph"][files[-4]], want["right_points"]) and int(res["graph"][files[-1]][0]) == want["n_stereo"]


def test_invalid_arguments(orc):
    import ctypes as C

    from snake_slam_amd import _lib
    from snake_slam_amd.frontend import Frontend, FrontendFrame

    lib = _lib.load()
    fe = Frontend()
    fr = FrontendFrame()
    img = np.zeros((480, 752), np.uint8)
    assert lib.snk_frontend_process(fe._h, None, 752, img.ctypes.data, 752, 752, 480, C.byref(fr)) != 0        # no left image
    assert lib.snk_frontend_process(fe._h, img.ctypes.data, 700, img.ctypes.data, 752, 752, 480, C.byref(fr)) != 0  # pitch < width
    assert lib.snk_frontend_process(fe._h, img.ctypes.data, 752, None, 752, 752, 480, C.byref(fr)) != 0        # stereo handle, no right image
    assert lib.snk_frontend_process(fe._h, img.ctypes.data, 752, img.ctypes.data, 752, 752, 480, None) != 0
    from snake_slam_amd import synth

    l, r = synth.stereo_frame(1, 752, 480)
    fr.capacity = 10  # too small: SNK_ERR_CAPACITY with the counts set
    assert lib.snk_frontend_process(fe._h, l.ctypes.data, 752, r.ctypes.data, 752, 752, 480, C.byref(fr)) == 4 and fr.n > 900
    fe.close()


def test_pipelined_submit_collect_is_bit_identical(orc):
    """snk_frontend_submit / snk_frontend_collect (the reference's FeatureDetection -> Preprocess stage queue,
    Snake/Preprocess/FeatureDetector.h:39, Preprocess.h:36): frames come back in submission order and every array is bit for bit what
    snk_frontend_process returns for that frame -- depths 1..4, through each slot's uncaptured, captured and replayed frames, and with
    `depth` frames in flight at once."""
    from snake_slam_amd import SnakeHipError, synth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Rectification

    orb = (1000, 1.2, 4, 20, 7)
    bounds, bf = (-120.0, -60.0, 880.0, 540.0), 47.9
    fe = Frontend(orb, Rectification.make(E_K, E_D), Rectification.make(E_K2, E_D2), bounds, bf)
    pairs = [synth.stereo_frame(40 + k, 752, 480) for k in range(6)]
    pairs[4] = (np.full((480, 752), 9, np.uint8), pairs[4][1])  # an empty left image in the stream
    try:
        want = [fe.Process(l, r) for l, r in pairs]
        for depth in (1, 2, 3, 4):
            fe.set_depth(depth)
            n_frames = 4 * depth + 3
            got = []
            for k in range(n_frames + depth - 1):
                if k < n_frames:
                    fe.Submit(*pairs[k % len(pairs)])
                    assert fe.in_flight() == min(k + 1, depth)
                if k >= depth - 1:
                    got.append(fe.Collect())
            assert fe.in_flight() == 0 and len(got) == n_frames
            for k, g in enumerate(got):
                assert_same(g, want[k % len(pairs)], f"depth {depth} frame {k}")
        # the synchronous call refuses to run under frames in flight; a collect with nothing submitted times out
        fe.Submit(*pairs[0])
        with pytest.raises(SnakeHipError, match="in flight"):
            fe.Process(*pairs[1])
        small = synth.stereo_frame(1, 320, 240, n_rects=60)
        with pytest.raises(SnakeHipError, match="in flight"):
            fe.Submit(*small)
        assert_same(fe.Collect(), want[0], "after the refused calls")
        with pytest.raises(SnakeHipError, match="no frame was submitted"):
            fe.Collect(timeout_ms=20)
        # another image size once the pipeline is empty, then back
        w_small = fe.Process(*small)
        fe.Submit(*small), fe.Submit(*small)
        assert_same(fe.Collect(), w_small, "small 0"), assert_same(fe.Collect(), w_small, "small 1")
        fe.Submit(*pairs[2])
        assert_same(fe.Collect(), want[2], "back to 752x480")
    finally:
        fe.close()


def test_submit_and_collect_on_different_threads(orc):
    """One thread submits, another collects (the reference's FeatureDetection and Preprocess threads): 60 frames through a depth-3
    pipeline, every frame identical to the synchronous call's result, in order."""
    import threading

    from snake_slam_amd import synth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Rectification

    fe = Frontend((1000, 1.2, 4, 20, 7), Rectification.make(E_K, E_D), Rectification.make(E_K2, E_D2), (-120.0, -60.0, 880.0, 540.0), 47.9)
    pairs = [synth.stereo_frame(70 + k, 752, 480) for k in range(5)]
    try:
        want = [fe.Process(l, r) for l, r in pairs]
        fe.set_depth(3)
        N, errors, got = 60, [], []

        def producer():
            try:
                for k in range(N):
                    fe.Submit(*pairs[k % len(pairs)])
            except Exception as e:  # noqa: BLE001
                errors.append(e)

        t = threading.Thread(target=producer)
        t.start()
        for k in range(N):
            got.append(fe.Collect(timeout_ms=20000))
        t.join()
        assert not errors, errors
        for k, g in enumerate(got):
            assert_same(g, want[k % len(pairs)], f"frame {k}")
    finally:
        fe.close()


def test_submit_pinned_and_peek(orc):
    """snk_frontend_submit_pinned (round 6: caller-owned page-locked images, no staging copy; Snake/Preprocess/Input.h:48) and
    snk_frontend_peek (what the collecting thread sizes its arrays from): every array bit for bit what snk_frontend_process returns,
    with the caller's pitch kept on the device (752: a multiple of four), with a padded pitch (768), with an odd pitch (755: the 2-D
    copy), with left and right adjacent in memory (one upload) and apart (two), alternating with the staged submit on the same
    slots (the recorded launch sequence is keyed by the row pitch), and for a mono handle."""
    import ctypes as C

    from snake_slam_amd import SnakeHipError, synth
    from snake_slam_amd.frontend import Frontend
    from snake_slam_amd.matcher import Rectification

    orb = (1000, 1.2, 4, 20, 7)
    bounds, bf = (-120.0, -60.0, 880.0, 540.0), 47.9
    fe = Frontend(orb, Rectification.make(E_K, E_D), Rectification.make(E_K2, E_D2), bounds, bf)
    pairs = [synth.stereo_frame(90 + k, 752, 480) for k in range(4)]
    try:
        want = [fe.Process(l, r) for l, r in pairs]
        lib = fe._lib
        w_, h_, cap = C.c_int(0), C.c_int(0), C.c_int(0)
        assert lib.snk_frontend_peek(fe._h, 10, C.byref(w_), C.byref(h_), C.byref(cap)) == 6  # SNK_ERR_TIMEOUT: nothing submitted
        for pitch in (752, 768, 755):
            bufs = [fe.pinned_images(752, 480, 2, pitch) for _ in range(4)]    # adjacent: right = left + pitch * height
            lone = [fe.pinned_images(752, 480, 1, pitch) for _ in range(4)]    # a right image somewhere else
            for k, (l, r) in enumerate(pairs):
                bufs[k][0, :, :752], bufs[k][1, :, :752] = l, r
                lone[k][0, :, :752] = r
            got = []
            for rnd in range(3):  # uncaptured, captured, replayed on every slot
                for k in range(4):
                    right = bufs[k][1, :, :752] if (k + rnd) % 2 == 0 else lone[k][0, :, :752]
                    fe.SubmitPinned(bufs[k][0, :, :752], right)
                    if k == 1:
                        assert lib.snk_frontend_peek(fe._h, -1, C.byref(w_), C.byref(h_), C.byref(cap)) == 0
                        assert (w_.value, h_.value) == (752, 480) and cap.value == fe._frame.capacity
                    if fe.in_flight() == 3:
                        got.append(fe.Collect())
                while fe.in_flight():
                    got.append(fe.Collect())
            assert len(got) == 12
            for k, g in enumerate(got):
                assert_same(g, want[k % 4], f"pitch {pitch} frame {k}")
            # the staged submit in between (its rows have pitch 768 on the device), then pinned again
            fe.Submit(*pairs[1]), fe.SubmitPinned(bufs[2][0, :, :752], bufs[2][1, :, :752]), fe.Submit(*pairs[3])
            assert_same(fe.Collect(), want[1], "staged"), assert_same(fe.Collect(copy=False), want[2], "pinned"), assert_same(fe.Collect(), want[3], "staged")
        rc = lib.snk_frontend_submit_pinned(fe._h, None, 752, None, 752, 752, 480)
        assert rc == 1 and b"bad left image" in lib.snk_last_error()  # SNK_ERR_INVALID_ARG
    finally:
        fe.close()
    mono = Frontend(orb, Rectification.make(E_K, E_D), None, bounds, bf, stereo=False)
    try:
        l = pairs[0][0]
        want_m = mono.Process(l)
        buf = mono.pinned_images(752, 480, 1)
        buf[0] = l
        mono.SubmitPinned(buf[0]), mono.SubmitPinned(buf[0])
        assert_same(mono.Collect(), want_m, "mono 0"), assert_same(mono.Collect(), want_m, "mono 1")
    finally:
        mono.close()
