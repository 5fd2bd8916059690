"""GPU: the reference calls the seams of this path from different OS threads at the same time (FeatureDetection ||
Preprocess || Tracking || LocalBundleAdjustment, SURVEY.md section 8b: one thread per seam, no shared handle).  Four
threads, each with its own handle(s) and stream, hammer their seam concurrently; every result must equal the result
of the same call made alone."""
import os
import threading

import numpy as np
import pytest

import track_helpers as T
from helpers import SEED, make_stereo_case, rand_desc

pytestmark = pytest.mark.gpu

# named test_zz_*: collected last, so that a failure here cannot hide the parity suites behind `pytest -x`
ROUNDS = int(os.environ.get("SNK_THREAD_ROUNDS", "12"))


def test_seams_run_concurrently_on_their_own_handles(orc):
    from snake_slam_amd import synth
    from snake_slam_amd.ba import BARec, lba_options
    from snake_slam_amd.matcher import BruteForceMatcher, Preprocess
    from snake_slam_amd.orb import ORBExtractor
    from snake_slam_amd.tracking import SnakeORBMatcher

    rng = np.random.default_rng(SEED + 4242)
    imgs = [synth.stereo_frame(i, 320, 240, n_rects=90)[0] for i in range(3)]
    stereo = [make_stereo_case(rng, 300, 280) for _ in range(3)]
    bf_sets = [(rand_desc(rng, 257), rand_desc(rng, 301)) for _ in range(3)]
    track = [T.make_tracking_case(orc, rng, n_clutter=300, m_pts=250) for _ in range(3)]
    scenes = [synth.ba_scene(n_kf=8, n_pt=150, obs_per_pt=4, seed=900 + i)[0] for i in range(3)]

    def run_orb(ext, k):
        kps, desc = ext.Detect(imgs[k % 3])
        return kps.tobytes() + desc.tobytes()

    def run_pre(h, k):
        left, dl, right, dr, bf, ls = stereo[k % 3]
        n, rp, dp = h[0].StereoMatching(left, dl, right, dr, bf, ls, True)
        q, t = bf_sets[k % 3]
        h[1].matchKnn2(q, t)
        m = h[1].filterMatches(90, 0.9)
        return bytes([n & 255]) + rp.tobytes() + dp.tobytes() + np.asarray(h[1].matches).tobytes() + bytes([m & 255])

    def run_track(m, k):
        frame, cam, pose, ls, world, _ = track[k % 3]
        n, idx = m.SearchByProjectionFrameFrame2(frame, cam, pose, T.lm_coarse(orc, world), 15.0, 75, 0, ls)
        return bytes([n & 255]) + idx.tobytes()

    def run_ba(ba, k):
        # the reference's call pattern (LocalBundleAdjustment.cpp:353-413): new scene, initAndSolve, chi-square pass,
        # one more iteration -- twice, so that both the plain-launch path (first solve(1)) and the explicitly built
        # graph (second solve(1)) run while the other seams' threads allocate, copy and launch
        ba.create(scenes[k % 3])
        ci, cf = ba.initAndSolve()
        chi = ba.residuals(0)
        ba.set_outliers(0, (chi > 5.29).astype(np.uint8))
        c1 = ba.solve(1)
        c2 = ba.solve(1)
        pose, pt, it = ba.state(0)
        return ci.tobytes() + cf.tobytes() + chi.tobytes() + c1[1].tobytes() + c2[1].tobytes() + pose.tobytes() + pt.tobytes()

    def run_ba_one_call(ba, k):
        # the same call through snk_ba_solve_local_scene: device-built lists, the chi-square pass and the conditional extra
        # iteration enqueued in one go, results through the handle's pinned buffer
        ba.create(scenes[k % 3])
        n, ci, cf, pose, pt, flags = ba.solve_local_scene(4.41, 5.29)
        return bytes([n & 255]) + np.float64(ci).tobytes() + np.float64(cf).tobytes() + pose.tobytes() + pt.tobytes() + flags.tobytes()

    def run_frontend(fe, k):
        # snk_frontend_process: the sizes go A, B, A, A, B, A, ... -- every change reconfigures the handle (allocations, the graph
        # dropped), every A after an A RECORDS the launch chain with a stream capture while the other seams' threads copy, allocate
        # and launch on their own streams (a capture must not be invalidated by them: nothing in the library uses the legacy stream)
        l, r = fe_pairs[k % 3]
        f = fe.Process(l, r)
        return b"".join(np.ascontiguousarray(f[key]).tobytes() for key in ("keypoints", "descriptors", "undistorted_keypoints", "permutation",
                                                                           "right_points", "depth", "descriptors_right")) + bytes([f["n_stereo"] & 255])

    def run_frontend_pipelined(fe2, k):
        # snk_frontend_submit / collect on a handle of its own: two frames in flight, the older one collected (round 5) -- the slots'
        # stream captures and cooperative scheduling beside every other seam's thread
        key = ("keypoints", "descriptors", "undistorted_keypoints", "permutation", "right_points", "depth", "descriptors_right")
        fe2.Submit(*fe_pairs[0])
        fe2.Submit(*fe_pairs[2])
        a, b = fe2.Collect(), fe2.Collect()
        return b"".join(np.ascontiguousarray(f[x]).tobytes() for f in (a, b) for x in key) + bytes([a["n_stereo"] & 255, b["n_stereo"] & 255])

    def run_gba(gba, k):
        # a global scene: the PCG of every LM iteration is ONE cooperative launch (pcgl_persist, grid barriers inside) while the other
        # seams' kernels come and go on the same device
        gba.reset()
        ci, cf = gba.initAndSolve()
        pose, pt, it = gba.state(0)
        return ci.tobytes() + cf.tobytes() + pose.tobytes() + pt.tobytes() + bytes([it & 255])

    from snake_slam_amd.frontend import Frontend

    fe_pairs = [synth.stereo_frame(50, 320, 240, n_rects=90), synth.stereo_frame(51, 352, 256, n_rects=90), synth.stereo_frame(52, 320, 240, n_rects=90)]
    fe = Frontend((300, 1.2, 3, 20, 7), bounds=(0.0, 0.0, 352.0, 256.0))
    fe2 = Frontend((300, 1.2, 3, 20, 7), bounds=(0.0, 0.0, 352.0, 256.0))
    gba = BARec(lba_options(max_iterations=2, max_pcg_iterations=25))
    gba.create(synth.ba_scene(n_kf=60, n_pt=900, obs_per_pt=6, seed=77, n_fixed=1)[0])  # 354 x 354 reduced system: beyond one workgroup's LDS
    gba.initAndSolve()
    ext = ORBExtractor(300, 1.2, 3, 20, 7)
    pre, bfm, trk, ba, ba2 = Preprocess(), BruteForceMatcher(), SnakeORBMatcher(), BARec(lba_options()), BARec(lba_options())
    seams = [(run_orb, ext), (run_pre, (pre, bfm)), (run_track, trk), (run_ba, ba), (run_ba_one_call, ba2), (run_frontend, fe),
             (run_frontend_pipelined, fe2), (run_gba, gba)]
    try:
        want = [[fn(h, k) for k in range(3)] for fn, h in seams]  # each call alone
        errors, got = [], [[None] * ROUNDS for _ in seams]
        start = threading.Barrier(len(seams))

        def worker(si):
            fn, h = seams[si]
            try:
                start.wait()
                for k in range(ROUNDS):
                    got[si][k] = fn(h, k)
            except Exception as e:  # noqa: BLE001
                errors.append((si, repr(e)))

        threads = [threading.Thread(target=worker, args=(si,)) for si in range(len(seams))]
        for t in threads:
            t.start()
        for t in threads:
            t.join(timeout=120)
        assert not any(t.is_alive() for t in threads), "a seam hung"
        assert errors == []
        for si in range(len(seams)):
            for k in range(ROUNDS):
                assert got[si][k] == want[si][k % 3], (si, k)
    finally:
        ext.close(), pre.close(), bfm.close(), trk.close(), ba.close(), ba2.close(), fe.close(), fe2.close(), gba.close()
