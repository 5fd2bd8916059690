"""GPU: the sequence mode (BASELINE.json config 5) -- one stereo sequence tracked frame by frame through the host entry
points (snake_slam_amd/sequence.py) -- against the same chain restated with the oracle (tests/seq_helpers.py).  Keypoints,
descriptors, stereo matches and BF pairs are bit-exact, so the two trajectories can only differ by the pose refinement's
summation order (<= 1e-9)."""
import numpy as np
import pytest

import seq_helpers as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("w,h,orb,cam", [
    (752, 480, (1000, 1.2, 4, 20, 7), S.CAM),
    (640, 400, (800, 1.2, 4, 20, 7), (400.0, 400.0, 320.0, 200.0, 100.0)),
    # BASELINE.json configs[2]: KITTI 1241x376, 2000 features, 7 levels (reference configs/kitti.ini:30-34), the FULL stereo
    # chain -- extract L+R, rectify, grid, StereoMatching, kNN-2 + filter vs the previous frame, pose refinement -- against
    # the oracle chain (camera: KITTI sequence 00's P0 / bf)
    (1241, 376, (2000, 1.2, 7, 20, 7), (718.856, 718.856, 607.1928, 185.2157, 386.1448)),
])
def test_sequence_trajectory_matches_the_oracle_chain(orc, w, h, orb, cam):
    from snake_slam_amd import synth
    from snake_slam_amd.sequence import SequenceTracker, trajectory_block, trajectory_rows

    frames = list(synth.sequence_frames(3, 5, w, h, n_rects=400 if w > 700 else 300))
    trk = SequenceTracker(cam, orb=dict(nfeatures=orb[0], scale_factor=orb[1], n_levels=orb[2], ini_th_fast=orb[3], min_th_fast=orb[4]),
                          width=w, height=h)
    try:
        for t, (l, r) in enumerate(frames):
            trk.process(l, r, float(t))
        rows = trajectory_rows(trajectory_block(trk.rows, 8))
        stats = dict(trk.stats)
    finally:
        trk.close()
    want, _ = S.oracle_sequence(orc, frames, w, h, orb=orb, cam=cam)
    assert rows.shape == want.shape == (5, 8)
    assert np.allclose(rows, want, rtol=0, atol=1e-9)
    assert stats["bf_pairs"] > 4 * 50 and stats["inliers"] > 4 * 20
    # the rig moves 0.05 baselines to the right per frame: recovered within 35 % (the integer-drawn rectangles and the
    # occlusion corners bias it low), no comparable drift in y / z
    step = 0.05 * cam[4] / cam[0]
    assert (np.diff(rows[:, 1]) > 0).all()
    assert 0.65 * 4 * step < rows[-1, 1] < 1.35 * 4 * step
    assert abs(rows[-1, 2]) < 0.5 * 4 * step and abs(rows[-1, 3]) < 0.6 * 4 * step


def test_lockstep_kitti_shape(orc):
    """The lockstep tracker at BASELINE.json configs[2]'s shape (1241x376, 2000 features, 7 levels): two sequences, three frames."""
    from snake_slam_amd import synth
    from snake_slam_amd.sequence import MultiSequenceTracker

    w, h, orb, cam = 1241, 376, (2000, 1.2, 7, 20, 7), (718.856, 718.856, 607.1928, 185.2157, 386.1448)
    okw = dict(nfeatures=orb[0], scale_factor=orb[1], n_levels=orb[2], ini_th_fast=orb[3], min_th_fast=orb[4])
    seqs = [list(synth.sequence_frames(30 + s, 3, w, h)) for s in range(2)]
    mt = MultiSequenceTracker(cam, 2, 3, orb=okw, width=w, height=h)
    try:
        for t in range(3):
            mt.process([q[t][0] for q in seqs], [q[t][1] for q in seqs], float(t))
        rows, stats = mt.results()
    finally:
        mt.close()
    for s in range(2):
        want, _ = S.oracle_sequence(orc, seqs[s], w, h, orb=orb, cam=cam)
        assert np.allclose(rows[s], want, rtol=0, atol=1e-9), s
    assert stats["keypoints"] > 2 * 3 * 2 * 1900 and stats["inliers"] > 2 * 2 * 100


def test_sequences_in_lockstep_match_the_oracle_chain(orc):
    """MultiSequenceTracker: S sequences on one GPU in lockstep, device resident (batched entry points + the two glue kernels
    snk_track_bf_matches_batch_dev / snk_track_backproject_batch_dev): every sequence's trajectory equals the oracle chain's
    (poses within 1e-9: everything before the pose refinement is bit-exact) and the host-API tracker's counters."""
    from snake_slam_amd import synth
    from snake_slam_amd.sequence import MultiSequenceTracker, SequenceTracker

    w, h, orb, cam = 640, 400, (800, 1.2, 4, 20, 7), (400.0, 400.0, 320.0, 200.0, 100.0)
    okw = dict(nfeatures=orb[0], scale_factor=orb[1], n_levels=orb[2], ini_th_fast=orb[3], min_th_fast=orb[4])
    S, T = 3, 4
    seqs = [list(synth.sequence_frames(10 + s, T, w, h, n_rects=300)) for s in range(S)]
    mt = MultiSequenceTracker(cam, S, T, orb=okw, width=w, height=h)
    try:
        for t in range(T):
            mt.process([seqs[s][t][0] for s in range(S)], [seqs[s][t][1] for s in range(S)], float(t))
        rows, stats = mt.results()
    finally:
        mt.close()
    tot = dict(keypoints=0, stereo=0, bf_pairs=0, inliers=0)
    for s in range(S):
        want, _ = S_helpers_oracle(orc, seqs[s], w, h, orb, cam)
        assert rows[s].shape == want.shape == (T, 8)
        assert np.allclose(rows[s], want, rtol=0, atol=1e-9), s
        trk = SequenceTracker(cam, orb=okw, width=w, height=h)
        try:
            for t, (l, r) in enumerate(seqs[s]):
                trk.process(l, r, float(t))
            for k in tot:
                tot[k] += trk.stats[k]
        finally:
            trk.close()
    assert stats["frames"] == S * T and all(stats[k] == tot[k] for k in tot), (stats, tot)
    assert stats["bf_pairs"] > S * (T - 1) * 50 and stats["inliers"] > S * (T - 1) * 20


def S_helpers_oracle(orc, frames, w, h, orb, cam):
    return S.oracle_sequence(orc, frames, w, h, orb=orb, cam=cam)


def test_lockstep_with_an_empty_sequence_and_a_single_sequence(orc):
    """Edge cases of the lockstep tracker: a sequence of featureless images next to ordinary ones (no keypoints, no pairs: its
    pose stays the identity and the others are unaffected), and S = 1 (a batch of two images, the small-launch paths)."""
    from snake_slam_amd import synth
    from snake_slam_amd.sequence import MultiSequenceTracker

    w, h, orb, cam = 640, 400, (800, 1.2, 4, 20, 7), (400.0, 400.0, 320.0, 200.0, 100.0)
    okw = dict(nfeatures=orb[0], scale_factor=orb[1], n_levels=orb[2], ini_th_fast=orb[3], min_th_fast=orb[4])
    T = 3
    real = [list(synth.sequence_frames(20 + s, T, w, h, n_rects=300)) for s in range(2)]
    blank = [(np.full((h, w), 90, np.uint8), np.full((h, w), 90, np.uint8)) for _ in range(T)]
    seqs = [real[0], blank, real[1]]
    mt = MultiSequenceTracker(cam, 3, T, orb=okw, width=w, height=h)
    try:
        for t in range(T):
            mt.process([s[t][0] for s in seqs], [s[t][1] for s in seqs], float(t))
        rows, stats = mt.results()
    finally:
        mt.close()
    for k, s in ((0, 0), (2, 1)):
        want, _ = S.oracle_sequence(orc, real[s], w, h, orb=orb, cam=cam)
        assert np.allclose(rows[k], want, rtol=0, atol=1e-9), k
    assert np.allclose(rows[1][:, 1:4], 0.0) and np.allclose(rows[1][:, 4:], [0, 0, 0, 1])  # identity throughout
    one = MultiSequenceTracker(cam, 1, T, orb=okw, width=w, height=h)
    try:
        for t in range(T):
            one.process([real[0][t][0]], [real[0][t][1]], float(t))
        r1, _ = one.results()
    finally:
        one.close()
    assert np.array_equal(r1[0], rows[0])  # the same sequence alone: bit-identical trajectory (no cross-talk between sequences)
