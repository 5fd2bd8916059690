"""CPU: pin the feature-grid / projection-matcher oracle (PARITY UNPINNED for the saiga helpers)."""
import numpy as np

import track_helpers as T
from helpers import SEED


def test_det_log_exp_accuracy(orc):
    rng = np.random.default_rng(1)
    for x in list(rng.uniform(1e-6, 1e6, 200)) + [1.0, 1.2, 0.5, 2.0]:
        assert abs(orc.det_log(x) - np.log(x)) <= 4e-16 * max(1.0, abs(np.log(x)))
    for y in list(rng.uniform(-20, 20, 200)) + [0.0, 1.0]:
        assert abs(orc.det_exp(y) / np.exp(y) - 1) <= 4e-15
    assert orc.det_log(1.0) == 0.0 and orc.det_exp(0.0) == 1.0


def test_feature_grid_definition(orc):
    rng = np.random.default_rng(2)
    n = 500
    kps = np.zeros(n, orc.KP64)
    kps["x"] = rng.uniform(T.BOUNDS[0] - 5, T.BOUNDS[2] + 5, n)  # some outside -> clamped
    kps["y"] = rng.uniform(T.BOUNDS[1] - 5, T.BOUNDS[3] + 5, n)
    perm, cs, cols, rows = orc.feature_grid(kps, T.BOUNDS)
    assert cols == int(np.ceil((T.BOUNDS[2] - T.BOUNDS[0]) / 20)) and rows == int(np.ceil((T.BOUNDS[3] - T.BOUNDS[1]) / 20))
    assert sorted(perm.tolist()) == list(range(n)) and cs[0] == 0 and cs[-1] == n
    cx = np.clip(np.floor((kps["x"] - T.BOUNDS[0]) / 20), 0, cols - 1).astype(int)
    cy = np.clip(np.floor((kps["y"] - T.BOUNDS[1]) / 20), 0, rows - 1).astype(int)
    cell = cx * rows + cy
    order = np.argsort(cell, kind="stable")  # x-major cells, original order inside a cell
    want = np.zeros(n, int)
    want[order] = np.arange(n)
    assert np.array_equal(perm, want)
    assert np.array_equal(np.diff(cs), np.bincount(cell, minlength=cols * rows))


def test_coarse_matches_brute_force(orc):
    for seed, direction, th in [(1, 0, 15.0), (2, 1, 10.0), (3, 2, 15.0), (4, 0, 30.0)]:
        rng = np.random.default_rng(SEED + seed)
        frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=300, m_pts=200)
        pts = T.lm_coarse(orc, world)
        n, idx = orc.match_coarse(frame, cam, pose, pts, th, 75, direction, ls)
        n2, idx2 = T.brute_coarse(frame, cam, pose, pts, th, 75, direction, ls)
        assert n == n2 and np.array_equal(idx, idx2)
        assert n > 20
        got = idx[idx >= 0]
        assert len(set(got.tolist())) == len(got) and (frame["taken"][got] == 0).all()


def test_fine_and_keyframe_invariants(orc):
    rng = np.random.default_rng(SEED + 9)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=400, m_pts=300)
    pts = T.lm_fine(orc, rng, world, pose, ls)
    n, idx, vis, valid = orc.match_fine(frame, cam, pose, pts, 5.0, 0.8, ls)
    assert n == (idx >= 0).sum() and n > 30
    got = idx[idx >= 0]
    assert len(set(got.tolist())) == len(got) and (frame["taken"][got] == 0).all()
    assert (valid <= pts["valid"]).all() and valid.sum() < pts["valid"].sum()  # culls only clear flags
    assert (vis <= valid).all() and (idx[valid == 0] == -1).all()
    # th == 1.0 takes the "no factor" branch and shrinks the windows
    n1, _, _, _ = orc.match_fine(frame, cam, pose, pts, 1.0, 0.8, ls)
    assert n1 <= n
    # keyframe matcher: sequential exclusivity
    skip = (rng.random(len(world["pos"])) < 0.1).astype(np.uint8)
    nk, ik = orc.match_keyframe(frame, cam, pose, world["pos"], world["desc"], skip, 15.0, 100)
    gk = ik[ik >= 0]
    assert nk == len(gk) and len(set(gk.tolist())) == len(gk) and (ik[skip == 1] == -1).all() and nk > 30


def test_keyframe_matcher_is_sequential_greedy(orc):
    """Two points project to the same place; the first takes the best feature, the second the next best."""
    from oracle.oracle import KP64

    kps = np.zeros(2, KP64)
    kps["x"], kps["y"] = [100.0, 103.0], [100.0, 100.0]
    desc = np.zeros((2, 4), np.uint64)
    desc[1, 0] = 0b111
    perm, cs, cols, rows = orc.feature_grid(kps, T.BOUNDS)
    assert perm.tolist() == [0, 1]
    frame = dict(kps=kps, desc=desc, right_points=np.full(2, -1, np.float32), taken=np.zeros(2, np.uint8), cell_start=cs,
                 bounds=T.BOUNDS, cols=cols, rows=rows)
    cam = (100.0, 100.0, 100.0, 100.0, 10.0)
    pose = np.array([0, 0, 0, 1.0, 0, 0, 0])
    pos = np.array([[0.0, 0.0, 1.0], [0.0, 0.0, 1.0]])  # both project to (100, 100)
    pd = np.zeros((2, 4), np.uint64)
    n, idx = orc.match_keyframe(frame, cam, pose, pos, pd, np.zeros(2, np.uint8), 10.0, 50)
    assert n == 2 and idx.tolist() == [0, 1]
    # the parallel matchers instead drop the second claimant
    pts = np.zeros(2, orc.LM_COARSE)
    pts["pos"], pts["normal"] = pos, [[0, 0, -1.0], [0, 0, -1.0]]
    nc, ic = orc.match_coarse(frame, cam, pose, pts, 10.0, 50, 0, np.ones(4, np.float32))
    assert nc == 1 and ic.tolist() == [0, -1]


def test_fuse_matches_brute_force(orc):
    """MappingORBMatcher::Fuse (LocalMap overload): grid query == exhaustive scan, every point independent."""
    for seed, th, of, fth in [(21, 4.0, 2.0, 50), (22, 3.0, 1.5, 60), (23, 6.0, 3.0, 40)]:
        rng = np.random.default_rng(SEED + seed)
        frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=300, m_pts=250)
        pts = T.fusion_points(orc, rng, world, pose, ls)
        mask = (rng.random(len(pts)) > 0.2).astype(np.uint8)
        n, idx = orc.match_fuse(frame, cam, pose, pts, mask, th, of, fth, ls)
        n2, idx2 = T.brute_fuse(orc, frame, cam, pose, pts, mask, th, of, fth, ls)
        assert n == n2 and np.array_equal(idx, idx2)
        assert n > 15 and (idx[mask == 0] == -1).all()
        # no mask == all-ones mask
        na, ia = orc.match_fuse(frame, cam, pose, pts, None, th, of, fth, ls)
        nb, ib = orc.match_fuse(frame, cam, pose, pts, np.ones(len(pts), np.uint8), th, of, fth, ls)
        assert na == nb and np.array_equal(ia, ib) and na >= n


def test_triangulation_project_matches_brute_force(orc):
    """MappingORBMatcher::SearchForTriangulationProject: depth-grid projection + epipolar + Hamming gates."""
    for seed, epi, fd in [(31, 4.0, 50), (32, 2.0, 40), (33, 8.0, 64)]:
        rng = np.random.default_rng(SEED + seed)
        c = T.make_triangulation_case(orc, rng, m_pts=300, n_clutter=200)
        n, idx = orc.match_triangulation_project(c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"], c["desc1"],
                                                 c["has1"], c["frame2"], c["np2"], c["E"], epi, fd)
        n2, idx2 = T.brute_triangulation(c, epi, fd)
        assert n == n2 and np.array_equal(idx, idx2)
        assert n > 10 and (idx[c["has1"] == 1] == -1).all()
        got = idx[idx >= 0]
        assert (c["frame2"]["taken"][got] == 0).all()


def _bow_call(orc, c, epi, fd):
    return orc.match_triangulation_bow(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["bow1"], c["np2"], c["desc2"],
                                       c["has2"], c["bow2"], epi, fd)


def test_triangulation_bow_matches_brute_force(orc):
    """MappingORBMatcher::SearchForTriangulation2: common vocabulary nodes, Hamming gate, then epipolar gate."""
    for seed, epi, fd in [(41, 4.0, 50), (42, 1.0, 40), (43, 8.0, 64), (44, 4.0, 20)]:
        rng = np.random.default_rng(SEED + seed)
        c = T.make_bow_case(rng, m_pts=300, n_clutter=150, n_nodes=40)
        n, pairs = _bow_call(orc, c, epi, fd)
        want = T.brute_triangulation_bow(c, epi, fd)
        assert n == len(want) and [tuple(p) for p in pairs.tolist()] == want
        assert n > 10
        assert (c["has1"][pairs[:, 0]] == 0).all() and (c["has2"][pairs[:, 1]] == 0).all()


def test_triangulation_bow_edge_cases(orc):
    rng = np.random.default_rng(SEED + 45)
    c = T.make_bow_case(rng, m_pts=60, n_clutter=20, n_nodes=8)
    # no common node
    c2 = dict(c)
    ids2, s2, f2 = c["bow2"]
    c2["bow2"] = ((ids2 + 1).astype(np.uint32), s2, f2)
    assert _bow_call(orc, c2, 4.0, 50)[0] == 0
    # empty feature vectors
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    c3 = dict(c, bow1=empty)
    assert _bow_call(orc, c3, 4.0, 50)[0] == 0
    c4 = dict(c, bow2=empty)
    assert _bow_call(orc, c4, 4.0, 50)[0] == 0
    # every feature of keyframe 2 already holds a point
    c5 = dict(c, has2=np.ones_like(c["has2"]))
    assert _bow_call(orc, c5, 4.0, 50)[0] == 0
    # ties in Hamming distance: identical descriptors in one node -> the LAST candidate of the node's list wins (:66)
    c6 = dict(c)
    c6["desc2"] = np.repeat(c["desc1"][:1], len(c["desc2"]), 0)
    c6["desc1"] = np.repeat(c["desc1"][:1], len(c["desc1"]), 0)
    c6["has1"] = np.zeros_like(c["has1"])
    c6["has2"] = np.zeros_like(c["has2"])
    n, pairs = _bow_call(orc, c6, 1e6, 50)
    want = T.brute_triangulation_bow(c6, 1e6, 50)
    assert [tuple(p) for p in pairs.tolist()] == want and n > 0
    ids2, s2, f2 = c6["bow2"]
    last_of = {int(nid): int(f2[s2[k + 1] - 1]) for k, nid in enumerate(ids2) if s2[k + 1] > s2[k]}
    ids1, s1, f1 = c6["bow1"]
    node_of1 = {int(f): int(ids1[k]) for k in range(len(ids1)) for f in f1[s1[k]:s1[k + 1]]}
    for i, j in pairs.tolist():
        assert j == last_of[node_of1[i]]


def test_triangulation_bf_matches_brute_force(orc):
    """MappingORBMatcher::SearchForTriangulationBF: epipolar gate (10 px) then Hamming, all pairs."""
    for seed, fd in [(51, 50), (52, 35), (53, 64)]:
        rng = np.random.default_rng(SEED + seed)
        c = T.make_bow_case(rng, m_pts=200, n_clutter=100)
        n, idx = orc.match_triangulation_bf(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["np2"], c["desc2"], c["has2"], fd)
        want = T.brute_triangulation_bf(c, fd)
        assert np.array_equal(idx, want) and n == (want >= 0).sum()
        assert n > 10 and (idx[c["has1"] == 1] == -1).all()
    # empty sides
    z2, zd, zh = np.zeros((0, 2)), np.zeros((0, 4), np.uint64), np.zeros(0, np.uint8)
    assert orc.match_triangulation_bf(c["cam"], c["E"], z2, zd, zh, c["np2"], c["desc2"], c["has2"], 50)[0] == 0
    n, idx = orc.match_triangulation_bf(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], z2, zd, zh, 50)
    assert n == 0 and (idx == -1).all()


def test_relink_matches_brute_force(orc):
    """DeferredMapper::Relink per-observation search: outlier erase, radius query, stereo gate, strict improvement."""
    for seed in (61, 62, 63):
        rng = np.random.default_rng(SEED + seed)
        frame, cam, pose, qs = T.make_relink_case(orc, rng, n_base=300)
        n, action, best = orc.match_relink(frame, cam, pose, qs)
        wa, wb = T.brute_relink(frame, cam, pose, qs)
        assert np.array_equal(action, wa) and np.array_equal(best, wb)
        assert n == (wa != 0).sum()
        assert (wa == 1).sum() > 10 and (wa == 2).sum() > 10 and (wa == 0).sum() > 10
        assert (best[action == 2] != qs["feature"][action == 2]).all()
    assert orc.match_relink(frame, cam, pose, qs[:0])[0] == 0
    # a tighter feature threshold can only remove relinks; thresholds are strict (<)
    n25, a25, b25 = orc.match_relink(frame, cam, pose, qs, feature_threshold=25)
    n1, a1, b1 = orc.match_relink(frame, cam, pose, qs, feature_threshold=1)
    assert ((a1 == 2) <= (a25 == 2)).all()
    for k in np.nonzero(a1 == 2)[0]:
        assert T._ham(qs["desc"][k], frame["desc"][b1[k]]) == 0


def test_matchers_do_not_depend_on_the_thread_count(orc):
    """The reference runs the per-point phase with 4 OpenMP threads (SnakeORBMatcher.cpp:219, :379); bench.py's CPU baseline
    does the same with the oracle.  The result must be the single-threaded one."""
    rng = np.random.default_rng(SEED + 321)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=500, m_pts=900)
    pc, pf = T.lm_coarse(orc, world), T.lm_fine(orc, rng, world, pose, ls)
    a = orc.match_coarse(frame, cam, pose, pc, 15.0, 75, 0, ls)
    b = orc.match_fine(frame, cam, pose, pf, 5.0, 0.8, ls)
    orc.set_match_threads(4)
    try:
        a4 = orc.match_coarse(frame, cam, pose, pc, 15.0, 75, 0, ls)
        b4 = orc.match_fine(frame, cam, pose, pf, 5.0, 0.8, ls)
    finally:
        orc.set_match_threads(1)
    assert a[0] == a4[0] and np.array_equal(a[1], a4[1])
    assert b[0] == b4[0] and all(np.array_equal(x, y) for x, y in zip(b[1:], b4[1:]))
