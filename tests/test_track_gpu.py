"""GPU: feature grid and projection matchers vs the oracle through the C ABI — bit-exact indices."""
import os

import numpy as np
import pytest

import track_helpers as T
from helpers import SEED

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def matcher():
    from snake_slam_amd.tracking import SnakeORBMatcher

    m = SnakeORBMatcher(0)
    yield m
    m.close()


def test_feature_grid_parity(orc):
    from snake_slam_amd.tracking import FeatureGrid

    g = FeatureGrid(0)
    rng = np.random.default_rng(3)
    for n in (0, 1, 2, 777, 1500, 4097):
        kps = np.zeros(n, orc.KP64)
        kps["x"] = rng.uniform(T.BOUNDS[0] - 5, T.BOUNDS[2] + 5, n)
        kps["y"] = rng.uniform(T.BOUNDS[1] - 5, T.BOUNDS[3] + 5, n)
        if n > 10:
            kps["x"][5:9] = kps["x"][4]  # several features in one cell
            kps["y"][5:9] = kps["y"][4]
        perm, cs, cols, rows = g.create(T.BOUNDS, kps)
        wperm, wcs, wcols, wrows = orc.feature_grid(kps, T.BOUNDS)
        assert (cols, rows) == (wcols, wrows)
        assert np.array_equal(perm, wperm) and np.array_equal(cs, wcs)
    g.close()


@pytest.mark.parametrize("seed,direction,th,fe", [(1, 0, 15.0, 75), (2, 1, 10.0, 75), (3, 2, 15.0, 75), (4, 0, 30.0, 100), (5, 0, 10.0, 50)])
def test_coarse_parity(orc, matcher, seed, direction, th, fe):
    rng = np.random.default_rng(SEED + seed)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=900, m_pts=1200)
    pts = T.lm_coarse(orc, world)
    n, idx = matcher.SearchByProjectionFrameFrame2(frame, cam, pose, pts, th, fe, direction, ls)
    wn, widx = orc.match_coarse(frame, cam, pose, pts, th, fe, direction, ls)
    assert n == wn and np.array_equal(idx, widx)
    assert n > 50


@pytest.mark.parametrize("seed,th,ratio", [(11, 5.0, 0.8), (12, 4.0, 0.8), (13, 1.0, 0.9), (14, 5.0, 0.6)])
def test_fine_parity(orc, matcher, seed, th, ratio):
    rng = np.random.default_rng(SEED + seed)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=1000, m_pts=3000)
    pts = T.lm_fine(orc, rng, world, pose, ls)
    n, idx, vis, valid = matcher.SearchByProjection2(frame, cam, pose, pts, th, ratio, ls)
    wn, widx, wvis, wvalid = orc.match_fine(frame, cam, pose, pts, th, ratio, ls)
    assert n == wn and np.array_equal(idx, widx)
    assert np.array_equal(vis, wvis) and np.array_equal(valid, wvalid)
    assert n > 100


def test_keyframe_parity(orc, matcher):
    for seed in (21, 22):
        rng = np.random.default_rng(SEED + seed)
        frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=500, m_pts=800)
        skip = (rng.random(len(world["pos"])) < 0.1).astype(np.uint8)
        n, idx = matcher.SearchByProjectionFrameToKeyframe(frame, cam, pose, world["pos"], world["desc"], skip, 15.0, 100)
        wn, widx = orc.match_keyframe(frame, cam, pose, world["pos"], world["desc"], skip, 15.0, 100)
        assert n == wn and np.array_equal(idx, widx) and n > 50


def test_empty_inputs(orc, matcher):
    rng = np.random.default_rng(1)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=50, m_pts=20)
    n, idx = matcher.SearchByProjectionFrameFrame2(frame, cam, pose, np.zeros(0, orc.LM_COARSE), 15.0, 75, 0, ls)
    assert n == 0 and len(idx) == 0
    empty = dict(frame, kps=frame["kps"][:0], desc=frame["desc"][:0], right_points=frame["right_points"][:0],
                 taken=frame["taken"][:0], cell_start=np.zeros_like(frame["cell_start"]))
    n, idx = matcher.SearchByProjectionFrameFrame2(empty, cam, pose, T.lm_coarse(orc, world), 15.0, 75, 0, ls)
    assert n == 0 and (idx == -1).all()


def test_feature_grid_batch_dev(orc):
    import torch
    from snake_slam_amd.tracking import FeatureGrid

    rng = np.random.default_rng(8)
    B, cap = 4, 1100
    ns = np.array([1100, 0, 513, 64], np.int32)
    K = np.zeros((B, cap), orc.KP64)
    D = rng.integers(0, 2**64, size=(B, cap, 4), dtype=np.uint64)
    for b in range(B):
        K["x"][b, : ns[b]] = rng.uniform(T.BOUNDS[0], T.BOUNDS[2], ns[b])
        K["y"][b, : ns[b]] = rng.uniform(T.BOUNDS[1], T.BOUNDS[3], ns[b])
        K["octave"][b, : ns[b]] = rng.integers(0, 4, ns[b])
    cols, rows = orc.grid_dims(T.BOUNDS)
    dev = torch.device("cuda:0")
    t = lambda a: torch.from_numpy(a).to(dev)
    dk, dd, dn = t(K.view(np.uint8).reshape(B, cap, 24)), t(D.view(np.int64)), t(ns)
    ok, od = torch.zeros_like(dk), torch.zeros_like(dd)
    perm = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
    cs = torch.full((B, cols * rows + 1), -1, dtype=torch.int32, device=dev)
    g = FeatureGrid(0)
    torch.cuda.synchronize()
    g.create_batch_dev(T.BOUNDS, dk, dd, dn, ok, od, perm, cs)
    g.sync()
    okh = ok.cpu().numpy().view(orc.KP64).reshape(B, cap)
    odh = od.cpu().numpy().view(np.uint64)
    for b in range(B):
        n = int(ns[b])
        wperm, wcs, _, _ = orc.feature_grid(K[b, :n], T.BOUNDS)
        assert np.array_equal(perm[b, :n].cpu().numpy(), wperm) and np.array_equal(cs[b].cpu().numpy(), wcs)
        wk, wd = np.zeros(n, orc.KP64), np.zeros((n, 4), np.uint64)
        wk[wperm], wd[wperm] = K[b, :n], D[b, :n]
        assert np.array_equal(okh[b, :n], wk) and np.array_equal(odh[b, :n], wd)
    g.close()


# ------------------------------------------------------------------ local-mapping matchers -----
@pytest.fixture(scope="module")
def mapping_matcher():
    from snake_slam_amd.tracking import MappingORBMatcher

    m = MappingORBMatcher()
    yield m
    m.close()


@pytest.mark.parametrize("seed,th,of,fth,masked", [(41, 4.0, 2.0, 50, True), (42, 3.0, 1.5, 60, False), (43, 6.0, 3.0, 40, True),
                                                   (44, 2.5, 1.0, 100, False)])
def test_fuse_parity(orc, mapping_matcher, seed, th, of, fth, masked):
    rng = np.random.default_rng(SEED + seed)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=1000, m_pts=2500)
    pts = T.fusion_points(orc, rng, world, pose, ls)
    mask = (rng.random(len(pts)) > 0.2).astype(np.uint8) if masked else None
    n, cands, idx = mapping_matcher.Fuse(frame, cam, pose, pts, mask, th, of, fth, ls)
    wn, widx = orc.match_fuse(frame, cam, pose, pts, mask, th, of, fth, ls)
    assert n == wn and np.array_equal(idx, widx)
    assert n > 100 and len(cands) == n
    assert cands == [(int(widx[i]), int(pts["id"][i])) for i in np.nonzero(widx >= 0)[0]]


@pytest.mark.parametrize("seed,epi,fd", [(51, 4.0, 50), (52, 2.0, 40), (53, 8.0, 64), (54, 1.0, 30)])
def test_triangulation_project_parity(orc, mapping_matcher, seed, epi, fd):
    rng = np.random.default_rng(SEED + seed)
    c = T.make_triangulation_case(orc, rng, m_pts=1500, n_clutter=600)
    n, pairs, idx = mapping_matcher.SearchForTriangulationProject(c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"],
                                                                  c["desc1"], c["has1"], c["frame2"], c["np2"], c["E"], epi, fd)
    wn, widx = orc.match_triangulation_project(c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"], c["desc1"],
                                               c["has1"], c["frame2"], c["np2"], c["E"], epi, fd)
    assert n == wn and np.array_equal(idx, widx)
    assert n > 20 and pairs == [(int(i), int(widx[i])) for i in np.nonzero(widx >= 0)[0]]


def test_triangulation_project_with_a_bound_frame(orc, mapping_matcher):
    """include/snake_hip.h documents frame2 == NULL = the frame bound with snk_match_bind_frame for this entry point too
    (round 2 rejected it); unbound + NULL must be an error, not a crash."""
    from snake_slam_amd import SnakeHipError

    rng = np.random.default_rng(SEED + 55)
    c = T.make_triangulation_case(orc, rng, m_pts=1200, n_clutter=500)
    args = (c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"], c["desc1"], c["has1"])
    want = mapping_matcher.SearchForTriangulationProject(*args, c["frame2"], c["np2"], c["E"], 4.0, 50)
    mapping_matcher.bind_frame(c["frame2"])
    try:
        got = mapping_matcher.SearchForTriangulationProject(*args, None, c["np2"], c["E"], 4.0, 50)
    finally:
        mapping_matcher.bind_frame(None)
    assert got[0] == want[0] > 20 and got[1] == want[1] and np.array_equal(got[2], want[2])
    with pytest.raises(SnakeHipError):
        mapping_matcher.SearchForTriangulationProject(*args, None, c["np2"], c["E"], 4.0, 50)


def test_mapping_matchers_empty_inputs(orc, mapping_matcher):
    rng = np.random.default_rng(SEED + 60)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=50, m_pts=20)
    n, cands, idx = mapping_matcher.Fuse(frame, cam, pose, np.zeros(0, orc.FUSION_POINT), None, 4.0, 2.0, 50, ls)
    assert n == 0 and cands == [] and len(idx) == 0
    c = T.make_triangulation_case(orc, rng, m_pts=30, n_clutter=10)
    n, pairs, idx = mapping_matcher.SearchForTriangulationProject(c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"][:0],
                                                                  c["np1"][:0], c["desc1"][:0], c["has1"][:0], c["frame2"], c["np2"],
                                                                  c["E"], 4.0, 50)
    assert n == 0 and pairs == []


def _orc_bow(orc, c, epi, fd):
    n, pairs = orc.match_triangulation_bow(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["bow1"], c["np2"], c["desc2"],
                                           c["has2"], c["bow2"], epi, fd)
    return n, [tuple(p) for p in pairs.tolist()]


def _gpu_bow(mm, c, epi, fd):
    return mm.SearchForTriangulation2(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["bow1"], c["np2"], c["desc2"], c["has2"],
                                      c["bow2"], epi, fd)


@pytest.mark.parametrize("seed,epi,fd,nodes", [(71, 4.0, 50, 120), (72, 1.0, 40, 30), (73, 8.0, 64, 400), (74, 4.0, 20, 3)])
def test_triangulation_bow_parity(orc, mapping_matcher, seed, epi, fd, nodes):
    """SearchForTriangulation2 through the C ABI == oracle, pair for pair in the reference's order.  3 nodes: lists far
    longer than the 16 lanes that share an item; 400 nodes: mostly 0-3 features per node."""
    rng = np.random.default_rng(SEED + seed)
    c = T.make_bow_case(rng, m_pts=1400, n_clutter=600, n_nodes=nodes)
    n, pairs = _gpu_bow(mapping_matcher, c, epi, fd)
    wn, wpairs = _orc_bow(orc, c, epi, fd)
    assert n == wn and pairs == wpairs
    assert n > 20


def test_triangulation_bow_ties_and_edges(orc, mapping_matcher):
    rng = np.random.default_rng(SEED + 75)
    c = T.make_bow_case(rng, m_pts=300, n_clutter=100, n_nodes=6)
    # identical descriptors everywhere: every candidate ties, the LAST of the node's list inside the band must win
    c6 = dict(c, desc1=np.repeat(c["desc1"][:1], len(c["desc1"]), 0), desc2=np.repeat(c["desc1"][:1], len(c["desc2"]), 0))
    for epi in (2.0, 1e6):
        n, pairs = _gpu_bow(mapping_matcher, c6, epi, 50)
        wn, wpairs = _orc_bow(orc, c6, epi, 50)
        assert n == wn and pairs == wpairs and n > 0
    empty = (np.zeros(0, np.uint32), np.zeros(1, np.int32), np.zeros(0, np.int32))
    assert _gpu_bow(mapping_matcher, dict(c, bow1=empty), 4.0, 50) == (0, [])
    assert _gpu_bow(mapping_matcher, dict(c, bow2=empty), 4.0, 50) == (0, [])
    ids2, s2, f2 = c["bow2"]
    assert _gpu_bow(mapping_matcher, dict(c, bow2=((ids2 + 1).astype(np.uint32), s2, f2)), 4.0, 50) == (0, [])
    assert _gpu_bow(mapping_matcher, dict(c, has1=np.ones_like(c["has1"])), 4.0, 50) == (0, [])
    assert _gpu_bow(mapping_matcher, dict(c, has2=np.ones_like(c["has2"])), 4.0, 50) == (0, [])
    # malformed feature vectors are refused, not read
    from snake_slam_amd._lib import SnakeHipError
    bad = (c["bow1"][0], c["bow1"][1], c["bow1"][2].copy())
    bad[2][0] = len(c["np1"])
    with pytest.raises(SnakeHipError):
        _gpu_bow(mapping_matcher, dict(c, bow1=bad), 4.0, 50)
    with pytest.raises(SnakeHipError):
        _gpu_bow(mapping_matcher, dict(c, bow2=(c["bow2"][0][::-1].copy(), c["bow2"][1], c["bow2"][2])), 4.0, 50)


@pytest.mark.parametrize("seed,fd,m,clutter", [(81, 50, 1400, 600), (82, 35, 700, 1300), (83, 64, 40, 10)])
def test_triangulation_bf_parity(orc, mapping_matcher, seed, fd, m, clutter):
    rng = np.random.default_rng(SEED + seed)
    c = T.make_bow_case(rng, m_pts=m, n_clutter=clutter)
    n, pairs, idx = mapping_matcher.SearchForTriangulationBF(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["np2"], c["desc2"],
                                                             c["has2"], fd)
    wn, widx = orc.match_triangulation_bf(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["np2"], c["desc2"], c["has2"], fd)
    assert n == wn and np.array_equal(idx, widx)
    assert n > 5 and pairs == [(int(i), int(widx[i])) for i in np.nonzero(widx >= 0)[0]]
    # empty sides
    z2, zd, zh = np.zeros((0, 2)), np.zeros((0, 4), np.uint64), np.zeros(0, np.uint8)
    assert mapping_matcher.SearchForTriangulationBF(c["cam"], c["E"], z2, zd, zh, c["np2"], c["desc2"], c["has2"], fd)[0] == 0
    n, pairs, idx = mapping_matcher.SearchForTriangulationBF(c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], z2, zd, zh, fd)
    assert n == 0 and (idx == -1).all()


@pytest.fixture(scope="module")
def deferred_mapper():
    from snake_slam_amd.tracking import DeferredMapper
    return DeferredMapper()


@pytest.mark.parametrize("seed,n_base", [(91, 1200), (92, 400), (93, 2500)])
def test_relink_parity(orc, deferred_mapper, seed, n_base):
    """DeferredMapper::Relink search through the C ABI == oracle (actions and targets)."""
    rng = np.random.default_rng(SEED + seed)
    frame, cam, pose, qs = T.make_relink_case(orc, rng, n_base=n_base)
    n, action, best = deferred_mapper.RelinkSearch(frame, cam, pose, qs)
    wn, wa, wb = orc.match_relink(frame, cam, pose, qs)
    assert n == wn and np.array_equal(action, wa) and np.array_equal(best, wb)
    assert (wa == 1).sum() > 10 and (wa == 2).sum() > 10 and (wa == 0).sum() > 10


def test_relink_edges(orc, deferred_mapper):
    from snake_slam_amd._lib import SnakeHipError
    rng = np.random.default_rng(SEED + 94)
    frame, cam, pose, qs = T.make_relink_case(orc, rng, n_base=100)
    n, action, best = deferred_mapper.RelinkSearch(frame, cam, pose, qs[:0])
    assert n == 0 and len(action) == 0
    # 17 queries: one full 16-query block plus a ragged one
    n, action, best = deferred_mapper.RelinkSearch(frame, cam, pose, qs[:17])
    wn, wa, wb = orc.match_relink(frame, cam, pose, qs[:17])
    assert n == wn and np.array_equal(action, wa) and np.array_equal(best, wb)
    bad = qs[:3].copy()
    bad["feature"][1] = len(frame["kps"])
    with pytest.raises(SnakeHipError):
        deferred_mapper.RelinkSearch(frame, cam, pose, bad)


def _pack_frames_dev(frames, cap):
    """Host frame dicts (grid order) -> padded device tensors of the batched, device-resident view."""
    import torch

    from snake_slam_amd.tracking import KP64_DTYPE

    B = len(frames)
    ncell = frames[0]["cols"] * frames[0]["rows"] + 1
    kps = np.zeros((B, cap), KP64_DTYPE)
    desc = np.zeros((B, cap, 4), np.uint64)
    rp = np.full((B, cap), -1.0, np.float32)
    taken = np.zeros((B, cap), np.uint8)
    cs = np.zeros((B, ncell), np.int32)
    n = np.zeros(B, np.int32)
    for b, f in enumerate(frames):
        k = len(f["kps"])
        n[b] = k
        kps[b, :k], desc[b, :k], rp[b, :k], taken[b, :k], cs[b] = f["kps"], f["desc"], f["right_points"], f["taken"], f["cell_start"]
    dev = torch.device("cuda", 0)
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(a.shape[0], -1)).to(dev)
    return dict(n=torch.from_numpy(n).to(dev), kps=t(kps).view(B, cap, 24), desc=torch.from_numpy(desc.view(np.int64)).to(dev),
                right_points=torch.from_numpy(rp).to(dev), taken=torch.from_numpy(taken).to(dev), cell_start=torch.from_numpy(cs).to(dev))


def test_coarse_then_fine_batch_dev_parity(orc, matcher):
    """The device-resident, batched forms (frames as the batched front-end leaves them in HBM, poses on the device)
    against the oracle, frame by frame: coarse -> `mvpMapPoints[idx] = mp` applied on the device -> fine, i.e. the
    per-frame chain of TrackingCoarse.cpp:234 / TrackingFine.cpp:149 for a whole batch in three launches per matcher."""
    import torch

    from snake_slam_amd.tracking import LM_COARSE_DTYPE, LM_FINE_DTYPE, frames_dev

    rng = np.random.default_rng(SEED + 77)
    sizes = [(300, 400, 900), (900, 1200, 2500), (0, 50, 40), (600, 1500, 3000), (1000, 700, 10)]
    cases = [T.make_tracking_case(orc, rng, n_clutter=c, m_pts=mp) for c, mp, _ in sizes]
    frames = [c[0] for c in cases]
    cam, ls = cases[0][1], cases[0][3]
    cap = max(len(f["kps"]) for f in frames) + 7
    B = len(cases)
    mc_cap = max(len(c[4]["pos"]) for c in cases) + 3
    coarse = [T.lm_coarse(orc, c[4]) for c in cases]
    fine = [T.lm_fine(orc, rng, c[4], c[2], ls)[: s[2]] for c, s in zip(cases, sizes)]
    mf_cap = max(len(f) for f in fine) + 5
    pc = np.zeros((B, mc_cap), LM_COARSE_DTYPE)
    pf = np.zeros((B, mf_cap), LM_FINE_DTYPE)
    nc, nf = np.zeros(B, np.int32), np.zeros(B, np.int32)
    for b in range(B):
        nc[b], nf[b] = len(coarse[b]), len(fine[b])
        pc[b, : nc[b]], pf[b, : nf[b]] = coarse[b], fine[b]
    dev = torch.device("cuda", 0)
    D = _pack_frames_dev(frames, cap)
    poses = torch.from_numpy(np.stack([c[2] for c in cases])).to(dev)
    d_pc = torch.from_numpy(pc.view(np.uint8).reshape(B, mc_cap, 88)).to(dev)
    d_pf = torch.from_numpy(pf.view(np.uint8).reshape(B, mf_cap, 96)).to(dev)
    d_nc, d_nf = torch.from_numpy(nc).to(dev), torch.from_numpy(nf).to(dev)
    mi_c = torch.full((B, mc_cap), -7, dtype=torch.int32, device=dev)
    mi_f = torch.full((B, mf_cap), -7, dtype=torch.int32, device=dev)
    vis = torch.full((B, mf_cap), 9, dtype=torch.uint8, device=dev)
    n_c = torch.zeros(B, dtype=torch.int32, device=dev)
    n_f = torch.zeros(B, dtype=torch.int32, device=dev)
    fd = frames_dev(T.BOUNDS, D["n"], D["kps"], D["desc"], D["right_points"], D["taken"], D["cell_start"])
    torch.cuda.synchronize()
    matcher.coarse_batch_dev(fd, cam, poses, d_pc, d_nc, 15.0, 75, 0, ls, mi_c, n_c)
    matcher.mark_taken_batch_dev(mi_c, d_nc, D["taken"])
    # the read-only form first (snk_match_project_fine_batch_ro_dev): same matches, the records untouched
    mi_r, vis_r, n_r = torch.full_like(mi_f, -7), torch.full_like(vis, 9), torch.zeros_like(n_f)
    pf_before = d_pf.clone()
    matcher.fine_batch_dev(fd, cam, poses, d_pf, d_nf, 5.0, 0.8, ls, mi_r, vis_r, n_r, write_valid=False)
    matcher.sync()
    assert torch.equal(d_pf, pf_before)
    matcher.fine_batch_dev(fd, cam, poses, d_pf, d_nf, 5.0, 0.8, ls, mi_f, vis, n_f)
    matcher.sync()
    assert torch.equal(mi_r, mi_f) and torch.equal(vis_r, vis) and torch.equal(n_r, n_f)
    assert not torch.equal(d_pf, pf_before)  # the in-place form did clear flags
    mi_c, mi_f, vis, n_c, n_f = mi_c.cpu().numpy(), mi_f.cpu().numpy(), vis.cpu().numpy(), n_c.cpu().numpy(), n_f.cpu().numpy()
    taken_after = D["taken"].cpu().numpy()
    pf_after = d_pf.cpu().numpy().view(LM_FINE_DTYPE).reshape(B, mf_cap)
    total = 0
    for b, (frame, _, pose, _, _, _) in enumerate(cases):
        wn, widx = orc.match_coarse(frame, cam, pose, coarse[b], 15.0, 75, 0, ls)
        assert n_c[b] == wn and np.array_equal(mi_c[b, : nc[b]], widx), b
        assert (mi_c[b, nc[b]:] == -1).all()
        f2 = dict(frame)
        f2["taken"] = frame["taken"].copy()
        f2["taken"][widx[widx >= 0]] = 1
        assert np.array_equal(taken_after[b, : len(frame["kps"])], f2["taken"]), b
        wn, widx, wvis, wvalid = orc.match_fine(f2, cam, pose, fine[b], 5.0, 0.8, ls)
        assert n_f[b] == wn and np.array_equal(mi_f[b, : nf[b]], widx), b
        assert np.array_equal(vis[b, : nf[b]], wvis) and np.array_equal(pf_after[b, : nf[b]]["valid"], wvalid), b
        assert np.array_equal(np.asarray(wvis) != 0, np.asarray(wvalid) != 0), b  # what the read-only form relies on: valid after = visible
        total += int(n_c[b]) + int(n_f[b])
    assert total > 500


@pytest.mark.skipif(os.environ.get("SNK_TRACK_NO_RECURSE") == "1", reason="child run")
@pytest.mark.parametrize("ppw", ["64", "7"])
def test_tracking_matchers_with_many_points_per_wavefront(ppw):
    """The projection matchers pick the number of points per wavefront from the size of the call (1 for a single frame,
    64 for bench-sized batches).  The parity tests above are small calls; here they run again in a child process with the
    value forced (SNK_TRACK_PPW), so that the multi-point path (geometry by lane = point, four 16-lane window scans at a
    time, results through LDS) is compared with the oracle too."""
    import os
    import subprocess
    import sys
    from pathlib import Path

    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, "-m", "pytest", str(root / "tests" / "test_track_gpu.py"), str(root / "tests" / "test_tracking_chain_gpu.py"),
                        "-m", "gpu", "-x", "-q", "-p", "no:cacheprovider", "-k", "coarse or fine or chain or refine"],
                       env=dict(os.environ, SNK_TRACK_PPW=ppw, SNK_TRACK_NO_RECURSE="1"), capture_output=True, text=True, cwd=str(root), timeout=600)
    assert r.returncode == 0, (ppw, r.stdout[-2000:], r.stderr[-1000:])
    assert " passed" in r.stdout


def test_bound_frame_gives_the_same_matches(orc, matcher):
    """snk_match_bind_frame: the frame view is uploaded once, the matchers are called with frame = NULL; between the coarse
    and the fine call only the taken mask is sent again (the adaptor's mvpMapPoints[idx] = mp)."""
    rng = np.random.default_rng(SEED + 505)
    frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=700, m_pts=1100)
    pc = T.lm_coarse(orc, world)
    pf = T.lm_fine(orc, rng, world, pose, ls)
    wn, widx = orc.match_coarse(frame, cam, pose, pc, 15.0, 75, 0, ls)
    f2 = dict(frame)
    f2["taken"] = frame["taken"].copy()
    f2["taken"][widx[widx >= 0]] = 1
    wn2, widx2, wvis, wvalid = orc.match_fine(f2, cam, pose, pf, 5.0, 0.8, ls)
    matcher.bind_frame(frame)
    try:
        n, idx = matcher.SearchByProjectionFrameFrame2(None, cam, pose, pc, 15.0, 75, 0, ls)
        assert n == wn and np.array_equal(idx, widx)
        matcher.bound_taken(f2["taken"])
        n2, idx2, vis, valid = matcher.SearchByProjection2(None, cam, pose, pf, 5.0, 0.8, ls)
        assert n2 == wn2 and np.array_equal(idx2, widx2) and np.array_equal(vis, wvis) and np.array_equal(valid, wvalid)
    finally:
        matcher.bind_frame(None)
    with pytest.raises(Exception):
        matcher.SearchByProjectionFrameFrame2(None, cam, pose, pc, 15.0, 75, 0, ls)  # nothing bound any more
