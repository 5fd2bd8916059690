"""GPU: HIP matchers vs the CPU oracle through the C ABI — bit-exact (integer work)."""
import numpy as np
import pytest

from helpers import SEED, knn_to_array, make_stereo_case, rand_desc

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bf():
    from snake_slam_amd.matcher import BruteForceMatcher

    m = BruteForceMatcher(0)
    yield m
    m.close()


@pytest.fixture(scope="module")
def st():
    from snake_slam_amd.matcher import StereoMatcher

    m = StereoMatcher(0)
    yield m
    m.close()


@pytest.mark.parametrize("nq,nt", [(1, 1), (5, 0), (7, 1), (3, 2), (64, 64), (65, 63), (1000, 1000), (2000, 2000), (1500, 777),
                                   (6000, 9001)])
def test_bf_knn2_parity(bf, orc, nq, nt):
    rng = np.random.default_rng(SEED + nq * 7 + nt)
    q, t = rand_desc(rng, nq), rand_desc(rng, nt)
    if nt > 8:  # ties: duplicates, and a query identical to two train rows
        t[nt // 2] = t[0]
        t[nt - 1] = t[3]
        q[0] = t[0]
        q[nq - 1] = t[3]
    bf.matchKnn2(q, t)
    want = orc.bf_knn2(q, t)
    assert np.array_equal(knn_to_array(bf.knn), knn_to_array(want))
    for th, ratio in [(60, 0.8), (120, 0.9), (60, 0.75), (256, 1.0)]:
        n = bf.filterMatches(th, ratio)
        wp = orc.bf_filter(want, th, ratio)
        assert n == wp.shape[0]
        assert np.array_equal(bf.matches, wp)


def test_bf_knn2_low_entropy_ties(bf, orc):
    """Descriptors drawn from 3 distinct values: almost everything ties; first index must win."""
    rng = np.random.default_rng(5)
    base = rand_desc(rng, 3)
    q = base[rng.integers(0, 3, 300)]
    t = base[rng.integers(0, 3, 500)]
    bf.matchKnn2(q, t)
    assert np.array_equal(knn_to_array(bf.knn), knn_to_array(orc.bf_knn2(q, t)))
    assert bf.filterMatches(60, 0.8) == orc.bf_filter(orc.bf_knn2(q, t), 60, 0.8).shape[0]


def test_bf_extremes(bf, orc):
    z = np.zeros((2, 4), np.uint64)
    o = np.full((2, 4), np.uint64(0xFFFFFFFFFFFFFFFF))
    bf.matchKnn2(z, o)
    got = knn_to_array(bf.knn)
    assert got.tolist() == [[-1, 256, -1, 256], [-1, 256, -1, 256]]  # 256 is 'infinite': never a neighbour
    assert np.array_equal(got, knn_to_array(orc.bf_knn2(z, o)))


def test_bf_matrix_core_extremes(bf, orc):
    """Sizes that take the MFMA kernel with the extreme popcounts: all-zero / all-one descriptors (dot products 0 and
    256, distances 0 and the 'infinite' 256), single-bit descriptors, a train set that is exactly one tile and one
    that is one past a tile."""
    rng = np.random.default_rng(SEED + 404)
    z = np.zeros(4, np.uint64)
    o = np.full(4, np.uint64(0xFFFFFFFFFFFFFFFF))
    one_bit = np.zeros((256, 4), np.uint64)
    for b in range(256):
        one_bit[b, b >> 6] = np.uint64(1) << np.uint64(b & 63)
    for nt in (32, 33, 64, 97):
        q = np.stack([z] * 20 + [o] * 20 + list(one_bit[:40]) + list(rand_desc(rng, 17)))
        t = np.stack(([o] * 10 + [z] * 5 + list(one_bit[100:110]) + list(rand_desc(rng, 200)))[:nt])
        bf.matchKnn2(q, t)
        assert np.array_equal(knn_to_array(bf.knn), knn_to_array(orc.bf_knn2(q, t))), nt
    # every pair at distance 256: nothing is a neighbour
    bf.matchKnn2(np.stack([z] * 40), np.stack([o] * 40))
    got = knn_to_array(bf.knn)
    assert (got[:, 0] == -1).all() and (got[:, 1] == 256).all() and (got[:, 2] == -1).all() and (got[:, 3] == 256).all()
    # every bit position on its own: the bit -> operand-byte mapping must be the same on both sides
    bf.matchKnn2(one_bit, one_bit)
    got = knn_to_array(bf.knn)
    assert (got[:, 0] == np.arange(256)).all() and (got[:, 1] == 0).all() and (got[:, 3] == 2).all()
    assert np.array_equal(got, knn_to_array(orc.bf_knn2(one_bit, one_bit)))


def test_bf_batch_dev_parity(bf, orc):
    import torch

    rng = np.random.default_rng(SEED)
    B, capq, capt = 24, 1000, 1000  # B*capq >= 16384 -> exercises the 4-queries-per-wave kernel
    nq = rng.integers(0, capq + 1, B).astype(np.int32)
    nt = rng.integers(0, capt + 1, B).astype(np.int32)
    nq[0], nt[0] = capq, capt
    nq[1], nt[1] = 0, 5
    nq[2], nt[2] = 9, 0
    q = rand_desc(rng, B * capq).reshape(B, capq, 4)
    t = rand_desc(rng, B * capt).reshape(B, capt, 4)
    dev = torch.device("cuda:0")
    qd = torch.from_numpy(q.view(np.int64)).to(dev)
    td = torch.from_numpy(t.view(np.int64)).to(dev)
    nqd = torch.from_numpy(nq).to(dev)
    ntd = torch.from_numpy(nt).to(dev)
    out = torch.full((B, capq, 4), -7, dtype=torch.int32, device=dev)
    pairs = torch.full((B, capq, 2), -7, dtype=torch.int32, device=dev)
    npairs = torch.zeros(B, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    bf.knn2_batch_dev(qd, nqd, td, ntd, out)
    bf.filter_batch_dev(out, nqd, 60, 0.8, pairs, npairs)
    bf.sync()
    out_h, pairs_h, np_h = out.cpu().numpy(), pairs.cpu().numpy(), npairs.cpu().numpy()
    for b in range(B):
        want = orc.bf_knn2(q[b, : nq[b]], t[b, : nt[b]])
        assert np.array_equal(out_h[b, : nq[b]], knn_to_array(want)), f"batch {b}"
        assert (out_h[b, nq[b]:] == -7).all()
        wp = orc.bf_filter(want, 60, 0.8)
        assert np_h[b] == wp.shape[0]
        assert np.array_equal(pairs_h[b, : np_h[b]], wp)


@pytest.mark.parametrize("nl,nr,relaxed", [(1000, 1000, True), (1000, 1000, False), (257, 63, True), (5, 2000, True), (300, 1, True),
                                           (500, 8192, True), (400, 9000, True)])  # 9000 > the LDS row index: unindexed kernel
def test_stereo_parity(st, orc, nl, nr, relaxed):
    rng = np.random.default_rng(SEED + nl + nr)
    left, dl, right, dr, bfv, ls = make_stereo_case(rng, nl, nr)
    n, rp, dp = st.StereoMatching(left, dl, right, dr, bfv, ls, relaxed)
    n2, rp2, dp2 = orc.stereo_match(left, dl, right, dr, bfv, ls, relaxed)
    assert n == n2
    assert np.array_equal(rp, rp2) and np.array_equal(dp, dp2)
    if nl >= 257 and nr >= 63:
        assert n > 0


def test_stereo_empty(st):
    from oracle.oracle import KP64

    ls = np.ones(4, np.float32)
    e = np.zeros(0, KP64)
    d = np.zeros((0, 4), np.uint64)
    n, rp, dp = st.StereoMatching(e, d, e, d, 47.9, ls)
    assert n == 0 and rp.size == 0


@pytest.mark.parametrize("B,relaxed", [(6, True), (10, True), (10, False), (9, True)])
def test_stereo_batch_dev_parity(st, orc, B, relaxed):
    """B < 8: row-index sort + 16-lane kernel; B >= 8: one workgroup per frame with the right side in LDS."""
    import torch

    rng = np.random.default_rng(SEED + 99 + B)
    capl, capr = 700, 650
    dev = torch.device("cuda:0")
    from oracle.oracle import KP64

    L = np.zeros((B, capl), KP64)
    R = np.zeros((B, capr), KP64)
    DL = np.zeros((B, capl, 4), np.uint64)
    DR = np.zeros((B, capr, 4), np.uint64)
    nl = np.array([700, 0, 123, 699, 64, 1, 700, 333, 17, 650][:B], np.int32)
    nr = np.array([650, 10, 0, 1, 650, 333, 649, 2, 650, 65][:B], np.int32)
    cases = []
    for b in range(B):
        if nl[b] and nr[b]:
            l, dl, r, dr, bfv, ls = make_stereo_case(rng, int(nl[b]), int(nr[b]))
            L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]] = l, dl, r, dr
        cases.append(b)
    ls = (np.float32(1.2) ** np.arange(4)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    Ld, Rd = t(L.view(np.uint8).reshape(B, capl, 24)), t(R.view(np.uint8).reshape(B, capr, 24))
    DLd, DRd = t(DL.view(np.int64)), t(DR.view(np.int64))
    nld, nrd = t(nl), t(nr)
    rp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
    dp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
    nm = torch.full((B,), -1, dtype=torch.int32, device=dev)
    torch.cuda.synchronize()
    st.match_batch_dev(Ld, DLd, nld, Rd, DRd, nrd, 47.9, ls, relaxed, rp, dp, nm)
    st.sync()
    rp_h, dp_h, nm_h = rp.cpu().numpy(), dp.cpu().numpy(), nm.cpu().numpy()
    for b in range(B):
        n2, rp2, dp2 = orc.stereo_match(L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]], 47.9, ls, relaxed)
        assert nm_h[b] == n2, f"batch {b}"
        assert np.array_equal(rp_h[b, : nl[b]], rp2) and np.array_equal(dp_h[b, : nl[b]], dp2)
        assert (rp_h[b, nl[b]:] == -1000).all()


def test_invalid_args_return_status(bf):
    from snake_slam_amd import _lib
    import ctypes as C

    lib = _lib.load()
    rc = lib.snk_bf_knn2(None, None, 1, None, 1, None)
    assert rc == 1
    assert b"NULL" in lib.snk_last_error() or b"invalid" in lib.snk_last_error()
    h = C.c_void_p()
    assert lib.snk_matcher_create(9999, None, C.byref(h)) != 0


DEFS = [("bf_filter.threshold_strict", 1), ("bf_filter.ratio_strict", 1), ("iround.mode", 1), ("iround.mode", 2)]


@pytest.fixture
def definitions(orc):
    """Sets a [DEFINED] switch in the library AND in the oracle; everything is back at its default afterwards."""
    from snake_slam_amd import _lib

    def set_both(key, value):
        _lib.set_definition(key, value)
        orc.set_definition(key, value)

    yield set_both
    for key in _lib.DEFINITIONS:
        _lib.set_definition(key, 0)
        orc.set_definition(key, 0)


@pytest.mark.parametrize("key,value", DEFS)
def test_definition_switches_parity(bf, st, orc, definitions, key, value):
    """snk_set_definition: every [DEFINED] comparison / rounding rule can be flipped at run time, the oracle mirrors it, and the
    two still agree bit for bit -- on inputs built so that the rule DECIDES something (the result differs from the default's)."""
    rng = np.random.default_rng(SEED + 9000 + value)
    # kNN-2 table with exact ties on both filter rules: dist1 == threshold and dist1 == ratio * dist2 (ratio 0.5, representable)
    q, t = rand_desc(rng, 400), rand_desc(rng, 300)
    bf.matchKnn2(q, t)
    knn = bf.knn.copy()
    th = int(np.sort(knn["dist1"])[len(knn) // 3])           # a threshold that some dist1 equal exactly
    knn["dist2"][::3] = 2 * knn["dist1"][::3]                  # every third row sits exactly on the ratio 0.5
    # stereo case in which the rounding rule decides band membership: right rows exactly on .5, left rows on integers (which no
    # rule moves).  floor(x + 0.5) and half-to-even differ on positive .5 rows with an even floor; floor(x + 0.5) and
    # half-away-from-zero only differ on NEGATIVE .5 rows (rectification can push points above the image), so for that rule the
    # whole case sits above the image
    left, dl, right, dr, bfv, ls = make_stereo_case(rng, 300, 280)
    right["y"] = np.floor(right["y"]) + 0.5
    left["y"] = np.floor(left["y"])
    if key == "iround.mode" and value == 1:
        right["y"] -= 600.0
        left["y"] -= 600.0

    def run():
        bf.knn = knn.copy()
        n = bf.filterMatches(th, 0.5)
        got_pairs = bf.matches.copy()
        want_pairs = orc.bf_filter(knn, th, 0.5)
        ns, rp, dp = st.StereoMatching(left, dl, right, dr, bfv, ls, True)
        wn, wrp, wdp = orc.stereo_match(left, dl, right, dr, bfv, ls, True)
        assert n == len(want_pairs) and np.array_equal(got_pairs, want_pairs)
        assert ns == wn and np.array_equal(rp, wrp) and np.array_equal(dp, wdp)
        return got_pairs, rp

    base_pairs, base_rp = run()
    definitions(key, value)
    pairs, rp = run()
    if key.startswith("bf_filter"):
        assert len(pairs) < len(base_pairs)      # the strict operator drops the rows that sit exactly on the bound
    else:
        assert not np.array_equal(rp, base_rp)   # another rounding of the .5 rows moves the row bands


@pytest.mark.parametrize("B", [4, 9])
@pytest.mark.parametrize("mode", [1, 2])
def test_definition_switches_parity_batched_stereo(st, orc, definitions, B, mode):
    """The batched StereoMatching entry point under the rounding definitions (stereo_frame_kernel -- the B >= 8 kernel
    -- bucketed right keypoints with floor(y + 0.5) whatever "iround.mode" said, while the row it compared against followed the
    mode).  Right rows exactly on .5, left rows on integers; mode 1 (half away from zero) only differs for negative rows, so that
    case sits above the image.  B = 4: sort + 16-lane kernel; B = 9: one workgroup per frame."""
    import torch
    from oracle.oracle import KP64

    rng = np.random.default_rng(SEED + 9100 + 10 * B + mode)
    capl, capr = 400, 380
    dev = torch.device("cuda:0")
    L, R = np.zeros((B, capl), KP64), np.zeros((B, capr), KP64)
    DL, DR = np.zeros((B, capl, 4), np.uint64), np.zeros((B, capr, 4), np.uint64)
    nl = rng.integers(200, capl + 1, B).astype(np.int32)
    nr = rng.integers(200, capr + 1, B).astype(np.int32)
    for b in range(B):
        l, dl, r, dr, bfv, ls = make_stereo_case(rng, int(nl[b]), int(nr[b]))
        r["y"] = np.floor(r["y"]) + 0.5
        l["y"] = np.floor(l["y"])
        if mode == 1:
            r["y"] -= 600.0
            l["y"] -= 600.0
        L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]] = l, dl, r, dr
    ls = (np.float32(1.2) ** np.arange(4)).astype(np.float32)
    t = lambda a: torch.from_numpy(a).to(dev)
    Ld, Rd = t(L.view(np.uint8).reshape(B, capl, 24)), t(R.view(np.uint8).reshape(B, capr, 24))
    DLd, DRd, nld, nrd = t(DL.view(np.int64)), t(DR.view(np.int64)), t(nl), t(nr)

    def run():
        rp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
        dp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
        nm = torch.full((B,), -1, dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        st.match_batch_dev(Ld, DLd, nld, Rd, DRd, nrd, 47.9, ls, True, rp, dp, nm)
        st.sync()
        rp_h, dp_h, nm_h = rp.cpu().numpy(), dp.cpu().numpy(), nm.cpu().numpy()
        for b in range(B):
            n2, rp2, dp2 = orc.stereo_match(L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]], 47.9, ls, True)
            assert nm_h[b] == n2, f"batch {b}"
            assert np.array_equal(rp_h[b, : nl[b]], rp2) and np.array_equal(dp_h[b, : nl[b]], dp2)
        return rp_h

    base = run()
    definitions("iround.mode", mode)
    assert not np.array_equal(run(), base)   # the rule decided something
