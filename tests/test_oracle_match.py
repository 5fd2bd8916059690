"""CPU: pin the matching oracle against known answers and an independent numpy restatement.

The reference holds no golden vectors for this path — parity is unpinned; these
tests pin the oracle to the written definition instead."""
import numpy as np
import pytest

from helpers import SEED, knn_to_array, make_stereo_case, np_hamming_matrix, np_knn2, rand_desc


def test_hamming_known_answers(orc):
    z = np.zeros(4, np.uint64)
    o = np.full(4, np.uint64(0xFFFFFFFFFFFFFFFF))
    assert orc.hamming(z, z) == 0
    assert orc.hamming(z, o) == 256
    for b in (0, 1, 63, 64, 127, 200, 255):
        s = z.copy()
        s[b >> 6] = np.uint64(1) << np.uint64(b & 63)
        assert orc.hamming(z, s) == 1
        assert orc.hamming(o, s) == 255
    rng = np.random.default_rng(SEED)
    a, b = rand_desc(rng, 50), rand_desc(rng, 50)
    m = np_hamming_matrix(a, b)
    for i in range(50):
        assert orc.hamming(a[i], b[i]) == m[i, i]


def test_bf_knn2_matches_numpy_definition(orc):
    rng = np.random.default_rng(SEED + 1)
    for nq, nt in [(1, 1), (3, 2), (37, 64), (100, 129), (64, 1000)]:
        q, t = rand_desc(rng, nq), rand_desc(rng, nt)
        # force ties: duplicate train rows and low-entropy descriptors
        if nt > 4:
            t[nt // 2] = t[0]
            t[nt - 1] = t[1]
            q[0] = t[0]
        got = knn_to_array(orc.bf_knn2(q, t))
        assert np.array_equal(got, np_knn2(q, t))


def test_bf_knn2_ties_first_index_wins(orc):
    t = np.zeros((5, 4), np.uint64)
    q = np.zeros((1, 4), np.uint64)
    got = knn_to_array(orc.bf_knn2(q, t))
    assert got.tolist() == [[0, 0, 1, 0]]
    t[0, 0] = 1  # idx0 at distance 1, others 0
    got = knn_to_array(orc.bf_knn2(q, t))
    assert got.tolist() == [[1, 0, 2, 0]]


def test_bf_knn2_empty_and_single_train(orc):
    rng = np.random.default_rng(3)
    q = rand_desc(rng, 4)
    got = knn_to_array(orc.bf_knn2(q, np.zeros((0, 4), np.uint64)))
    assert (got == np.array([-1, 256, -1, 256])).all()
    got = knn_to_array(orc.bf_knn2(q, q[:1]))
    assert (got[:, 0] == 0).all() and (got[:, 2] == -1).all() and (got[:, 3] == 256).all()
    assert got[0, 1] == 0


def test_bf_filter_definition(orc):
    from oracle.oracle import KNN2

    knn = np.zeros(6, KNN2)
    knn["idx1"] = [5, 4, 3, -1, 1, 0]
    knn["dist1"] = [60, 61, 40, 256, 48, 0]
    knn["dist2"] = [75, 100, 50, 256, 60, 0]
    knn["idx2"] = [1, 1, 1, -1, 2, 3]
    pairs = orc.bf_filter(knn, 60, 0.8)
    # 60<=60 and 60<=0.8*75=60 keep; 61>60 drop; 40<=40 keep; none drop; 48<=48 keep; 0<=0 keep
    assert pairs.tolist() == [[0, 5], [2, 3], [4, 1], [5, 0]]


def test_stereo_hand_checked(orc):
    from oracle.oracle import KP64

    ls = np.array([1.0, 1.2, 1.44, 1.728], np.float32)
    left = np.zeros(1, KP64)
    left["x"], left["y"], left["angle"], left["octave"] = 100.0, 50.2, 10.0, 0
    dl = np.zeros((1, 4), np.uint64)
    right = np.zeros(4, KP64)
    right["x"] = [90.0, 95.0, 101.0, 70.0]
    right["y"] = [50.0, 52.4, 50.0, 53.0]  # rows 50, 52, 50, 53; band is 50 +- 2
    right["angle"] = [12.0, 10.0, 10.0, 10.0]
    right["octave"] = [0, 1, 0, 0]
    dr = np.zeros((4, 4), np.uint64)
    dr[0, 0] = 0b111        # dist 3
    dr[1, 0] = 0b1          # dist 1 -> best
    dr[2, 0] = 0            # dist 0 but disparity -1 < 0 -> gated out
    dr[3, 0] = 0            # row 53 outside band
    n, rp, dp = orc.stereo_match(left, dl, right, dr, 47.9, ls, relaxed=True)
    assert n == 1
    assert rp[0] == np.float32(95.0)
    assert dp[0] == np.float32(47.9 / 5.0)
    # strict mode: ratio 0.7 -> 1 > 0.7*3 false -> keep; angle diff 0 ok
    n, rp, dp = orc.stereo_match(left, dl, right, dr, 47.9, ls, relaxed=False)
    assert n == 1
    # ratio rejection: make second best equal
    dr[0, 0] = 0b1
    n, rp, dp = orc.stereo_match(left, dl, right, dr, 47.9, ls, relaxed=True)
    assert n == 0 and rp[0] == np.float32(-1000)


def test_stereo_365_wrap_quirk(orc):
    """The reference wraps angles with 365, not 360 (Preprocess.cpp:216-217): 2 vs 358 degrees is
    |2+365-358| = 9 <= 25 (accepted), and 0 vs 340 is 25 -> accepted although the true
    difference is 20."""
    from oracle.oracle import KP64

    ls = np.array([1.0], np.float32)
    left = np.zeros(1, KP64)
    left["x"], left["y"], left["angle"] = 100.0, 50.0, 0.0
    right = np.zeros(1, KP64)
    right["x"], right["y"], right["angle"] = 90.0, 50.0, 339.0
    d = np.zeros((1, 4), np.uint64)
    n, _, _ = orc.stereo_match(left, d, right, d, 47.9, ls, relaxed=True)
    assert n == 0  # |0+365-339| = 26 > 25
    right["angle"] = 340.0
    n, _, _ = orc.stereo_match(left, d, right, d, 47.9, ls, relaxed=True)
    assert n == 1


def _stereo_numpy(left, dl, right, dr, bf, ls, relaxed):
    """Independent restatement via the lexicographic-min formulation used by the kernel."""
    nl = left.shape[0]
    rp = np.full(nl, -1000.0, np.float32)
    dp = np.full(nl, -1000.0, np.float32)
    n = 0
    if right.shape[0] == 0:
        return n, rp, dp
    D = np_hamming_matrix(dl, dr)
    yr = np.floor(right["y"] + 0.5).astype(np.int64)
    maxd = np.float32(bf * 0.5)
    for i in range(nl):
        y = int(np.floor(left["y"][i] + 0.5))
        r = int(np.ceil(np.float32(2.0) * ls[left["octave"][i]]))
        disp = left["x"][i] - right["x"]
        ok = (np.abs(yr - y) <= r) & (disp >= 0) & (disp <= float(maxd)) & (np.abs(left["octave"][i] - right["octave"]) <= 1)
        ok &= D[i] < 250
        idx = np.nonzero(ok)[0]
        best, second, bid = 250, 250, -1
        if idx.size:
            keys = sorted((int(D[i, j]), int(yr[j]), int(j)) for j in idx)
            best, bid = keys[0][0], keys[0][2]
            if len(keys) > 1:
                second = keys[1][0]
        if best > (75 if relaxed else 40):
            continue
        if best > (0.9 if relaxed else 0.7) * second:
            continue
        a1, a2 = np.float32(left["angle"][i]), np.float32(right["angle"][bid])
        rot = min(abs(a1 - a2), abs((a1 + np.float32(365)) - a2), abs(a1 - (a2 + np.float32(365))))
        if rot > (25 if relaxed else 5):
            continue
        rpt = right["x"][bid]
        d = left["x"][i] - rpt
        if d <= 0.001:
            d = 0.001
            rpt = left["x"][i] - d
        rp[i] = np.float32(rpt)
        dp[i] = np.float32(bf / d)
        n += 1
    return n, rp, dp


def test_stereo_matches_independent_restatement(orc):
    rng = np.random.default_rng(SEED + 2)
    for nl, nr, relaxed in [(200, 180, True), (300, 330, False), (64, 1, True)]:
        left, dl, right, dr, bf, ls = make_stereo_case(rng, nl, nr)
        n, rp, dp = orc.stereo_match(left, dl, right, dr, bf, ls, relaxed)
        n2, rp2, dp2 = _stereo_numpy(left, dl, right, dr, bf, ls, relaxed)
        assert n == n2 and n > 0 or nr == 1
        assert np.array_equal(rp, rp2) and np.array_equal(dp, dp2)


def test_definition_switches_of_the_oracle(orc):
    """orc_set_definition mirrors snk_set_definition (include/snake_hip.h): filterMatches operator strictness and the
    iRound rule.  Checked on hand-built rows where each rule decides."""
    knn = np.zeros(4, orc.KNN2)
    knn["idx1"], knn["idx2"] = [0, 1, 2, 3], [1, 2, 3, 0]
    knn["dist1"], knn["dist2"] = [60, 59, 40, 40], [200, 200, 80, 81]   # row 0 on the threshold, row 2 on the ratio 0.5
    try:
        assert orc.bf_filter(knn, 60, 0.5)[:, 0].tolist() == [0, 1, 2, 3]
        orc.set_definition("bf_filter.threshold_strict", 1)
        assert orc.bf_filter(knn, 60, 0.5)[:, 0].tolist() == [1, 2, 3]
        orc.set_definition("bf_filter.ratio_strict", 1)
        assert orc.bf_filter(knn, 60, 0.5)[:, 0].tolist() == [1, 3]
        with pytest.raises(ValueError):
            orc.set_definition("iround.mode", 3)
        with pytest.raises(ValueError):
            orc.set_definition("no.such.key", 0)
        # iRound: one left keypoint at y = 10.5 (r = 2 at octave 0), right keypoints at y = 7.5 / 8.5 / 12.5 / 13.5, same descriptor
        left = np.zeros(1, orc.KP64)
        left["x"], left["y"] = 100.0, 10.5
        d = np.full((1, 4), np.uint64(0x0123456789ABCDEF))
        ls = np.array([1.0], np.float32)

        def hit(ry):
            right = np.zeros(1, orc.KP64)
            right["x"], right["y"] = 90.0, ry
            return orc.stereo_match(left, d, right, d, 100.0, ls, True)[0]

        # floor(x + 0.5): rows 11 vs 8 / 9 / 13 / 14 -> band 9..13
        orc.set_definition("iround.mode", 0)
        assert [hit(7.5), hit(8.5), hit(12.5), hit(13.5)] == [0, 1, 1, 0]
        # half to even: 10.5 -> 10, band 8..12; 7.5 -> 8, 8.5 -> 8, 12.5 -> 12, 13.5 -> 14
        orc.set_definition("iround.mode", 2)
        assert [hit(7.5), hit(8.5), hit(12.5), hit(13.5)] == [1, 1, 1, 0]
        # half away from zero equals mode 0 for positive rows and differs for negative ones: -10.5 -> -11 (mode 1) / -10 (mode 0)
        left["y"] = -10.5
        orc.set_definition("iround.mode", 0)
        a = [hit(-13.5), hit(-7.5)]      # rows -13 / -7 against band -12..-8
        orc.set_definition("iround.mode", 1)
        b = [hit(-13.5), hit(-7.5)]      # rows -14 / -8 against band -13..-9
        assert a == [0, 0] and b == [0, 0]
        a2 = (orc.set_definition("iround.mode", 0), hit(-12.5))[1]   # row -12 in band -12..-8
        b2 = (orc.set_definition("iround.mode", 1), hit(-12.5))[1]   # row -13 in band -13..-9
        assert a2 == 1 and b2 == 1
        orc.set_definition("iround.mode", 0)
        c0 = hit(-8.5)   # row -8, band -12..-8 -> in
        orc.set_definition("iround.mode", 1)
        c1 = hit(-8.5)   # row -9, band -13..-9 -> in
        assert c0 == 1 and c1 == 1
    finally:
        for k in ("bf_filter.threshold_strict", "bf_filter.ratio_strict", "iround.mode"):
            orc.set_definition(k, 0)
