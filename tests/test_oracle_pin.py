"""CPU: the ORB oracle against answers produced by an independent implementation (scikit-image 0.18.3), stored in
tests/golden/skimage_pin.npz by tests/golden/pin_against_skimage.py (run in the build container, where skimage exists).
Pins the BRIEF pattern table (skimage's copy of OpenCV's bit_pattern_31) and the FAST-9/16 segment test; it cannot pin
the saiga-side choices (pyramid, distribution, blur), see DESIGN.md section 2."""
import re
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
G = ROOT / "tests" / "golden"
PIN = np.load(G / "skimage_pin.npz")


def _inc_table(path):
    txt = re.sub(r"/\*.*?\*/", "", Path(path).read_text(), flags=re.S)
    return np.array([int(v) for v in re.findall(r"-?\d+", txt)], np.int8).reshape(256, 4)


def test_brief_pattern_is_opencv_bit_pattern_31(orc):
    want = PIN["pattern"]
    assert want.shape == (256, 4)
    assert np.array_equal(np.asarray(orc.brief_pattern()).reshape(256, 4).astype(np.int8), want)
    # the oracle's table and the kernels' table are separate files: both must be the pinned one
    assert np.array_equal(_inc_table(ROOT / "oracle" / "brief_pattern.inc"), want)
    assert np.array_equal(_inc_table(ROOT / "snake_slam_amd" / "csrc" / "brief_pattern_31.inc"), want)


def test_fast9_segment_test_matches_skimage_corner_fast(orc):
    nfeat, sf, nl, ini, mn = PIN["params"]
    levels, _ = orc.pyramid(orc.orb_params(int(nfeat), float(sf), int(nl), int(ini), int(mn)), PIN["img"])
    total = 0
    for l, lv in enumerate(levels):
        h, w = lv.shape
        S = np.zeros((h, w), np.int32)
        for y in range(3, h - 3):
            for x in range(3, w - 3):
                S[y, x] = orc.fast_score(lv, x, y)
        for t in (int(ini), int(mn)):
            ys, xs = np.nonzero(S > t)
            mine = np.stack([xs, ys], 1).astype(np.int16)
            assert np.array_equal(mine, PIN[f"fast_l{l}_t{t}"]), (l, t)
            total += len(mine)
    assert total > 1000


def test_ba_optimum_against_scipy(orc):
    """An independent pin of "snk-ba v1" (VERDICT round 3, item 6): the minimiser of the robust cost found by
    scipy.optimize.least_squares (tests/ba_scipy.py: the cost written from its definition, rotation-vector parameterisation,
    complex-step Jacobian, trust-region steps -- nothing shared with the oracle) against the oracle's LM run to convergence.  Same
    cost to 1e-12 relative, same poses and points to 1e-8: observation model, weights, Huber on the residual NORM, the stereo
    residual, the handling of constant cameras / points and the gauge all have to agree for that."""
    import ba_scipy
    from scipy.spatial.transform import Rotation
    from snake_slam_amd import synth

    kw = dict(max_iterations=60, max_pcg_iterations=2000, pcg_tol=1e-14)
    sc, _ = synth.ba_scene(n_kf=6, n_pt=120, obs_per_pt=4, seed=61, outlier_frac=0.05)
    sc2, _ = synth.ba_scene(n_kf=5, n_pt=90, obs_per_pt=3, seed=62, stereo_frac=0.0, n_fixed=2, outlier_frac=0.03)  # mono only, two fixed cameras
    sc2["pt_const"][:10] = 1
    for s in (sc, sc2):
        R, t, pt, cost, _ = ba_scipy.optimum(s)
        wpose, wpt, _, cf, _ = orc.ba_solve(s, orc.ba_options(**kw))
        assert abs(cf - cost) <= 1e-12 * cost, (cf, cost)
        assert np.abs(Rotation.from_quat(wpose[:, :4]).as_matrix() - R).max() <= 1e-8
        assert np.abs(wpose[:, 4:] - t).max() <= 1e-8 and np.abs(wpt - pt).max() <= 2e-8
        # the Huber branch was active for some observations and not for others: the pin covers both
        chi = orc.ba_chi2(dict(s, pose=wpose, pt=wpt), None)
        th = np.where(s["obs_depth"] > 0, 2.3**2, 2.1**2)
        assert (chi > th).sum() >= 3 and (chi <= th).sum() > 100


def test_ic_moments_and_angle_against_skimage():
    """(iii) of tests/golden/pin_against_skimage2.py: the oracle's integer moments over the radius-15 disc equal the sums over
    scikit-image's OFAST_MASK, and its angle (OpenCV's fastAtan2 polynomial, degrees in [0, 360)) is within 0.02 degrees of
    skimage.feature.corner_orientations' atan2."""
    from oracle import oracle as orc

    orc.build()
    g = np.load(G / "skimage_pin2.npz")
    img = np.load(G / "skimage_pin.npz")["img"]
    worst = 0.0
    for x, y, m10, m01, rad in zip(g["xs"], g["ys"], g["m10"], g["m01"], g["orientation_rad"]):
        a, b = orc.ic_moments(img, int(x), int(y))
        assert (a, b) == (int(m10), int(m01)), (x, y)
        deg = float(orc.fast_atan2(float(b), float(a)))
        want = np.rad2deg(rad) % 360.0
        worst = max(worst, min(abs(deg - want), 360.0 - abs(deg - want)))
    assert worst <= 0.02, worst
    umax = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]  # ORB-SLAM2's table; skimage's mask is the same disc
    assert g["mask"].sum() == sum(2 * u + 1 for u in umax) + sum(2 * u + 1 for u in umax[1:]) and len(g["xs"]) >= 64


def test_brief_steering_against_skimage():
    """(iv): the 256 steered tests on an UNBLURRED plane with forced angles against scikit-image's _orb_loop -- pattern order,
    rotation direction, the roles of x and y and the bit order.  Bits may differ only where a steered coordinate of the pair sits on
    a rounding boundary (float sincos + half-to-even here, double + half-away-from-zero there); every difference is checked to be
    such a case, and they are rare."""
    from oracle import oracle as orc

    orc.build()
    g = np.load(G / "skimage_pin2.npz")
    img = np.load(G / "skimage_pin.npz")["img"]
    pat = np.asarray(orc.brief_pattern()).reshape(256, 4).astype(np.float64)
    bits = np.unpackbits(g["bits"], axis=-1).astype(bool)
    n_diff = n_all = 0
    for a, deg in enumerate(g["angles_deg"]):
        s, c = np.sin(np.deg2rad(deg)), np.cos(np.deg2rad(deg))
        for k, (x, y) in enumerate(zip(g["xs"], g["ys"])):
            d = orc.descriptor_on_plane(img, int(x), int(y), float(deg))
            mine = np.array([(int(d[b >> 6]) >> (b & 63)) & 1 for b in range(256)], bool)
            diff = np.nonzero(mine != bits[a, k])[0]
            n_all += 256
            n_diff += len(diff)
            for b in diff:
                # the four steered coordinates of pair b in double precision: one of them within 1e-3 of a half-integer
                co = [c * pat[b, 0] - s * pat[b, 1], s * pat[b, 0] + c * pat[b, 1], c * pat[b, 2] - s * pat[b, 3], s * pat[b, 2] + c * pat[b, 3]]
                assert min(abs(abs(v - np.floor(v)) - 0.5) for v in co) < 1e-3, (deg, k, b, co)
    assert n_all == 9 * len(g["xs"]) * 256 and n_diff <= n_all // 2000, (n_diff, n_all)


def test_harris_response_against_skimage_derivatives(orc):
    """Round 5 ("orb.response" = 1): the oracle's Harris response against tests/golden/skimage_pin3.npz -- the 7 x 7 box sums of
    skimage's Sobel derivative products (exact integers) and det - 0.04 trace^2 on them in double (pin_against_skimage3.py).  The three
    sums must be equal, the float response equal to float rounding, the rank map order-preserving; skimage's own corner_harris (a
    Gaussian window) must rank the FAST corners among the points almost identically."""
    P = np.load(G / "skimage_pin3.npz")
    img = np.ascontiguousarray(PIN["img"])
    resp = np.zeros(len(P["xs"]), np.float32)
    for i, (x, y) in enumerate(zip(P["xs"], P["ys"])):
        assert orc.harris_abc(img, x, y) == (int(P["a"][i]), int(P["b"][i]), int(P["c"][i])), (x, y)
        resp[i] = orc.harris_response(img, x, y)
    want = P["response"]
    # float rounding of the three terms before their (cancelling) difference: 1e-6 of their magnitudes
    A, B, Cc = (P[k].astype(np.float64) for k in "abc")
    mag = (A * B + Cc * Cc + 0.04 * (A + B) ** 2) * (1.0 / 7140.0) ** 4
    assert np.all(np.abs(resp - want) <= 1e-6 * mag)
    assert (resp < 0).any() and (resp > 0).any()  # edges and corners
    order = np.argsort(resp, kind="stable")
    ranks = np.array([orc.harris_rank(r) for r in resp], np.uint64)
    assert np.all(np.diff(ranks[order].astype(np.int64)) >= 0), "harris_rank must preserve the order of the float response"
    assert float(P["spearman"]) > 0.8


def test_blur_and_pyramid_against_scipy_ndimage(orc):
    """Two more [DEFINED] steps of "snk-orb v1" against library code that shares nothing with the oracle (scipy.ndimage, installed
    here -- no fixture needed):
    * the 7 x 7 blur = the separable integer kernel {18, 33, 49, 56, 49, 33, 18} applied to rows then columns in exact integers, ONE
      rounding (acc + 2^15) >> 16, reflect-101 borders = scipy.ndimage.correlate1d(mode='mirror') on int64 -- every pixel equal;
    * the pyramid down-scale = bilinear at source coordinate (d + 0.5) * (src / dst) - 0.5: scipy.ndimage.map_coordinates(order=1,
      mode='nearest') evaluates the same interpolant in double; the oracle rounds the weights to 11 bits and the result once, so it
      must lie within 0.5 (the rounding) + 0.25 (the weights) of the double value everywhere."""
    import scipy.ndimage as ndi

    rng = np.random.default_rng(77)
    w = np.array([18, 33, 49, 56, 49, 33, 18], np.int64)
    for img in (PIN["img"], rng.integers(0, 256, (61, 83), dtype=np.uint8), rng.integers(0, 256, (9, 12), dtype=np.uint8)):
        img = np.ascontiguousarray(img)
        h1 = ndi.correlate1d(img.astype(np.int64), w, axis=1, mode="mirror")
        v1 = ndi.correlate1d(h1, w, axis=0, mode="mirror")
        want = ((v1 + (1 << 15)) >> 16).astype(np.uint8)
        assert np.array_equal(orc.blur_image(img), want)
    src = np.ascontiguousarray(PIN["img"])
    sh, sw = src.shape
    for dw, dh in ((int(round(sw / 1.2)), int(round(sh / 1.2))), (sw // 2, sh // 2), (sw - 1, sh - 3)):
        got = orc.resize(src, dw, dh).astype(np.int32)
        ys = np.clip((np.arange(dh) + 0.5) * (sh / dh) - 0.5, 0, sh - 1)
        xs = np.clip((np.arange(dw) + 0.5) * (sw / dw) - 0.5, 0, sw - 1)
        yy, xx = np.meshgrid(ys, xs, indexing="ij")
        ref = ndi.map_coordinates(src.astype(np.float64), [yy, xx], order=1, mode="nearest")
        # one rounding of the result (<= 0.5) + the 11-bit weights (two taps x two axes: <= 255 * 4 * 2^-12 = 0.25)
        err = np.abs(got - ref)
        assert err.max() <= 0.5 + 0.25 and (err <= 0.5 + 1e-9).mean() > 0.9, (dw, dh, float(err.max()))
