"""Seeded fuzz of the matchers against the oracle (bit-exact): brute-force kNN-2 + filter, the stereo row-band matcher, the
projection matchers (coarse / fine / keyframe) and the local-mapping matchers (fuse, triangulation by projection / BoW / brute force,
relink) through the host API, with random sizes (including 0, 1, one more / less than a
wavefront), thresholds, radii and descriptor entropies.  Not part of the test suite; run after changes to matcher.hip / track.hip:

    python tools/fuzz_match.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import track_helpers as T  # noqa: E402
from helpers import knn_to_array, make_stereo_case, rand_desc  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd.matcher import BruteForceMatcher, StereoMatcher  # noqa: E402
from snake_slam_amd.tracking import DeferredMapper, MappingORBMatcher, SnakeORBMatcher  # noqa: E402


def sizes(rng, hi):
    return int(rng.choice([0, 1, 2, 63, 64, 65, int(rng.integers(3, hi)), int(rng.integers(3, hi))]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    bf, st, pm, mm, dm = BruteForceMatcher(0), StereoMatcher(0), SnakeORBMatcher(0), MappingORBMatcher(), DeferredMapper()
    t0, n = time.time(), {"bf": 0, "stereo": 0, "coarse": 0, "fine": 0, "keyframe": 0, "fuse": 0, "tri_project": 0, "tri_bow": 0,
                          "tri_bf": 0, "relink": 0, "grid": 0, "rgbd": 0, "frontend": 0}
    from snake_slam_amd.matcher import Preprocess, Rectification, RgbdModel  # noqa: E402
    from snake_slam_amd.tracking import FeatureGrid  # noqa: E402
    from snake_slam_amd.frontend import Frontend  # noqa: E402
    from snake_slam_amd import synth  # noqa: E402
    from oracle.oracle import KP64  # noqa: E402

    grid, pre = FeatureGrid(0), Preprocess(0)
    fe_cache = {}
    from snake_slam_amd import _lib  # noqa: E402

    while time.time() - t0 < a.seconds:
        kind = int(rng.integers(0, 13))
        if kind <= 1:  # the [DEFINED] switches (snk_set_definition), the same random setting in the library and in the oracle
            for key, (lo, hi) in _lib.DEFINITIONS.items():
                v = int(rng.integers(lo, hi + 1)) if rng.random() < 0.5 else 0
                _lib.set_definition(key, v)
                orc.set_definition(key, v)
        if kind == 0:
            nq, nt = sizes(rng, 2500), sizes(rng, 2500)
            q, t = rand_desc(rng, nq), rand_desc(rng, nt)
            if rng.random() < 0.3 and nq and nt:  # low entropy: ties everywhere
                base = rand_desc(rng, 4)
                q, t = base[rng.integers(0, 4, nq)], base[rng.integers(0, 4, nt)]
            bf.matchKnn2(q, t)
            want = orc.bf_knn2(q, t)
            ok = np.array_equal(knn_to_array(bf.knn), knn_to_array(want))
            th, ratio = int(rng.integers(1, 257)), float(rng.choice([0.6, 0.75, 0.8, 0.9, 1.0]))
            cnt = bf.filterMatches(th, ratio)
            wp = orc.bf_filter(want, th, ratio)
            ok = ok and cnt == wp.shape[0] and np.array_equal(bf.matches, wp)
            what = f"bf {nq}x{nt} th {th} ratio {ratio}"
            n["bf"] += 1
        elif kind == 1:
            nl, nr = sizes(rng, 3000), sizes(rng, 3000)
            relaxed = bool(rng.integers(0, 2))
            if nl == 0 or nr == 0:
                continue  # make_stereo_case needs one keypoint on each side; the empty cases are in the test suite
            kl, dl, kr, dr, bfv, ls = make_stereo_case(rng, nl, nr, n_levels=int(rng.integers(1, 8)))
            if rng.random() < 0.5:  # rows exactly on .5 (where the iRound definitions differ), some of them negative
                kl["y"], kr["y"] = np.floor(kl["y"]) + 0.5, np.floor(kr["y"]) + 0.5
                kl["y"][: nl // 8] -= 70.0
                kr["y"][: nr // 8] -= 70.0
            got = st.StereoMatching(kl, dl, kr, dr, bfv, ls, relaxed)
            want = orc.stereo_match(kl, dl, kr, dr, bfv, ls, relaxed)
            ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            what = f"stereo {nl}x{nr} relaxed {relaxed}"
            n["stereo"] += 1
        elif kind == 10:  # feature grid (counting form, or the network where the grid is too large for it): points in and out of the bounds, clusters
            npt = sizes(rng, 3000)
            w, h = float(rng.integers(40, 2600)), float(rng.integers(40, 1600))
            bounds = (float(rng.integers(-200, 1)), float(rng.integers(-200, 1)), w, h)
            k = np.zeros(npt, KP64)
            k["x"], k["y"] = rng.uniform(bounds[0] - 50, w + 50, npt), rng.uniform(bounds[1] - 50, h + 50, npt)
            if npt > 10 and rng.random() < 0.4:  # a crowded cell
                k["x"][: npt // 2], k["y"][: npt // 2] = rng.uniform(100, 118, npt // 2), rng.uniform(60, 78, npt // 2)
            got = grid.create(bounds, k)
            want = orc.feature_grid(k, bounds)
            ok = np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1]) and got[2:] == tuple(want[2:])
            what = f"grid n {npt} bounds {bounds}"
            n["grid"] += 1
        elif kind == 11:  # Preprocess::ComputeStereoFromRGBD
            npt = sizes(rng, 3000)
            w, h = int(rng.integers(64, 1300)), int(rng.integers(64, 800))
            K = (float(rng.uniform(300, 800)), float(rng.uniform(300, 800)), w / 2.0, h / 2.0)
            Kd = (K[0] * float(rng.uniform(0.9, 1.1)), K[1] * float(rng.uniform(0.9, 1.1)), K[2] + float(rng.uniform(-3, 3)), K[3] + float(rng.uniform(-3, 3)))
            Dd = tuple(float(v) for v in rng.uniform(-1, 1, 8) * np.array([0.1, 0.05, 0.01, 0.01, 0.01, 0.01, 1e-3, 1e-3])) if rng.random() < 0.7 else (0.0,) * 8
            k = np.zeros(npt, KP64)
            k["x"], k["y"] = rng.uniform(0.2 * w, 0.8 * w, npt), rng.uniform(0.2 * h, 0.8 * h, npt)
            img = np.where(rng.random((h, w)) < 0.3, 0.0, rng.uniform(0.2, 19.9, (h, w))).astype(np.float32)
            want = orc.rgbd_stereo(k, K, Dd, Kd, 40.0, img)
            try:
                got = pre.ComputeStereoFromRGBD(RgbdModel.make(K, Dd, Kd, 40.0), k, img)
                ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
            except Exception:
                ok = want[0] < 0  # the reference's abort, reported as an error
            what = f"rgbd n {npt} image {w}x{h}"
            n["rgbd"] += 1
        elif kind == 12:  # snk_frontend_process against the oracle chain (a handle per image size, so the graph path is exercised too)
            w, h = int(rng.choice([96, 160, 320, 333])), int(rng.choice([96, 128, 240]))
            orb = (int(rng.choice([100, 300])), 1.2, int(rng.integers(1, 4)), 20, 7)
            key = (w, h, orb)
            if key not in fe_cache:
                if len(fe_cache) >= 6:
                    fe_cache.pop(next(iter(fe_cache))).close()
                fe_cache[key] = Frontend(orb, bounds=(0.0, 0.0, float(w), float(h)), bf=40.0)
            fe = fe_cache[key]
            l, r = synth.stereo_frame(int(rng.integers(0, 1 << 20)), w, h, n_rects=int(rng.integers(5, 90)))
            got = fe.Process(l, r)
            p_ = orc.orb_params(*orb)
            kl, dl = orc.orb_detect(p_, l)
            kr, dr = orc.orb_detect(p_, r)
            rect = orc.rectification((1.0, 1.0, 0.0, 0.0))
            ul, nl_ = orc.rectify(rect, kl)
            ur, _ = orc.rectify(rect, kr)
            perm = np.asarray(orc.feature_grid(ul, (0.0, 0.0, float(w), float(h)))[0])
            g, gd = np.zeros_like(ul), np.zeros_like(dl)
            g[perm], gd[perm] = ul, dl
            wn, wrp, wdp = orc.stereo_match(g, gd, ur, dr, 40.0, fe.level_scale, True) if len(kl) and len(kr) else (0, np.full(len(kl), -1000, np.float32), np.full(len(kl), -1000, np.float32))
            ok = got["N"] == len(kl) and got["n_right"] == len(kr) and got["n_stereo"] == wn and np.array_equal(got["descriptors"], gd) and \
                np.array_equal(got["right_points"], wrp) and np.array_equal(got["depth"], wdp) and np.array_equal(got["descriptors_right"], dr) and \
                np.array_equal(got["permutation"], perm)
            what = f"frontend {w}x{h} orb {orb}"
            n["frontend"] += 1
        elif kind >= 5:
            m_pts, clutter = max(30, sizes(rng, 2000)), sizes(rng, 1500)
            if kind == 5:
                frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=clutter, m_pts=m_pts, n_levels=int(rng.integers(2, 8)))
                pts = T.fusion_points(orc, rng, world, pose, ls)
                mask = (rng.random(len(pts)) > 0.2).astype(np.uint8) if rng.random() < 0.5 else None
                th, of, fth = float(rng.uniform(1, 8)), float(rng.uniform(0.5, 4)), int(rng.integers(20, 120))
                got = mm.Fuse(frame, cam, pose, pts, mask, th, of, fth, ls)
                want = orc.match_fuse(frame, cam, pose, pts, mask, th, of, fth, ls)
                ok = got[0] == want[0] and np.array_equal(got[2], want[1])
                what, key = f"fuse m {len(pts)} n {len(frame['kps'])} th {th} of {of} fth {fth}", "fuse"
            elif kind == 6:
                c = T.make_triangulation_case(orc, rng, m_pts=m_pts, n_clutter=clutter)
                epi, fd = float(rng.uniform(0.5, 10)), int(rng.integers(20, 100))
                args = (c["grid"], c["pose1"], c["pose2"], c["cam"], c["kps1"], c["np1"], c["desc1"], c["has1"], c["frame2"], c["np2"], c["E"], epi, fd)
                got = mm.SearchForTriangulationProject(*args)
                want = orc.match_triangulation_project(*args)
                ok = got[0] == want[0] and np.array_equal(got[2], want[1])
                what, key = f"tri_project m {m_pts} clutter {clutter} epi {epi} fd {fd}", "tri_project"
            elif kind == 7:
                c = T.make_bow_case(rng, m_pts=m_pts, n_clutter=clutter, n_nodes=int(rng.choice([1, 3, 30, 120, 400])))
                epi, fd = float(rng.uniform(0.5, 10)), int(rng.integers(20, 100))
                args = (c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["bow1"], c["np2"], c["desc2"], c["has2"], c["bow2"], epi, fd)
                got = mm.SearchForTriangulation2(*args)
                want = orc.match_triangulation_bow(*args)
                ok = got[0] == want[0] and [tuple(map(int, p)) for p in got[1]] == [tuple(p) for p in np.asarray(want[1]).tolist()]
                what, key = f"tri_bow m {m_pts} clutter {clutter} epi {epi} fd {fd}", "tri_bow"
            elif kind == 8:
                c = T.make_bow_case(rng, m_pts=m_pts, n_clutter=clutter)
                fd = int(rng.integers(20, 100))
                args = (c["cam"], c["E"], c["np1"], c["desc1"], c["has1"], c["np2"], c["desc2"], c["has2"], fd)
                got = mm.SearchForTriangulationBF(*args)
                want = orc.match_triangulation_bf(*args)
                ok = got[0] == want[0] and np.array_equal(got[2], want[1])
                what, key = f"tri_bf m {m_pts} clutter {clutter} fd {fd}", "tri_bf"
            else:
                frame, cam, pose, qs = T.make_relink_case(orc, rng, n_base=max(20, sizes(rng, 2500)))
                got = dm.RelinkSearch(frame, cam, pose, qs)
                want = orc.match_relink(frame, cam, pose, qs)
                ok = got[0] == want[0] and np.array_equal(got[1], want[1]) and np.array_equal(got[2], want[2])
                what, key = f"relink n {len(frame['kps'])} q {len(qs)}", "relink"
            n[key] += 1
        else:
            frame, cam, pose, ls, world, _ = T.make_tracking_case(orc, rng, n_clutter=sizes(rng, 2500), m_pts=max(20, sizes(rng, 3000)),
                                                                  n_levels=int(rng.integers(1, 8)), stereo_frac=float(rng.random()),
                                                                  taken_frac=float(rng.random()) * 0.3)
            if kind == 2:
                pts = T.lm_coarse(orc, world)
                th, fe, direction = float(rng.uniform(2, 40)), int(rng.integers(20, 150)), int(rng.integers(0, 3))
                got = pm.SearchByProjectionFrameFrame2(frame, cam, pose, pts, th, fe, direction, ls)
                want = orc.match_coarse(frame, cam, pose, pts, th, fe, direction, ls)
                ok = got[0] == want[0] and np.array_equal(got[1], want[1])
                what, key = f"coarse m {len(pts)} n {len(frame['kps'])} th {th} fe {fe} dir {direction}", "coarse"
            elif kind == 3:
                pts = T.lm_fine(orc, rng, world, pose, ls)
                th, ratio = float(rng.uniform(0.5, 8)), float(rng.choice([0.6, 0.8, 0.9]))
                got = pm.SearchByProjection2(frame, cam, pose, pts.copy(), th, ratio, ls)
                want = orc.match_fine(frame, cam, pose, pts.copy(), th, ratio, ls)
                ok = got[0] == want[0] and all(np.array_equal(g, w) for g, w in zip(got[1:], want[1:]))
                what, key = f"fine m {len(pts)} n {len(frame['kps'])} th {th} ratio {ratio}", "fine"
            else:
                skip = (rng.random(len(world["pos"])) < 0.1).astype(np.uint8)
                th, fe = float(rng.uniform(2, 40)), int(rng.integers(20, 150))
                got = pm.SearchByProjectionFrameToKeyframe(frame, cam, pose, world["pos"], world["desc"], skip, th, fe)
                want = orc.match_keyframe(frame, cam, pose, world["pos"], world["desc"], skip, th, fe)
                ok = got[0] == want[0] and np.array_equal(got[1], want[1])
                what, key = f"keyframe m {len(skip)} n {len(frame['kps'])} th {th} fe {fe}", "keyframe"
            n[key] += 1
        if not ok:
            print(f"MISMATCH: {what} (seed {a.seed}, case {sum(n.values())})")
            return 1
    print(f"fuzz_match: {n}, all bit-exact (seed {a.seed}, {time.time() - t0:.0f} s)")
    for h in (bf, st, pm, mm):
        h.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
