"""Seeded fuzz of the BATCHED, device-resident stereo and brute-force matchers and of the rectification kernel against the oracle:
random batch sizes (both stereo kernels: B < 8 row-index sort, B >= 8 one workgroup per frame; both kNN-2 kernels), capacities, counts
per frame including 0 and 1, level counts, relaxed / strict gates, random distortion / rotation / new-K rectifications.

    python tools/fuzz_match_batch.py [--seconds 120] [--seed 1]
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from helpers import knn_to_array, make_stereo_case, rand_desc  # noqa: E402
from oracle import oracle as orc  # noqa: E402
from oracle.oracle import KP64  # noqa: E402
from snake_slam_amd.matcher import BruteForceMatcher, Preprocess, Rectification, StereoMatcher  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--seconds", type=float, default=120.0)
    ap.add_argument("--seed", type=int, default=1)
    a = ap.parse_args()
    orc.build()
    rng = np.random.default_rng(a.seed)
    dev = torch.device("cuda:0")
    bf, st, pp = BruteForceMatcher(0), StereoMatcher(0), Preprocess(0)
    t = lambda x: torch.from_numpy(x).to(dev)  # noqa: E731
    t0, n = time.time(), {"bf": 0, "stereo": 0, "rectify": 0}
    while time.time() - t0 < a.seconds:
        kind = int(rng.integers(0, 3))
        B = int(rng.choice([1, 2, 7, 8, 9, 24, 40]))
        if kind == 0:
            capq, capt = int(rng.choice([1, 64, 300, 1000])), int(rng.choice([1, 65, 400, 1000]))
            nq, nt = rng.integers(0, capq + 1, B).astype(np.int32), rng.integers(0, capt + 1, B).astype(np.int32)
            q, tr = rand_desc(rng, B * capq).reshape(B, capq, 4), rand_desc(rng, B * capt).reshape(B, capt, 4)
            out = torch.full((B, capq, 4), -7, dtype=torch.int32, device=dev)
            pairs = torch.full((B, capq, 2), -7, dtype=torch.int32, device=dev)
            npairs = torch.zeros(B, dtype=torch.int32, device=dev)
            th, ratio = int(rng.integers(1, 257)), float(rng.choice([0.6, 0.8, 0.9, 1.0]))
            # the inputs stay referenced until the handle's stream has been synchronised: the matcher runs on its own stream, which
            # torch's allocator knows nothing about -- a temporary freed after the (asynchronous) call could be handed out again
            d_q, d_nq, d_t, d_nt = t(q.view(np.int64)), t(nq), t(tr.view(np.int64)), t(nt)
            torch.cuda.synchronize()
            bf.knn2_batch_dev(d_q, d_nq, d_t, d_nt, out)
            bf.filter_batch_dev(out, d_nq, th, ratio, pairs, npairs)
            bf.sync()
            out_h, pairs_h, np_h = out.cpu().numpy(), pairs.cpu().numpy(), npairs.cpu().numpy()
            ok, detail = True, ""
            for b in range(B):
                want = orc.bf_knn2(q[b, : nq[b]], tr[b, : nt[b]])
                wp = orc.bf_filter(want, th, ratio)
                parts = (np.array_equal(out_h[b, : nq[b]], knn_to_array(want)), bool((out_h[b, nq[b]:] == -7).all()), np_h[b] == wp.shape[0],
                         np.array_equal(pairs_h[b, : min(np_h[b], wp.shape[0])], wp[: min(np_h[b], wp.shape[0])]))
                if not all(parts) and ok:
                    detail = f" frame {b} nq {nq[b]} nt {nt[b]} (knn, tail, count, pairs) ok = {parts}; knn {out_h[b, : nq[b]].tolist()[:3]} " \
                             f"want {knn_to_array(want).tolist()[:3]}; pairs {np_h[b]} {pairs_h[b, :3].tolist()} want {wp.tolist()[:3]}"
                ok = ok and all(parts)
            what = f"bf B {B} cap {capq}x{capt} th {th} ratio {ratio}" + (detail if not ok else "")
            n["bf"] += 1
        elif kind == 1:
            capl, capr = int(rng.choice([70, 700, 1500])), int(rng.choice([65, 650, 1400]))
            levels, relaxed = int(rng.integers(1, 8)), bool(rng.integers(0, 2))
            L, R = np.zeros((B, capl), KP64), np.zeros((B, capr), KP64)
            DL, DR = np.zeros((B, capl, 4), np.uint64), np.zeros((B, capr, 4), np.uint64)
            nl = np.array([int(rng.choice([0, 1, capl, int(rng.integers(0, capl + 1))])) for _ in range(B)], np.int32)
            nr = np.array([int(rng.choice([0, 1, capr, int(rng.integers(0, capr + 1))])) for _ in range(B)], np.int32)
            bfv = 47.9
            for b in range(B):
                if nl[b] and nr[b]:
                    l, dl, r, dr, bfv, _ = make_stereo_case(rng, int(nl[b]), int(nr[b]), n_levels=levels)
                    L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]] = l, dl, r, dr
            ls = (np.float32(1.2) ** np.arange(levels)).astype(np.float32)
            rp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
            dp = torch.full((B, capl), -1000.0, dtype=torch.float32, device=dev)
            nm = torch.full((B,), -1, dtype=torch.int32, device=dev)
            d_in = (t(L.view(np.uint8).reshape(B, capl, 24)), t(DL.view(np.int64)), t(nl), t(R.view(np.uint8).reshape(B, capr, 24)),
                    t(DR.view(np.int64)), t(nr))
            torch.cuda.synchronize()
            st.match_batch_dev(*d_in, bfv, ls, relaxed, rp, dp, nm)
            st.sync()
            rp_h, dp_h, nm_h = rp.cpu().numpy(), dp.cpu().numpy(), nm.cpu().numpy()
            ok = True
            for b in range(B):
                n2, rp2, dp2 = orc.stereo_match(L[b, : nl[b]], DL[b, : nl[b]], R[b, : nr[b]], DR[b, : nr[b]], bfv, ls, relaxed)
                ok = ok and nm_h[b] == n2 and np.array_equal(rp_h[b, : nl[b]], rp2) and np.array_equal(dp_h[b, : nl[b]], dp2) and \
                    (rp_h[b, nl[b]:] == -1000).all()
            what = f"stereo B {B} cap {capl}x{capr} levels {levels} relaxed {relaxed}"
            n["stereo"] += 1
        else:
            cnt = int(rng.choice([0, 1, 63, 64, 65, int(rng.integers(2, 3000))]))
            k = np.zeros(cnt, orc.KEYPOINT)
            k["x"], k["y"] = rng.uniform(-20, 780, cnt).astype(np.float32), rng.uniform(-20, 500, cnt).astype(np.float32)
            k["angle"], k["octave"] = rng.uniform(0, 360, cnt).astype(np.float32), rng.integers(0, 8, cnt)
            K = (float(rng.uniform(300, 700)), float(rng.uniform(300, 700)), float(rng.uniform(300, 420)), float(rng.uniform(200, 280)))
            nd = int(rng.choice([4, 5, 8]))  # 4 / 5 / 8 distortion coefficients in use, the rest zero
            D = tuple(np.concatenate([rng.normal(0, [0.2, 0.1, 0.01, 0.01, 0.01, 0.001, 0.001, 0.001][:nd]), np.zeros(8 - nd)])) if rng.random() < 0.8 else None
            ang = float(rng.normal(0, 0.02))
            Rm = np.array([[np.cos(ang), 0, np.sin(ang)], [0, 1, 0], [-np.sin(ang), 0, np.cos(ang)]]) if rng.random() < 0.5 else None
            Kd = tuple(float(v) * float(rng.uniform(0.9, 1.1)) for v in K) if rng.random() < 0.5 else None
            ro, rg = orc.rectification(K, D, Rm, Kd), Rectification.make(K, D, Rm, Kd)
            want, wn = orc.rectify(ro, k)
            got, gn = pp.rectify(rg, k)
            ok = all(np.array_equal(got[f], want[f]) for f in ("x", "y", "angle", "octave")) and np.array_equal(gn, wn)
            what = f"rectify n {cnt} D {None if D is None else nd} R {Rm is not None} Kd {Kd is not None}"
            n["rectify"] += 1
        if not ok:
            print(f"MISMATCH: {what} (seed {a.seed}, case {sum(n.values())})")
            return 1
    for h in (bf, st, pp):
        h.close()
    print(f"fuzz_match_batch: {n}, all bit-exact (seed {a.seed}, {time.time() - t0:.0f} s)")
    return 0


if __name__ == "__main__":
    sys.exit(main())
