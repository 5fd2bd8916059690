"""Diagnosis of tests/test_ba_gpu.py::test_big_batch_of_unequal_scenes under a forced BA path (round-3 red gate): every scene of
the 300-scene batch against the oracle, per scene the sizes, the PCG iteration counts of both sides, cost and RMSE deltas; the
scenes that miss the tolerance are solved again (a) alone on the same path, (b) with a PCG that may converge, on both sides.

    SNK_BA_NO_SCHUR_SET=1 python tools/probes/diag_big_batch.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as orc  # noqa: E402
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.ba import BARec, lba_options  # noqa: E402


def rmse(a, b):
    return float(np.sqrt(((np.asarray(a) - np.asarray(b)) ** 2).sum(axis=-1).mean())) if len(a) else 0.0


def make_scenes():
    rng = np.random.default_rng(17)
    scenes = []
    for k in range(300):
        n_kf = int(rng.integers(3, 9))
        opp = int(rng.integers(2, n_kf + 1))
        n_min = max(24, -(-8 * n_kf // opp))
        n_pt = int(rng.choice([n_min, max(n_min, 65 * n_kf // 2), max(n_min, 129 * n_kf // 3), int(rng.integers(n_min, n_min + 400))]))
        sc, _ = synth.ba_scene(n_kf=n_kf, n_pt=n_pt, obs_per_pt=opp, seed=1000 + k, n_fixed=int(rng.integers(1, 3)), outlier_frac=0.02)
        if k % 7 == 0:
            sc["pt_const"] = (rng.random(n_pt) < 0.15).astype(np.uint8)
        scenes.append(sc)
    return scenes


def main():
    orc.build()
    env = {k: v for k, v in os.environ.items() if k.startswith("SNK_")}
    print("env", env)
    scenes = make_scenes()
    ba = BARec(lba_options())
    ba.create(scenes)
    ci, cf = ba.initAndSolve()
    bad = []
    for k, s in enumerate(scenes):
        wpose, wpt, wci, wcf, wpcg = orc.ba_solve(s, orc.ba_options())
        pose, pt, pcg = ba.state(k)
        dci = abs(ci[k] - wci) / max(1.0, wci)
        dcf = abs(cf[k] - wcf) / max(1.0, wcf)
        r = max(rmse(pose, wpose), rmse(pt, wpt))
        ok = dci <= 1e-9 and dcf <= 1e-7 and r <= 1e-5
        if not ok:
            bad.append(k)
            print(f"scene {k}: {len(s['pose'])} kf ({int(s['img_const'].sum())} const) x {len(s['pt'])} pts ({int(s['pt_const'].sum())} const) x "
                  f"{len(s['obs_img'])} obs; ci rel {dci:.3g} cf rel {dcf:.3g} (cf {cf[k]:.9g} vs {wcf:.9g}) rmse {r:.3g} pcg {pcg} vs {wpcg}")
    ba.close()
    print(f"{len(bad)} of {len(scenes)} scenes outside tolerance in the batch: {bad}")
    for k in bad:
        s = scenes[k]
        one = BARec(lba_options())
        one.create(s)
        ci1, cf1 = one.initAndSolve()
        pose1, pt1, pcg1 = one.state(0)
        one.close()
        wpose, wpt, wci, wcf, wpcg = orc.ba_solve(s, orc.ba_options())
        conv = BARec(lba_options(max_pcg_iterations=2000))
        conv.create(s)
        ci2, cf2 = conv.initAndSolve()
        pose2, pt2, pcg2 = conv.state(0)
        conv.close()
        wpose2, wpt2, _, wcf2, wpcg2 = orc.ba_solve(s, orc.ba_options(max_pcg_iterations=2000))
        print(f"scene {k} alone: cf rel {abs(cf1[0] - wcf) / max(1.0, wcf):.3g} rmse {max(rmse(pose1, wpose), rmse(pt1, wpt)):.3g} pcg {pcg1} vs {wpcg}; "
              f"converged PCG: cf rel {abs(cf2[0] - wcf2) / max(1.0, wcf2):.3g} rmse {max(rmse(pose2, wpose2), rmse(pt2, wpt2)):.3g} pcg {pcg2} vs {wpcg2}")
    return 0


if __name__ == "__main__":
    sys.exit(main())
