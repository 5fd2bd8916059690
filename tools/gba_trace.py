"""The global-BA leg alone (what tools/profile_gba.sh traces): FullBA(4) -- 4 LM iterations, PCG limit 40 (reference
Snake/Optimizer/GlobalBundleAdjustment.cpp:32-43) -- on 300 keyframes x 15 000 points x 10 observations, three timed solves."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from snake_slam_amd import synth  # noqa: E402
from snake_slam_amd.ba import BARec, lba_options  # noqa: E402

sc, gt = synth.ba_scene(n_kf=300, n_pt=15000, obs_per_pt=10, seed=31, n_fixed=1)
ba = BARec(lba_options(max_iterations=4, max_pcg_iterations=40))
ba.create(sc)
ba.initAndSolve()
for _ in range(3):
    ba.reset()
    t0 = time.perf_counter()
    ci, cf = ba.initAndSolve()
    print("FullBA(4) 300 KF x 150 k obs: %.2f ms  cost %.6g -> %.6g" % ((time.perf_counter() - t0) * 1e3, ci[0], cf[0]), flush=True)
ba.close()
