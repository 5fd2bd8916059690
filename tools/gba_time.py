import sys, time
sys.path.insert(0, __import__('os').path.dirname(__import__('os').path.dirname(__import__('os').path.abspath(__file__))))
import numpy as np
from snake_slam_amd import synth
from snake_slam_amd.ba import BARec, lba_options
from oracle import oracle as orc
for n_kf, n_pt in [(120, 6000), (300, 15000), (600, 30000)]:
    sc, gt = synth.ba_scene(n_kf=n_kf, n_pt=n_pt, obs_per_pt=10, seed=31, n_fixed=1)
    ba = BARec(lba_options(max_iterations=4, max_pcg_iterations=40))
    t0 = time.perf_counter(); ba.create(sc); t1 = time.perf_counter()
    ci, cf = ba.initAndSolve(); t2 = time.perf_counter()
    ta = time.perf_counter(); ba.create(sc); tb = time.perf_counter()  # the same handle again: buffers and pinned lists keep their capacity
    print("   second create on the handle %.1f ms" % ((tb - ta) * 1e3), flush=True)
    ba.initAndSolve()
    ba.reset(); t3 = time.perf_counter(); ci, cf = ba.initAndSolve(); t4 = time.perf_counter()
    print(n_kf, n_pt, "create %.1f ms solve(first) %.1f ms solve %.1f ms" % ((t1-t0)*1e3, (t2-t1)*1e3, (t4-t3)*1e3), ci[0], cf[0], flush=True)
    pose, pt, _ = ba.state(0)
    ba.close()
    if n_kf <= 300:
        t0 = time.perf_counter()
        wpose, wpt, wci, wcf, it = orc.ba_solve(sc, orc.ba_options(4, 40))
        t1 = time.perf_counter()
        print("   oracle %.1f ms" % ((t1-t0)*1e3), wci, wcf, "rmse pose %.2e pt %.2e" % (np.sqrt(((pose-wpose)**2).sum(-1).mean()), np.sqrt(((pt-wpt)**2).sum(-1).mean())), flush=True)
