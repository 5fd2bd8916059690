"""Experiment: 256 windows in one handle vs 2 x 128 on two streams."""
import sys, time
sys.path.insert(0, ".")
import torch
from snake_slam_amd import synth
from snake_slam_amd.ba import BARec, lba_options

NW, IT, STEPS = 256, 3, 10
distinct = [synth.ba_scene(seed=synth.SEED + k)[0] for k in range(4)]
def run(handles):
    for h in handles:
        h.reset(); h.solve_async(IT)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(STEPS):
        for h in handles:
            h.reset(); h.solve_async(IT)
    torch.cuda.synchronize()
    return NW * IT * STEPS / (time.perf_counter() - t0)
one = BARec(lba_options()); one.create([distinct[k % 4] for k in range(NW)])
print("1 handle  : %.0f LM it/s" % run([one]))
for parts in (2, 4):
    streams = [torch.cuda.Stream() for _ in range(parts)]
    hs = []
    for s in streams:
        h = BARec(lba_options(), stream=s.cuda_stream); h.create([distinct[k % 4] for k in range(NW // parts)]); hs.append(h)
    print("%d handles : %.0f LM it/s" % (parts, run(hs)))
