"""Throughput with several batches in flight (one context = one set of streams per batch): the head of one decode
(entropy decode, connectivity) overlaps the tail of the previous one (prediction, finalisation).
usage: python tools/pipeline_check.py [meshes] [in_flight ...]"""
import sys, time
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import torch
import draco_sharp_amd as dsa, draco_sharp_amd.synth as synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
depths = [int(x) for x in sys.argv[2:]] or [1, 2, 3]
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n)
steps = 12
for depth in depths:
    ctxs = [dsa.Context(0) for _ in range(depth)]
    batches = [dsa.Batch(c, blob=blob, offsets=offs) for c in ctxs]
    for b in batches:
        b.decode(wait=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for s in range(steps):
        b = batches[s % depth]
        if s >= depth:
            b.wait()
        b.decode(wait=False)
    for b in batches:
        b.wait()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    ok = all(b.status(i) == 0 for b in batches for i in (0, n // 2, n - 1))
    print("in flight %d: %.2f ms per step, %.0f meshes/s, ok=%s" % (depth, dt * 1e3, n / dt, ok), flush=True)
    for b in batches:
        b.close()
    for c in ctxs:
        c.close()
