"""When the traversal / connectivity wave of every mesh of a batch started and how long it ran (s_memrealtime stamps, 100 MHz),
against the kernel's duration: shows dispatch rounds and stragglers.  usage: python tools/wave_times.py [meshes]"""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n)
ctx = dsa.Context(0); ctx.set_profiling(True)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(6):
    b.decode()
    print({k: round(v, 2) for k, v in b.stage_times().items()})
d = np.array([b.debug_array(i, 4, np.uint32, 20) for i in range(n)]).astype(np.int64)
for name, ticks, start, dur in (("connectivity", 13, 14, 15), ("traverse", 6, 16, 17)):
    s0 = (d[:, start] - d[:, start].min()) / 1e5          # ms
    du = d[:, dur] / 1e5
    end = s0 + du
    print("%s: clock %.3f GHz; start ms pct[0,10,50,90,100] %s; duration %s; end %s" % (
        name, np.median(d[:, ticks] / np.maximum(d[:, dur], 1)) * 0.1,
        np.round(np.percentile(s0, [0, 10, 50, 90, 100]), 2).tolist(), np.round(np.percentile(du, [0, 10, 50, 90, 100]), 2).tolist(),
        np.round(np.percentile(end, [0, 10, 50, 90, 100]), 2).tolist()))
    h, e = np.histogram(s0, bins=12); print("  start histogram:", h.tolist(), np.round(e, 1).tolist())
    h, e = np.histogram(du, bins=12); print("  duration histogram:", h.tolist(), np.round(e, 1).tolist())
    # by position in the launch: mean start and duration of 16 consecutive groups of meshes
    g = n // 16
    print("  by mesh index (16 groups): start", np.round([s0[i * g:(i + 1) * g].mean() for i in range(16)], 1).tolist())
    print("                          duration", np.round([du[i * g:(i + 1) * g].mean() for i in range(16)], 1).tolist())
