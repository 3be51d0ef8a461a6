"""Attempt-by-attempt trace of the traversal of one bench mesh from a -DDSA_TRAV_TRACE build (DSA_LIB=build_abl/lib_trace.so)."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, 4)
ctx = dsa.Context(0)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
b.decode()
t = b.debug_array(0, 6, np.uint32, 33000)
n = int(t[0]); print("attempts", n)
w = t[1:1 + min(n, 8000)]
kind, K, ln, win, lin = w & 15, (w >> 4) & 255, (w >> 12) & 255, (w >> 20) & 255, (w >> 28) & 1
print("first 120 (kind K len window lin):")
print(" ".join("%d:%d/%d/%d%s" % (kind[i], K[i], ln[i], win[i], "L" if lin[i] else "") for i in range(min(120, len(w)))))
print("middle:")
m = len(w) // 2
print(" ".join("%d:%d/%d/%d%s" % (kind[i], K[i], ln[i], win[i], "L" if lin[i] else "") for i in range(m, min(m + 80, len(w)))))
d = t[8192:8192 + 8 * 64 * 16].reshape(8, 64, 16)
for at in range(8):
    i = 1000 + at
    print("attempt", i, "kind/K/len", kind[i], K[i], ln[i])
    for l in range(4):
        r = d[at, l].astype(np.int64)
        print("  lane %d  a %d b %d tipA %d tipB %d lcA %d rcB %d lcB %d marks %08x" % (l, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7]))
        print("     pred a %d b %d tipA %d tipB %d lcA %d rcB %d a_next %d" % (r[8], r[9], r[10], r[11], r[12], r[13], r[14]))
