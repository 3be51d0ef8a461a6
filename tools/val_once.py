"""One valence-coded batch decoded a few times (a subject for rocprofv3).  usage: python tools/val_once.py [meshes]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n, opt=synth.options(predictive_connectivity=2))
ctx = dsa.Context(0)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(3): b.decode()
print("failed", sum(1 for i in range(n) if b.status(i) != 0))
