"""An encode batch, twice (a subject for rocprofv3; the second call is the warm one).  usage: python tools/enc_once.py [meshes]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import time, numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 1000)
meshes = [dsa.MeshData(pos, faces, nrm, uv) for _ in range(n)]
ctx = dsa.Context(0); enc = dsa.DracoEncoder(ctx)
enc.EncodeBatch(meshes)
t0 = time.perf_counter(); out = enc.EncodeBatch(meshes); dt = time.perf_counter() - t0
print("%d meshes: %.1f ms, %.0f meshes/s" % (n, dt * 1e3, n / dt))
