"""Encode direction: N of the bench meshes through dsa_encode_batch, with the connectivity on the device (default) and on the
host cores (DSA_ENC_HOST_CONN=1 in the environment).  usage: python tools/encode_timing.py [meshes ...]"""
import sys, time; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
counts = [int(x) for x in sys.argv[1:]] or [128, 512]
ctx = dsa.Context(0)
enc = dsa.DracoEncoder(ctx)
base = [synth.make_mesh(synth.GRID, 128, 256, 1000 + i) for i in range(16)]
for n in counts:
    meshes = [dsa.MeshData(*(lambda m: (m[0], m[3], m[1], m[2]))(base[i % 16])) for i in range(n)]
    enc.EncodeBatch(meshes[:2])
    t0 = time.perf_counter()
    out = enc.EncodeBatch(meshes)
    dt = time.perf_counter() - t0
    check = synth.encode_mesh(meshes[0].positions, meshes[0].faces, meshes[0].normals, meshes[0].texcoords)
    print("%d meshes: %.1f ms, %.0f meshes/s, first stream equals the CPU coder's: %s" % (n, dt * 1e3, n / dt, out[0] == check), flush=True)
