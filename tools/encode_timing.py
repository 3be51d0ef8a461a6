"""Wall time of dsa_encode_batch against the CPU coder on 64k-triangle meshes (BASELINE.json configs[4])."""
import sys, time
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
meshes = []
for i in range(n):
    pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 1000 + i)
    meshes.append(dsa.MeshData(pos, faces, nrm, uv))
ctx = dsa.Context(0)
enc = dsa.DracoEncoder(ctx)
enc.EncodeBatch(meshes[:4])
t0 = time.perf_counter(); out = enc.EncodeBatch(meshes); t1 = time.perf_counter()
t2 = time.perf_counter(); ref = [synth.encode_mesh(m.positions, m.faces, m.normals, m.texcoords) for m in meshes[:16]]; t3 = time.perf_counter()
assert out[:16] == ref
print({"meshes": n, "gpu_path_s": round(t1 - t0, 3), "gpu_path_meshes_per_s": round(n / (t1 - t0), 1),
       "cpu_coder_1thread_meshes_per_s": round(16 / (t3 - t2), 1), "bytes_per_mesh": sum(map(len, out)) // n})
