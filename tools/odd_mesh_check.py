"""One-off: parity + time on awkward topologies (thousands of components, one huge vertex fan, long thin strip)."""
import sys; sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import numpy as np, time, oracle, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
import test_gpu_parity as T
rng = np.random.default_rng(3)
meshes = {}
# 20000 disjoint quads
n = 20000
base = rng.uniform(-1, 1, (n, 1, 3)).astype(np.float32)
quad = np.array([[0, 0, 0], [1, 0, 0], [1, 1, 0], [0, 1, 0]], np.float32) * 0.01
pos = (base + quad[None]).reshape(-1, 3)
faces = (np.arange(n)[:, None, None] * 4 + np.array([[0, 1, 2], [0, 2, 3]])[None]).reshape(-1, 3).astype(np.uint32)
meshes["components"] = (pos, faces)
# fan: centre vertex with 60000 triangles around it (closed disc)
k = 60000
ang = np.linspace(0, 2 * np.pi, k, endpoint=False)
pos = np.concatenate([[[0, 0, 0]], np.stack([np.cos(ang), np.sin(ang), 0.1 * np.sin(7 * ang)], 1)]).astype(np.float32)
faces = np.stack([np.zeros(k, np.int64), 1 + np.arange(k), 1 + (np.arange(k) + 1) % k], 1).astype(np.uint32)
meshes["fan"] = (pos, faces)
# strip: 2 x 100000 vertices
m = 100000
x = np.arange(m, dtype=np.float32) / m
pos = np.concatenate([np.stack([x, np.zeros(m, np.float32), np.sin(40 * x)], 1), np.stack([x, np.full(m, 0.01, np.float32), np.sin(40 * x)], 1)]).astype(np.float32)
i = np.arange(m - 1)
faces = np.concatenate([np.stack([i, i + 1, m + i], 1), np.stack([i + 1, m + i + 1, m + i], 1)]).astype(np.uint32)
meshes["strip"] = (pos, faces)
ctx = dsa.Context(0); ctx.set_profiling(True)
for name, (pos, faces) in meshes.items():
    s = synth.encode_mesh(pos, faces)
    b = dsa.Batch(ctx, [s]); b.decode(); t0 = time.time(); b.decode(); dt = time.time() - t0
    assert b.status(0) == 0, (name, b.status(0), b.mesh_info(0).detail)
    T.assert_same(b.result(0), oracle.decode(s), b, 0)
    print(name, len(faces), "faces,", len(s), "bytes: decode %.3f s" % dt, {k: round(v, 1) for k, v in b.stage_times().items()})
    b.close()
