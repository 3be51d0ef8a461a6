"""Damaged meshes (flipped, rewired, duplicated, missing faces; random face soups) through dsa_encode_batch with the connectivity on the
device and on the host: the same meshes must be coded (to the same bytes) and the same refused.  usage: python tools/fuzz_encode.py [count]"""
import os, sys
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
count = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
rng = np.random.default_rng(11)
ctx = dsa.Context(0); enc = dsa.DracoEncoder(ctx)
meshes = []
for it in range(count):
    mode = it % 6
    if mode >= 4:                 # faces taken away: new holes, new components, now and then a vertex that is no longer manifold
        p, n, u, f = synth.make_mesh(int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS])), int(rng.integers(4, 24)), int(rng.integers(4, 24)), int(rng.integers(0, 1 << 30)))
        keep = np.ones(len(f), bool); keep[rng.integers(0, len(f), int(rng.integers(1, 12)))] = False
        faces = f[keep]; used = np.unique(faces)
        remap = np.zeros(len(p), np.int64); remap[used] = np.arange(len(used))
        meshes.append(dsa.MeshData(p[used], np.ascontiguousarray(remap[faces], dtype=np.uint32)))
        continue
    if mode == 0:
        nv = int(rng.integers(4, 40)); faces = rng.integers(0, nv, (int(rng.integers(1, 80)), 3)).astype(np.uint32)
    else:
        p, n, u, f = synth.make_mesh(int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES])), int(rng.integers(4, 14)), int(rng.integers(4, 14)), int(rng.integers(0, 1 << 30)))
        faces = f.copy(); nv = len(p)
        for _ in range(int(rng.integers(1, 4))):
            k = int(rng.integers(0, len(faces)))
            if mode == 1: faces[k] = faces[k][::-1]
            elif mode == 2: faces[k, int(rng.integers(0, 3))] = int(rng.integers(0, nv))
            else: faces = np.concatenate([faces, faces[k:k + 1][:, [1, 2, 0]] if rng.integers(0, 2) else faces[int(rng.integers(0, len(faces)))][None, ::-1]])
    meshes.append(dsa.MeshData(rng.normal(size=(nv, 3)).astype(np.float32), np.ascontiguousarray(faces, dtype=np.uint32)))
def run(conn):
    os.environ["DSA_ENC_HOST_CONN"] = conn; os.environ["DSA_ENC_HOST_PLAN"] = conn
    out = []
    for k in range(0, count, 500):                      # a batch refuses its bad meshes one by one: results per mesh
        res = enc.EncodeBatch(meshes[k:k + 500], return_errors=True) if "return_errors" in enc.EncodeBatch.__code__.co_varnames else None
        if res is None:
            for m in meshes[k:k + 500]:
                try: out.append(enc.EncodeBatch([m])[0])
                except Exception as e: out.append(type(e).__name__)
        else: out.extend(res)
    return out
dev, host = run("0"), run("1")
coded = sum(1 for d in dev if isinstance(d, (bytes, bytearray)))
diff = sum(1 for d, h in zip(dev, host) if d != h)
print("%d meshes: %d coded, %d refused, %d verdicts / streams differ between device and host connectivity" % (count, coded, count - coded, diff))
sys.exit(1 if diff else 0)
