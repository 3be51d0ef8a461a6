"""Randomised differential run of the encode direction: random meshes and encoder settings through dsa_encode_batch with the
connectivity and the symbol plans forced onto the device, byte for byte against the CPU coder, then decoded again.
usage: python tools/soak_encode.py [count] [seed]"""
import os, sys
os.environ["DSA_ENC_HOST_CONN"] = "0"; os.environ["DSA_ENC_HOST_PLAN"] = "0"
import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
count = int(sys.argv[1]) if len(sys.argv) > 1 else 300
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
ctx = dsa.Context(0)
enc = dsa.DracoEncoder(ctx)
bad = done = 0
while done < count:
    cfg = dsa.Config(position_bits=int(rng.integers(2, 19)), texcoord_bits=int(rng.integers(2, 17)), normal_bits=int(rng.integers(2, 15)),
                     speed=int(rng.integers(0, 11)), single_connectivity=bool(rng.integers(0, 2)), symbol_scheme=int(rng.choice([-1, -1, 0, 1])),
                     position_prediction=int(rng.choice([0, 1])), texcoord_prediction=int(rng.choice([0, 1])))
    group = []
    for _ in range(int(rng.integers(1, 12))):
        kind = int(rng.choice([synth.GRID, synth.TORUS, synth.SPHERE, synth.HOLES, synth.TWO_PARTS]))
        nx, ny = int(rng.integers(4, 70)), int(rng.integers(4, 60))
        if kind == synth.HOLES: nx, ny = max(nx, 12), max(ny, 12)
        pos, nrm, uv, faces = synth.make_mesh(kind, nx, ny, int(rng.integers(0, 1 << 30)))
        gen = None
        if rng.integers(0, 3) == 0:                  # a generic uint8 attribute of 1 - 4 components (ABI 4)
            gen = rng.integers(0, 256, (len(pos), int(rng.integers(1, 5)))).astype(np.uint8)
        group.append((pos, faces, nrm if rng.integers(0, 4) else None, uv if rng.integers(0, 4) else None, gen))
    got = enc.EncodeBatch([dsa.MeshData(*m) for m in group], cfg)
    opt = synth.options(pos_bits=cfg.position_bits, uv_bits=cfg.texcoord_bits, normal_bits=cfg.normal_bits, single_connectivity=1 if cfg.single_connectivity else 0,
                        force_scheme=cfg.symbol_scheme, compression_level=10 - cfg.speed, pos_prediction=cfg.position_prediction, uv_prediction=cfg.texcoord_prediction)
    for (p, f, n, u, gen), g in zip(group, got):
        opt.generic_components = gen.shape[1] if gen is not None else 1
        if g != synth.encode_mesh(p, f, n, u, generic=gen, opt=opt):
            bad += 1; print("differs:", cfg.__dict__, len(f))
    b = dsa.Batch(ctx, got); b.decode()
    bad += sum(1 for i in range(len(got)) if b.status(i) != 0)
    b.close()
    done += len(group)
print("%d meshes, %d bad" % (done, bad))
sys.exit(1 if bad else 0)
