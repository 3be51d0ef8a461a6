"""Step time of the bench batch under other encoder settings (diagnostics): python tools/variant_timing.py [meshes]
Rows of profiles/README.md "Other precisions and symbol schemes"."""
import sys; import os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
only = sys.argv[2] if len(sys.argv) > 2 else ""
ctx = dsa.Context(0); ctx.set_profiling(True)
for name, opt in (("bench (11/8/10 bit, auto scheme)", {}), ("14-bit positions", {"pos_bits": 14}), ("tagged symbols (force_scheme=0)", {"force_scheme": 0}),
                  ("14-bit positions + 12-bit UVs + 10-bit normals", {"pos_bits": 14, "uv_bits": 12, "normal_bits": 10})):
    if only and only not in name:
        continue
    blob, offs = synth.make_batch(synth.GRID, 128, 256, 1000, n, opt=synth.options(**opt))
    b = dsa.Batch(ctx, blob=blob, offsets=offs)
    for _ in range(3): b.decode()
    bad = sum(1 for i in range(n) if b.status(i) != 0)
    info = b.debug_array(0, 5, np.uint32, 64).reshape(16, 4)[:3]
    print(name, "| ms", {k: round(v, 2) for k, v in b.stage_times().items()}, "| failed", bad, "| (source, alphabet, precision, rans bytes):", info.tolist(), flush=True)
    b.close()
