"""Derives tests/golden/house_04_expected.npz from the reference's sample pair.

Inputs (read-only, only available in the build container):
  /root/reference/src/Draco.Examples/Samples/house_04.obj      (ground-truth geometry)
  /root/reference/src/Draco.Examples/Samples/house_04.obj.drc  (copied verbatim next to this file)
Output: v (float32 Nx3), vt (float32 Mx2), fv / fvt (int32 Fx3, zero-based) -- data only.
"""
import shutil
import sys

import numpy as np

SRC = "/root/reference/src/Draco.Examples/Samples/"
v, vt, fv, fvt = [], [], [], []
for line in open(SRC + "house_04.obj"):
    p = line.split()
    if not p:
        continue
    if p[0] == "v":
        v.append([float(x) for x in p[1:4]])
    elif p[0] == "vt":
        vt.append([float(x) for x in p[1:3]])
    elif p[0] == "f":
        assert len(p) == 4, "triangles only"
        a = [q.split("/") for q in p[1:4]]
        fv.append([int(q[0]) - 1 for q in a])
        fvt.append([int(q[1]) - 1 for q in a])
out = sys.argv[1] if len(sys.argv) > 1 else "tests/golden/"
np.savez_compressed(out + "house_04_expected.npz", v=np.array(v, np.float32), vt=np.array(vt, np.float32),
                    fv=np.array(fv, np.int32), fvt=np.array(fvt, np.int32))
shutil.copyfile(SRC + "house_04.obj.drc", out + "house_04.obj.drc")
print(len(v), len(vt), len(fv))
