"""Phase times of one encode chunk (DSA_ENC_TIMING=1 on stderr), a warm second call.  usage: python tools/enc_phases.py [meshes]"""
import os, sys
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), '..')
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pos, nrm, uv, faces = synth.make_mesh(synth.GRID, 128, 256, 1000)
meshes = [dsa.MeshData(pos, faces, nrm, uv) for _ in range(n)]
os.environ["DSA_ENC_TIMING"] = "1"
ctx = dsa.Context(0); enc = dsa.DracoEncoder(ctx)
os.environ["DSA_ENC_TIMING"] = "1"
enc.EncodeBatch(meshes)
sys.stderr.write("---- warm call\n"); sys.stderr.flush()
enc.EncodeBatch(meshes)
