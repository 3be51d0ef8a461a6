import sys, os; ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tests'))
import numpy as np, draco_sharp_amd as dsa, draco_sharp_amd.synth as synth
n = int(sys.argv[1]) if len(sys.argv)>1 else 512
blob, offs = synth.make_batch(synth.GRID,128,256,1000,n)
ctx = dsa.Context(0); ctx.set_profiling(True)
b = dsa.Batch(ctx, blob=blob, offsets=offs)
for _ in range(2): b.decode()
print({k: round(v,2) for k,v in b.stage_times().items()})
d = np.array([b.debug_array(i,4,np.uint32,20) for i in range(0,n,max(1,n//16))])
print("dbg clocks (median over sampled meshes) x1e6:", np.round(np.median(d,axis=0)/1e6,2)); print("raw median:", np.median(d,axis=0).astype(int).tolist())
a = np.array([b.debug_array(i,5,np.uint32,64)[:12].reshape(3,4) for i in range(0,n,max(1,n//8))])
print("attr (source, nsym, precision, rans bytes) mesh0:", a[0].tolist()); print("nsym range per attr:", a[:,:,1].min(0), a[:,:,1].max(0))
m = np.median(d, axis=0)
print("shader clock GHz (s_memtime / s_memrealtime): traverse %.3f connectivity %.3f" % (m[6] / m[17] * 0.1, m[13] / m[15] * 0.1))
