import sys, time
sys.path.insert(0, "/root/repo")
import numpy as np, icp_amd
from icp_amd import synth
P, Q = synth.make_batch(64, 1000, seed=5)
P, Q = np.tile(P, (64, 1, 1)), np.tile(Q, (64, 1, 1))
for k in range(4):
    t = time.perf_counter(); out = icp_amd.icp_gpu_batch(P, Q, 20); dt = time.perf_counter() - t
    print("icp_gpu_batch 4096 x 1000 x 1000 x 20, host buffers in and out: %.1f ms" % (dt * 1e3), flush=True)
b = icp_amd.Batch(4096, 1000, 1000)
for k in range(3):
    t = time.perf_counter(); b.upload(P, Q); t1 = time.perf_counter(); ms = b.run(20, timed=True); t2 = time.perf_counter(); o = b.download(); t3 = time.perf_counter()
    print("upload %.1f ms, run %.1f ms (kernel %.1f), download %.1f ms" % ((t1 - t) * 1e3, (t2 - t1) * 1e3, ms, (t3 - t2) * 1e3), flush=True)
b.close()
