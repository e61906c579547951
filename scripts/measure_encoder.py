import sys, time
sys.path.insert(0, "/root/repo")
import torch
from anrag.encoder import LocalEncoder
enc = LocalEncoder()
qs = ["what dose of asthma inhaler for adults with chronic kidney disease %d" % i for i in range(64)]
import numpy as np
for mode in (False, True):
    enc.use_graphs = mode
    for q in qs[:8]: enc.encode_query(q)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    outs = [enc.encode_query(q) for q in qs]
    dt = (time.perf_counter() - t0) / len(qs)
    print("encode_query (%s): %.2f ms/query" % ("hipGraph replay" if mode else "eager", dt * 1e3))
    if mode: print("max |graph - eager| = %.2e; graphs captured for lengths %s" % (float(np.max(np.abs(np.stack(outs) - ref))), sorted(k for k, v in enc._graphs.items() if v)))
    else: ref = np.stack(outs)
t0 = time.perf_counter(); e = enc.encode(qs * 16, batch_size=256); dt = time.perf_counter() - t0
print("encode 1024 short texts: %.1f ms total, %.0f texts/s" % (dt * 1e3, 1024 / dt))
