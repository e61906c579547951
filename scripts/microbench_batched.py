#!/usr/bin/env python3
"""On-box timing of K2 (batched queries, fp32 MFMA): python scripts/microbench_batched.py [n_rows] [dim] [nq] [k]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anrag import synth, _native as nat
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
d = int(sys.argv[2]) if len(sys.argv) > 2 else 768
nq = int(sys.argv[3]) if len(sys.argv) > 3 else 256
k = int(sys.argv[4]) if len(sys.argv) > 4 else 10
prec = sys.argv[5] if len(sys.argv) > 5 else "f32"
dev = torch.device("cuda", 0)
E = synth.dense_corpus(n, d, 1234, dev)
Q, rows = synth.dense_queries(E, nq, 4321)
torch.cuda.synchronize()
idx = Index(0); idx.dense_load((E.data_ptr(), n, d))
idx.set_batched_precision(prec)
lib = nat.load_library()
out = torch.zeros((nq, k, 2), dtype=torch.int64, device=dev)
flag = torch.zeros(nq, dtype=torch.int32, device=dev)
torch.cuda.synchronize()  # torch fills on its own stream; the library writes on its streams
def run():
    nat.check(lib.anrag_dense_search_batch_device(idx.handle, Q.data_ptr(), nq, k, None, out.data_ptr(), flag.data_ptr()))
run(); idx.sync()
ref = (Q @ E.T).topk(k, dim=1)
got = out.cpu().numpy()
print("flags:", int(flag.abs().sum().item()), " ids match torch:", np.array_equal(got[:, :, 1], ref.indices.cpu().numpy()),
      " max|dscore|:", float(np.abs(got[:, :, 0].copy().view(np.float64) - ref.values.cpu().numpy()).max()))
idx.profile(True, kernels=[nat.KERNEL_DENSE_BATCHED]); idx.profile_reset()
iters = int(os.environ.get("ITERS", "10"))
t0 = time.perf_counter()
for _ in range(iters): run()
idx.sync(); wall = (time.perf_counter() - t0) / iters
got2 = out.cpu().numpy()
print("after the timed loop: flags:", int(flag.abs().sum().item()), " ids match torch:", np.array_equal(got2[:, :, 1], ref.indices.cpu().numpy()),
      " same as the first pass:", np.array_equal(got2, got))
ms, launches = idx.profile_read(nat.KERNEL_DENSE_BATCHED)
ms /= launches
flop = 2.0 * nq * n * d
print(f"[{prec}] n={n} d={d} nq={nq} k={k}: wall/pass {wall*1e3:.3f} ms ({nq/wall:.0f} q/s)  GEMM passes (sample+filter) {ms:.3f} ms "
      f"-> {flop/ms/1e9:.1f} TFLOP/s on the full pass alone est., {flop/(ms*1e-3)/1e12/157.3*100:.1f}% of 157.3 TF")
