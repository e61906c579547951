#!/usr/bin/env python3
"""In-kernel timeline of K3 (diagnostic build: `make -C a-nice-rag_amd/csrc dbg`, loaded through ANRAG_LIB).
Phases per workgroup from 100 MHz wall-clock stamps; prints the median / max over workgroups, averaged over queries.
usage: ANRAG_LIB=a-nice-rag_amd/libanrag_dbg.so python scripts/k3_timeline.py [n_docs]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import _native as nat
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
dev = torch.device("cuda:0")
post = synth.bm25_postings(n, 200_000, 777, dev)
idf = synth.bm25_idf(post["df"].cpu().numpy(), n)
terms = synth.bm25_queries(post, 64, 99)
torch.cuda.synchronize()
idx = Index(0)
idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
              (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], float(post["total_len"]) / n,
              synth.BM25_K1, synth.BM25_B)
T = torch.full((64, 16), -1, dtype=torch.int32, device=dev)
for i, t in enumerate(terms):
    T[i, : len(t)] = torch.from_numpy(np.asarray(t, np.int32)).to(dev)
nt = [len(t) for t in terms]
out = torch.zeros((64, 25, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib = nat.load_library()
lib.anrag_debug_k3_stamps.argtypes = [C.c_void_p, C.c_int]
n_parts = (n + 4095) // 4096 if n >= 256 * 4096 else None
names = ["table", "gather", "apply", "bound (group bests + sort)", "survivors", "final sort + store"]
acc = []
for q in range(64):
    for rep in range(3):
        nat.check(lib.anrag_bm25_search_device(idx.handle, T[q].data_ptr(), nt[q], 25, None, out[q].data_ptr()))
        idx.sync()
    buf = np.zeros(4096 * 12, dtype=np.uint64)
    assert lib.anrag_debug_k3_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(4096, 12).astype(np.int64)
    live = st[:, 0] > 0
    st = st[live]
    t0 = st[:, 0].min()
    acc.append(np.concatenate([[np.median(st[:, 0] - t0), (st[:, 0] - t0).max()],
                               np.median(np.diff(st[:, :7], axis=1), axis=0),
                               [np.median(st[:, 6] - t0), (st[:, 6] - t0).max(), live.sum(),
                                np.median(st[:, 8] - st[:, 1]), np.median(st[:, 9] - st[:, 8]), np.median(st[:, 2] - st[:, 9]),
                                np.median(st[:, 10] - st[:, 1]), np.median(st[:, 11] - st[:, 1]),
                                np.median(st[:, 7] / np.maximum(st[:, 6] - st[:, 0], 1))]]))
a = np.mean(acc, axis=0) / 100.0  # 100 MHz -> us
print(f"shader clock inside the kernel: {np.mean(acc, axis=0)[-1] * 100:.0f} MHz")
a = np.append(a, 0)
print(f"slot table: 16 lanes {a[15]:.2f} us, then the barrier {a[14] - a[15]:.2f} us")
print(f"gather = slot table {a[14]:.2f} + issue of the loads {a[11] - a[14]:.2f} + loads in flight {a[12]:.2f} us")
print(f"n_docs={n}, workgroups {int(np.mean(acc, axis=0)[10])}: start skew median {a[0]:.2f} max {a[1]:.2f} us")
for i, nm in enumerate(names):
    print(f"  {nm:32s} {a[2 + i]:6.2f} us (median over workgroups)")
print(f"  end of workgroup: median {a[8]:.2f} us, last {a[9]:.2f} us after the first workgroup's start")
