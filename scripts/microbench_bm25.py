#!/usr/bin/env python3
"""K3 alone: BM25 queries over synthetic postings of bench.py's shape, operands resident in HBM.
Reports the kernel's mean duration against its algorithmic bytes (SURVEY.md 8d restated for this layout:
sum over the query's terms of df(t) * 12 B -- a posting is an int32 doc id + an fp64 impact).
usage: python scripts/microbench_bm25.py [n_docs] [iters] [queries per launch: 1 (default) or up to 8]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import _native as nat
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 400
group = int(sys.argv[3]) if len(sys.argv) > 3 else 1
dev = torch.device("cuda:0")
post = synth.bm25_postings(n, 200_000, 777, dev)
df = post["df"].cpu().numpy()
idf = synth.bm25_idf(df, n)
terms = synth.bm25_queries(post, 64, 99, n_terms=int(os.environ.get("N_TERMS", "9")))
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
torch.cuda.synchronize()  # torch fills / uploads on its own stream; the library reads and writes on its streams
torch.cuda.synchronize()
lib = nat.load_library()
alg = np.mean([sum(int(df[t]) for t in tl) * 12 for tl in terms])


import ctypes as C

PT = (C.c_void_p * 64)(*[T[q].data_ptr() for q in range(64)])
PO = (C.c_void_p * 64)(*[out[q].data_ptr() for q in range(64)])
PN = (C.c_int32 * 64)(*nt)


def step(i):
    q = i % 64
    if group == 1:
        nat.check(lib.anrag_bm25_search_device(idx.handle, T[q].data_ptr(), nt[q], 25, None, out[q].data_ptr()))
    elif q % group == 0:  # the group [q, q + group) in one call (one K3 launch, one tail launch)
        off = q
        nat.check(lib.anrag_bm25_search_group_device(
            idx.handle, C.cast(C.byref(PT, off * 8), C.c_void_p), C.cast(C.byref(PN, off * 4), C.c_void_p),
            min(group, 64 - q), 25, None, C.cast(C.byref(PO, off * 8), C.c_void_p)))


for i in range(20):
    step(i)
idx.sync()
idx.profile(True, kernels=[nat.KERNEL_BM25], every=4)
idx.profile_reset()
t0 = time.perf_counter()
for i in range(iters):
    step(i)
idx.sync()
wall = (time.perf_counter() - t0) / iters
ms, launches = idx.profile_read(nat.KERNEL_BM25)
units = idx.profile_units(nat.KERNEL_BM25)
k_us = ms / max(units, 1) * 1e3  # per QUERY (a launch carries `group` of them)
print(f"n_docs={n}, {group} per launch: {1/wall:.0f} q/s ({wall*1e6:.1f} us/query); K3 {k_us:.1f} us/query "
      f"({ms / max(launches, 1) * 1e3:.1f} us/launch); algorithmic "
      f"{alg/1e6:.2f} MB/query -> {alg/(k_us*1e-6)/1e9:.0f} GB/s ({alg/(k_us*1e-6)/8e12*100:.1f}% of 8 TB/s); "
      f"mean terms {np.mean(nt):.1f}, mean sum df {alg/12:.0f}")
