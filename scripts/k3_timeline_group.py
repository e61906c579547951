#!/usr/bin/env python3
"""In-kernel timeline of K3 at 8 queries per launch (diagnostic build: `make -C a-nice-rag_amd/csrc dbg`, loaded through
ANRAG_LIB; ANRAG_BM25_FORM=tall|wide picks the form).  Per workgroup: start, phase ends, end (100 MHz wall clock).
Prints the phase medians, the workgroups' lifetimes against their number in flight, and the launch's span.
usage: ANRAG_LIB=a-nice-rag_amd/libanrag_dbg.so python scripts/k3_timeline_group.py [n_docs] [group]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from anrag import _native as nat
from anrag import synth
from anrag.index import Index

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
group = int(sys.argv[2]) if len(sys.argv) > 2 else 8
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
PT = (C.c_void_p * 64)(*[T[q].data_ptr() for q in range(64)])
PO = (C.c_void_p * 64)(*[out[q].data_ptr() for q in range(64)])
PN = (C.c_int32 * 64)(*nt)
rows = []
for q0 in range(0, 64, group):
    for rep in range(3):
        if group == 1:
            nat.check(lib.anrag_bm25_search_device(idx.handle, T[q0].data_ptr(), nt[q0], 25, None, out[q0].data_ptr()))
        else:
            nat.check(lib.anrag_bm25_search_group_device(
                idx.handle, C.cast(C.byref(PT, q0 * 8), C.c_void_p), C.cast(C.byref(PN, q0 * 4), C.c_void_p), group, 25, None,
                C.cast(C.byref(PO, q0 * 8), C.c_void_p)))
        idx.sync()
    buf = np.zeros(4096 * 12, dtype=np.uint64)
    assert lib.anrag_debug_k3_stamps(buf.ctypes.data, buf.size) == 0
    st = buf.reshape(4096, 12).astype(np.int64)
    st = st[st[:, 0] > 0]
    t0 = st[:, 0].min()
    start, end = (st[:, 0] - t0) / 100.0, (st[:, 6] - t0) / 100.0
    life = end - start
    # workgroups in flight at each workgroup's start
    order = np.argsort(start)
    ends_sorted = np.sort(end)
    inflight = np.array([np.sum((start <= s) & (end > s)) for s in start])
    ph = np.diff(st[:, :7], axis=1) / 100.0
    rows.append(dict(wgs=len(st), span=end.max(), life_med=np.median(life), life_p90=np.percentile(life, 90),
                     inflight_med=np.median(inflight), inflight_max=inflight.max(), phases=np.median(ph, axis=0),
                     clock=np.median(st[:, 7] / np.maximum(st[:, 6] - st[:, 0], 1)) * 100,
                     first_wave_life=np.median(life[start < 1.0]) if np.any(start < 1.0) else float("nan"),
                     late_life=np.median(life[start > np.percentile(start, 50)])))
names = ["table", "gather", "apply", "bound", "survivors", "ranks + store"]
m = {k: np.mean([r[k] for r in rows], axis=0) for k in rows[0]}
print(f"n_docs={n}, {group} queries per launch, form {os.environ.get('ANRAG_BM25_FORM', 'default')}: {m['wgs']:.0f} workgroups, "
      f"launch span {m['span']:.1f} us = {m['span'] / group:.2f} us per query; shader clock {m['clock']:.0f} MHz")
print(f"workgroup lifetime: median {m['life_med']:.2f} us, p90 {m['life_p90']:.2f}; of the first wave of workgroups {m['first_wave_life']:.2f}, "
      f"of the later half {m['late_life']:.2f}; workgroups in flight at a workgroup's start: median {m['inflight_med']:.0f}, max {m['inflight_max']:.0f}")
print("phases (median over workgroups, us): " + ", ".join(f"{nm} {v:.2f}" for nm, v in zip(names, m["phases"])))
