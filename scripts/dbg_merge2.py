import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anrag import synth, _native as nat
from anrag.index import Index
N, D, k = 1000000, 768, 25
dev = torch.device("cuda", 0)
E = synth.dense_corpus(N, D, 1234, dev)
Q, planted = synth.dense_queries(E, 32, 4321)
print("planted", planted[:4].tolist())
idx = Index(0); idx.dense_load((E.data_ptr(), N, D))
qh = Q.cpu().numpy()
d1, s1, c1 = idx.dense_search(qh[:8], 10)
d2, s2, c2 = idx.dense_search(qh, 10)
print("K1 top1", d1[:4, 0], "K2 top1", d2[:4, 0])
half = N // 2
lib = nat.load_library()
a, b, util = Index(0), Index(0), Index(0)
a.dense_load((E.data_ptr(), half, D), doc_id_base=0)
b.dense_load((E.data_ptr() + half * D * 4, N - half, D), doc_id_base=half)
lists = torch.zeros((2, k, 2), dtype=torch.int64, device=dev)
out = torch.zeros((k, 2), dtype=torch.int64, device=dev)
for qi in range(4):
    a.dense_search_device(Q[qi].data_ptr(), 1, k, 0, lists[0].data_ptr())
    b.dense_search_device(Q[qi].data_ptr(), 1, k, 0, lists[1].data_ptr())
    a.sync(); b.sync()
    L = lists.cpu().numpy()
    nat.check(lib.anrag_merge_candidates_device(util.handle, lists.data_ptr(), 2, k, k, out.data_ptr()))
    util.sync()
    whole = idx.dense_search(Q[qi].cpu().numpy(), k)[0][0]
    o = out.cpu().numpy()
    print(qi, "a0", L[0, 0, 1], "b0", L[1, 0, 1], "merged0", o[0, 1], "whole0", whole[0], "eq", o[:, 1].tolist() == whole.tolist())
