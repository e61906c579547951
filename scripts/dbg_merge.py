import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anrag import synth, _native as nat
from anrag.index import Index
N, D, k = 1000000, 768, 25
dev = torch.device("cuda", 0)
E = synth.dense_corpus(N, D, 1, dev)
Q, planted = synth.dense_queries(E, 4, 2)
half = N // 2
lib = nat.load_library()
a, b, util, whole = Index(0), Index(0), Index(0), Index(0)
whole.dense_load((E.data_ptr(), N, D))
a.dense_load((E.data_ptr(), half, D), doc_id_base=0)
b.dense_load((E.data_ptr() + half * D * 4, N - half, D), doc_id_base=half)
lists = torch.zeros((2, k, 2), dtype=torch.int64, device=dev)
out = torch.zeros((k, 2), dtype=torch.int64, device=dev)
a.dense_search_device(Q[0].data_ptr(), 1, k, 0, lists[0].data_ptr())
b.dense_search_device(Q[0].data_ptr(), 1, k, 0, lists[1].data_ptr())
a.sync(); b.sync()
L = lists.cpu().numpy()
print("planted", int(planted[0]))
print("list a", L[0, :, 1], L[0, :, 0].copy().view(np.float64))
print("list b", L[1, :, 1], L[1, :, 0].copy().view(np.float64))
nat.check(lib.anrag_merge_candidates_device(util.handle, lists.data_ptr(), 2, k, k, out.data_ptr()))
util.sync()
o = out.cpu().numpy()
print("merged", o[:, 1], o[:, 0].copy().view(np.float64))
print("whole", whole.dense_search(Q[0].cpu().numpy(), k)[0])
