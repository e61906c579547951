import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from anrag import synth, _native as nat
from anrag.index import Index
n, d, TOPN = 1_000_000, 768, 10
dev = torch.device("cuda", 0)
E = synth.dense_corpus(n, d, 1234, dev)
Qb, _ = synth.dense_queries_global(n, d, 256, 4322, 1234, dev)
torch.cuda.synchronize()
idx = Index(0); idx.dense_load((E.data_ptr(), n, d))
lib = nat.load_library()
outb = torch.zeros((256, TOPN, 2), dtype=torch.int64, device=dev)
flag = torch.zeros(256, dtype=torch.int32, device=dev)
ref = torch.zeros((4, TOPN, 2), dtype=torch.int64, device=dev)
nat.check(lib.anrag_dense_search_device(idx.handle, Qb.data_ptr(), 4, TOPN, None, ref.data_ptr()))
idx.sync()
tk = (Qb[:4] @ E.T).topk(TOPN, dim=1)
print("K1 ref ids == torch:", torch.equal(ref[:, :, 1], tk.indices))
def cmp(tag):
    print(tag, "K2 ids == K1 ref:", torch.equal(outb[:4, :, 1], ref[:, :, 1]), " K2 ids == torch:", torch.equal(outb[:4, :, 1], tk.indices),
          " ref still == torch:", torch.equal(ref[:, :, 1], tk.indices), flush=True)
for mode in ("f32", "bf16x3"):
    idx.set_batched_precision(mode)
    def run(k):
        for _ in range(k):
            nat.check(lib.anrag_dense_search_batch_device(idx.handle, Qb.data_ptr(), 256, TOPN, None, outb.data_ptr(), flag.data_ptr()))
        idx.sync()
    run(2); cmp(mode + " after 2")
    run(30); cmp(mode + " after 30")
    idx.profile(True, kernels=[nat.KERNEL_DENSE_BATCHED], every=1); idx.profile_reset()
    run(60); cmp(mode + " after 60 profiled")
    print(idx.profile_read(nat.KERNEL_DENSE_BATCHED)); idx.profile(False)
