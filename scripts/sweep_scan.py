#!/usr/bin/env python3
"""K1 at several shapes in ONE process: 1-query launches and 8-query launches, every launch bracketed (HIP events).
The kernel variant is picked by the library from env knobs (ANRAG_SCAN_R), so run once per variant.
usage: python scripts/sweep_scan.py [dim]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from anrag import _native as nat, synth
from anrag.index import Index

d = int(sys.argv[1]) if len(sys.argv) > 1 else 768
dev = torch.device("cuda:0")
E = synth.dense_corpus(1_000_000, d, 1234, dev)
Q, _ = synth.dense_queries(E, 64, 4321)
torch.cuda.synchronize()
out = torch.zeros((64, 10, 2), dtype=torch.int64, device=dev)
torch.cuda.synchronize()
lib = nat.load_library()
res = []
for rows in (1_000_000, 500_000, 125_000, 100_000):
    with Index(0) as idx:
        idx.dense_load((E.data_ptr(), rows, d))
        for group in (1, 8):
            def run(n):
                for i in range(0, n, group):
                    nat.check(lib.anrag_dense_search_device(idx.handle, Q[i % 64].data_ptr(), group, 10, None,
                                                            out[i % 64].data_ptr()))
                idx.sync()
            run(64)
            idx.profile(True, kernels=[nat.KERNEL_DENSE_SCAN], every=1)
            idx.profile_reset()
            run(512 if rows < 500_000 else 128)
            ms, n = idx.profile_read(nat.KERNEL_DENSE_SCAN)
            idx.profile(False)
            per_q = ms / n / group * 1e3
            res.append(f"{rows // 1000}k/{group}: {per_q:.1f} us ({rows * d * 4 / per_q / 1e6 / 8000 * 100:.1f}%)")
print(f"R={os.environ.get('ANRAG_SCAN_R', 'default')} d={d}  " + "  ".join(res))
