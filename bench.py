#!/usr/bin/env python3
"""Headline benchmark: hybrid retrieval queries/sec on a 1M x 768 corpus (BASELINE.json configs[2]).

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

`--gpus N` with N > 1 and no launcher around it (WORLD_SIZE unset) starts the N ranks ITSELF: N fresh child
processes of this script, one per device, before this process has made any GPU call; rank 0's JSON line is
relayed.  Fewer than N visible devices is an error (exit 2), never a silently smaller job.

A step = ONE query at batch=1 through the whole hot path: dense dot-product scan + top-25, BM25
term-at-a-time + top-25, weighted RRF (5:1, k=40), top-10 -- the body of the reference's
`retrieve_documents` for one dense model + BM25 (src/query_rag_retrieval.py:197-220, :304-378).
Inputs (corpus matrix, postings, queries, term ids) are resident in HBM before the timed region;
queries are enqueued back to back with no host synchronisation in between.

N > 1: the SAME corpus row-sharded N ways -- the synthetic corpus is a function of (seed, global row)
(a-nice-rag_amd/synth.py), so the union of the shards is bit-identical to the N = 1 corpus and the queries
are the same: per-rank legs, one RCCL all-gather of 2 x 25 candidate records per rank and query on a
communication stream, replicated merge + fusion (a-nice-rag_amd/sharded.py).  After the timed region rank 0
builds a SINGLE index over the whole corpus and checks the sharded answers of 8 queries against it
(`sharded_matches_single`).  Default: strong scaling of the 1M-row corpus (the metric's "1M x 768 corpus, 1/2/4/8
MI355X"); `--rows-per-gpu R` = weak scaling (C5: `--rows-per-gpu 1000000 --dim 1024` on 8 GPUs = 8M x 1024).

Prints ONE JSON line on rank 0 (fields of the build contract + `roofline` + `cpu_baseline` + `also`: the C2 / C4 /
K3 / tail measurements of the same run, each with what it takes to recompute its roofline fraction).
Exit code 1 when a parity check fails (GPU vs CPU port, sharded vs single index).
"""
from __future__ import annotations

import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (guides/MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 measured copy)
F32_MFMA_PEAK_TF = 157.3
BF16_MFMA_PEAK_TF = 2500.0
CORPUS_SEED, QUERY_SEED, BM25_SEED, BM25_QUERY_SEED = 1234, 4321, 777, 99
W_DENSE, W_BM25, WRRF_K = 5.0, 1.0, 40.0  # src/config.py:30-36, retrieval_eval.py:279
MAX_TERMS = 16


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=400)
    ap.add_argument("--warmup", type=int, default=40)
    ap.add_argument("--rows", type=int, default=1_000_000, help="corpus rows (whole job): strong scaling")
    ap.add_argument("--rows-per-gpu", type=int, default=0,
                    help="weak scaling: corpus rows = this x ranks (C5: 1000000 with --dim 1024 on 8 GPUs)")
    ap.add_argument("--dim", type=int, default=768)
    ap.add_argument("--vocab", type=int, default=200_000)
    ap.add_argument("--similarity-k", type=int, default=25)
    ap.add_argument("--top-n", type=int, default=10)
    ap.add_argument("--queries", type=int, default=64, help="distinct synthetic queries cycled through")
    ap.add_argument("--workload", choices=["hybrid", "dense", "batched"], default="hybrid",
                    help="hybrid = headline (C3); dense = K1 only (C2-shaped); batched = 256-query MFMA passes (C4)")
    ap.add_argument("--batch", type=int, default=256, help="queries per pass of --workload batched")
    ap.add_argument("--batch-precision", choices=["f32", "bf16x3"], default="f32",
                    help="--workload batched: exact f32 MFMA (default) or opt-in split-precision bf16 x 3 products")
    ap.add_argument("--exchange-group", type=int, default=8,
                    help="N > 1: in-flight queries that share one all-gather (each is still scanned alone)")
    ap.add_argument("--filter", action="store_true",
                    help="diagnostic: source-prefix filter as retrieval_eval.py:280 passes it ('CG,NG'); rows "
                         "carry one of 300 source ids, 15 %% of them outside the filter (SURVEY.md 8d)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-queries", type=int, default=12, help="queries of the bounded CPU sample")
    ap.add_argument("--no-also", action="store_true", help="skip the C2 / C4 / K3 side measurements")
    ap.add_argument("--check-queries", type=int, default=8,
                    help="N > 1: queries of the sharded answer checked against a single index on rank 0 (0 = off)")
    ap.add_argument("--launch-selftest", action="store_true",
                    help="test hook (tests/test_bench_launcher.py): the ranks run on the CPU over gloo and only "
                         "check the launcher, the rendezvous and the block-seeded corpus generator")
    return ap.parse_args(argv)


# ====================================================================== launcher
def _free_port() -> int:
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args) -> int:
    """Start `--gpus` ranks of this script as CHILD processes (never an exec of a process that has touched the GPU:
    this parent makes no HIP call at all), relay rank 0's stdout, return the job's exit code."""
    n = args.gpus
    if not args.launch_selftest:
        import torch  # device_count() does not initialise the GPU on this image; nothing else is called here

        have = torch.cuda.device_count()
        if have < n:
            print(f"[bench] --gpus {n} but only {have} GPU(s) visible: refusing to run a smaller job",
                  file=sys.stderr, flush=True)
            return 2
    port = _free_port()
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this pool
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else sys.stderr, text=(r == 0)))
    import threading

    relayed = []
    reader = threading.Thread(target=lambda: relayed.append(procs[0].stdout.read()), daemon=True)
    reader.start()  # drain rank 0's pipe while it runs: a full pipe would block it
    failed = 0
    while any(p.poll() is None for p in procs):
        for p in procs:
            rc = p.poll()
            if rc not in (None, 0) and not failed:
                failed = rc
        if failed:  # a dead rank leaves the others waiting in a collective: end them (exact PIDs)
            time.sleep(2.0)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            break
        time.sleep(0.2)
    for p in procs:
        p.wait()
        if p.returncode != 0 and not failed:
            failed = p.returncode
    reader.join(timeout=10.0)
    sys.stdout.write("".join(relayed))
    sys.stdout.flush()
    if failed:
        print(f"[bench] a rank exited with code {failed}", file=sys.stderr, flush=True)
        return failed if 0 < failed < 256 else 1
    return 0


def launch_selftest_rank(args) -> int:
    """CPU/gloo body of --launch-selftest: rendezvous of the launched ranks + shards of the block-seeded corpus
    against the whole (no kernels: the product has no CPU path)."""
    import torch
    import torch.distributed as dist

    from anrag import synth
    from anrag.sharded import shard_bounds

    world, rank = int(os.environ["WORLD_SIZE"]), int(os.environ["RANK"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rows, dim = args.rows, args.dim
    lo, hi = shard_bounds(rows, world, rank)
    shard = synth.dense_corpus(hi - lo, dim, CORPUS_SEED, "cpu", row_lo=lo)
    sums = torch.zeros(rows, dtype=torch.float64)
    sums[lo:hi] = shard.double().sum(dim=1)
    dist.all_reduce(sums)
    ok = True
    if rank == 0:
        whole = synth.dense_corpus(rows, dim, CORPUS_SEED, "cpu")
        ok = bool(torch.equal(sums, whole.double().sum(dim=1)))
        print(json.dumps({"selftest": "launcher", "n_gpus": dist.get_world_size(), "gpus_arg": args.gpus,
                          "rows": rows, "union_matches_single": ok}), flush=True)
    dist.barrier()
    dist.destroy_process_group()
    return 0 if ok else 1


# ====================================================================== one rank
def main() -> int:
    args = parse_args()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args)
    if args.launch_selftest:
        return launch_selftest_rank(args)
    return run_rank(args)


def run_rank(args) -> int:
    import torch
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        print(f"[bench] --gpus {args.gpus} but WORLD_SIZE={world}: launch with --nproc-per-node {args.gpus} "
              "(or without a launcher: bench.py starts its own ranks)", file=sys.stderr, flush=True)
        return 2
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: libanrag has no CPU path")
    if local_rank >= torch.cuda.device_count():
        if torch.cuda.device_count() == 1:
            local_rank = 0  # the launcher narrowed this rank's visibility to one device
        else:
            print(f"[bench] LOCAL_RANK {local_rank} but {torch.cuda.device_count()} devices visible",
                  file=sys.stderr, flush=True)
            return 2
    # ANRAG_FORCE_SHARDED=1: take the sharded (all-gather) route even at world size 1 -- a rehearsal of the
    # N > 1 code path on a one-GPU box
    sharded = world > 1 or os.environ.get("ANRAG_FORCE_SHARDED") == "1"
    if sharded:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=rank, world_size=world,
                                device_id=torch.device("cuda", local_rank))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)

    from anrag import _native as nat
    from anrag import synth
    from anrag.index import Index
    from anrag.sharded import HipShardEngine, ShardedSearcher, shard_bounds

    K, TOPN = args.similarity_k, args.top_n
    hybrid = args.workload == "hybrid"
    batched = args.workload == "batched"
    if batched:
        args.queries = args.batch
    weak = args.rows_per_gpu > 0
    if weak:
        args.rows = args.rows_per_gpu * world
    lo, hi = shard_bounds(args.rows, world, rank)
    n_local = hi - lo

    # ------------------------------------------------------------------ synthetic shard, resident in HBM
    t_build = time.time()
    E = synth.dense_corpus(n_local, args.dim, CORPUS_SEED, device, row_lo=lo)  # rows [lo, hi) of the global corpus
    idx = Index(local_rank)
    torch.cuda.synchronize()  # device-pointer operands must be complete: the library copies on its own stream
    src = allowed_rows = d_allow = allow = None
    if args.filter:
        if batched:
            raise SystemExit("--filter is a batch=1 diagnostic")
        src = ((np.arange(n_local, dtype=np.int64) + lo) % 300).astype(np.uint16)  # 300 guideline codes, cyclic
        allow = np.zeros(65536, np.uint8)
        allow[45:300] = 1  # ids 0..44 = the ~15 % QS/TA/PH sources the 'CG,NG' filter drops
        allowed_rows = allow[src].astype(bool)
        # 2048 words, bit (s % 32) of word s / 32 = source s allowed
        d_allow = torch.from_numpy(np.packbits(allow, bitorder="little").view(np.int32).copy()).to(device)
    idx.dense_load((E.data_ptr(), n_local, args.dim), source_id=src, doc_id_base=lo)
    if batched:
        idx.set_batched_precision(args.batch_precision)
    # queries: planted next to GLOBAL rows, computed alike on every rank (the broadcast is a belt to those braces)
    Q, planted = synth.dense_queries_global(args.rows, args.dim, args.queries, QUERY_SEED, CORPUS_SEED, device)
    if world > 1:
        dist.broadcast(Q, 0)
    post = idf = avgdl = None
    term_lists = [np.zeros(0, np.int32)] * args.queries
    if hybrid:
        post = synth.bm25_postings(n_local, args.vocab, BM25_SEED, device, doc_lo=lo)
        df = post["df"].clone()
        tot = torch.tensor([post["total_len"]], device=device, dtype=torch.int64)
        if world > 1:  # GLOBAL statistics, replicated (sharded.py docstring)
            dist.all_reduce(df)
            dist.all_reduce(tot)
        avgdl = int(tot.item()) / args.rows
        idf = synth.bm25_idf(df.cpu().numpy(), args.rows)
        torch.cuda.synchronize()
        idx.bm25_load(post["indptr"], (post["post_doc"].data_ptr(), post["post_doc"].numel()),
                      (post["post_tf"].data_ptr(), post["post_tf"].numel()), idf, post["doc_len"], avgdl,
                      synth.BM25_K1, synth.BM25_B, source_id=src, doc_id_base=lo)
        term_lists = synth.bm25_queries_global(args.rows, args.vocab, args.queries, BM25_QUERY_SEED, BM25_SEED, device)
    # query term ids as one padded device tensor
    T = torch.full((args.queries, MAX_TERMS), -1, dtype=torch.int32, device=device)
    NT = torch.zeros(args.queries, dtype=torch.int32, device=device)
    for i, t in enumerate(term_lists):
        t = np.asarray(t, dtype=np.int32)[:MAX_TERMS]
        if len(t):
            T[i, : len(t)] = torch.from_numpy(t).to(device)
        NT[i] = len(t)
    if world > 1:
        dist.broadcast(T, 0)
        dist.broadcast(NT, 0)
    n_terms = [int(x) for x in NT.cpu().tolist()]
    term_lists = [T[i, : n_terms[i]].cpu().numpy() for i in range(args.queries)]
    torch.cuda.synchronize()
    build_s = time.time() - t_build

    # ------------------------------------------------------------------ the step
    lib = nat.load_library()
    out = cnt = searcher = None
    if not sharded:
        out = torch.zeros((args.queries, TOPN, 2), dtype=torch.int64, device=device)
        cnt = torch.zeros(args.queries, dtype=torch.int32, device=device)
        torch.cuda.synchronize()  # (torch fills on its own stream; the library writes on its streams)
        allow_ptr = d_allow.data_ptr() if d_allow is not None else None

        def step(i):
            qi = i % args.queries
            if hybrid:
                nat.check(lib.anrag_hybrid_search_device(
                    idx.handle, Q[qi].data_ptr(), T[qi].data_ptr(), n_terms[qi], K, W_DENSE, W_BM25, WRRF_K, TOPN,
                    allow_ptr, allow_ptr, out[qi].data_ptr(), cnt[qi:].data_ptr()))
            elif batched:  # one step = one pass of args.batch queries
                nat.check(lib.anrag_dense_search_batch_device(idx.handle, Q.data_ptr(), args.batch, TOPN, None,
                                                              out.data_ptr(), cnt.data_ptr()))
            else:
                nat.check(lib.anrag_dense_search_device(idx.handle, Q[qi].data_ptr(), 1, TOPN, allow_ptr,
                                                        out[qi].data_ptr()))

        def finish():
            idx.sync()
    else:
        if not hybrid:
            raise SystemExit("--workload dense / batched are single-GPU diagnostics")
        engine = HipShardEngine(idx, device)
        searcher = ShardedSearcher(engine, k=K, top_n=TOPN, w_dense=W_DENSE, w_bm25=W_BM25, wrrf_k=WRRF_K,
                                   depth=4, group=args.exchange_group, device=device)
        if args.filter:
            searcher.set_filter(allow[:300], allow[:300])  # source ids are global: (row % 300) on every rank

        def step(i):
            qi = i % args.queries
            return searcher.submit(Q[qi], T[qi], n_terms[qi])

        def finish():
            searcher.drain()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i)
    if sharded and args.warmup == 0:
        step(0)  # the first all-gather sets the collective up: never inside the timed region
    finish()
    # HIP events around the dominant kernel only, on a SAMPLE of its launches, on the stream it is launched on: a
    # bracket holds its stream for ~10 us, which at batch=1 would be charged to every query (measured: 463 -> 452
    # us/step at 1M rows; a sharded K1 launch carries a whole exchange group and paid 10 us per 2 launches = 2 % at the
    # 125k-row shard): every 8th launch here, every launch in the separate pass below.
    main_kernel = nat.KERNEL_DENSE_BATCHED if batched else nat.KERNEL_DENSE_SCAN
    idx.profile(True, kernels=[main_kernel], every=1 if batched else 8)
    idx.profile_reset()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i)
    finish()
    barrier()
    elapsed = time.perf_counter() - t0
    scan_ms, scan_n = idx.profile_read(main_kernel)
    scan_units = idx.profile_units(main_kernel)  # queries the timed launches carried (sharded: a group per K1 launch)
    sampled = "every %s launch of the timed region" % ("" if batched else "8th")
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    # a short timed region leaves a handful of bracketed launches: add a separate pass, outside the timed region,
    # in which EVERY launch of the kernel is bracketed, so that the roofline never rests on three samples
    MIN_SAMPLES = 32
    if scan_n < MIN_SAMPLES and not batched:
        extra = 64 * (args.exchange_group if sharded else 1)
        idx.profile(True, kernels=[main_kernel], every=1)
        for i in range(extra):
            step(i)
        finish()
        barrier()
        scan_ms, scan_n = idx.profile_read(main_kernel)
        scan_units = idx.profile_units(main_kernel)
        sampled += " + every launch of a %d-query pass after it" % extra
    idx.profile(False)

    # single-query latency (SURVEY.md 8d asks for p50/p99 next to the throughput): one query in flight,
    # enqueue -> all streams idle, host clock.  Outside the timed region.
    latency = None
    if not sharded and not batched:
        lat = []
        for i in range(200):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            step(i)
            finish()
            lat.append((time.perf_counter() - t1) * 1e6)
        lat = np.sort(np.asarray(lat[20:]))
        latency = {"p50": float(lat[len(lat) // 2]), "p99": float(lat[min(len(lat) - 1, int(len(lat) * 0.99))]),
                   "n": int(len(lat)), "what": "one query in flight: enqueue + wait for its result, host clock"}

    # ------------------------------------------------------------------ sharded answer == single-index answer
    ok = True
    shard_check = None
    if sharded and args.check_queries > 0:
        nq = min(args.check_queries, args.queries)
        tickets = [step(i) for i in range(nq)]
        finish()
        got = [searcher.result(tk) for tk in tickets]
        barrier()
        if rank == 0:
            shard_check = check_against_single_index(args, synth, Index, nat, lib, device, local_rank, Q, T, n_terms,
                                                     got, nq, K, TOPN, idf, avgdl, allow, d_allow)
            ok = ok and shard_check["sharded_matches_single"]
        barrier()

    # ------------------------------------------------------------------ report (rank 0)
    if rank == 0:
        scan_avg_ms = scan_ms / max(scan_n, 1)
        # N*D*4 per query (this rank's rows), + N*2 with a filter
        per_query_bytes = n_local * args.dim * 4 + (n_local * 2 if args.filter else 0)
        queries_per_launch = (scan_units / scan_n) if (scan_n and not batched) else 1.0
        alg_bytes = per_query_bytes * queries_per_launch  # a K1 launch of the sharded path scans a GROUP of queries,
        achieved = alg_bytes / (scan_avg_ms * 1e-3) / 1e9 if scan_n else 0.0  # each its own pass over the shard
        traffic, traffic_source = carried_traffic(n_local, args.dim, queries_per_launch)
        per_step = args.batch if batched else 1
        line = {
            "metric": "queries/sec, hybrid (dense + BM25) RRF top-10 at batch=1, %s corpus" % (
                "%d x %d" % (args.rows, args.dim) if (weak or args.rows != 1_000_000 or args.dim != 768) else "1M x 768")
            if hybrid else ("queries/sec, dense top-%d, batch=%d (MFMA)" % (TOPN, args.batch) if batched else
                            "queries/sec, dense top-%d at batch=1" % TOPN),
            "value": args.steps * per_step / elapsed,
            "unit": "queries/s",
            "n_gpus": dist.get_world_size() if sharded else 1,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak" if weak else "strong",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": (("C5-shaped" if weak else "C3") + ": %d x %d hybrid (dense + BM25 CSR postings) RRF top-%d, batch=1"
                             % (args.rows, args.dim, TOPN))
                if hybrid else (("C4: %d x %d dense, batch=%d queries per pass (fp32 MFMA), top-%d"
                                 % (args.rows, args.dim, args.batch, TOPN)) if batched else
                                ("dense-only brute-force top-%d, %d x %d, batch=1" % (TOPN, args.rows, args.dim))),
                "rows": args.rows, "dim": args.dim, "rows_per_gpu": n_local,
                "postings_per_gpu": (int(post["post_doc"].numel()) if post else 0), "vocab": args.vocab if hybrid else 0,
                "similarity_k": K, "top_n": TOPN, "wrrf_k": WRRF_K, "weights": [W_DENSE, W_BM25],
                "sharding": ("rows/%d + RCCL all-gather of per-shard top-k, %d queries per all-gather"
                             % (world, args.exchange_group)) if sharded else "none",
                "bm25_arith": "f64", "source_filter": "CG,NG-shaped: 255 of 300 source ids allowed" if args.filter else None,
                "corpus": "block-seeded: a function of (seed, global row), identical for every --gpus",
            },
            "roofline": {
                "kernel": "dense_scan_kernel (K1)", "bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_source,
                "algorithmic_bytes_per_launch": alg_bytes, "avg_launch_ms": scan_avg_ms, "launches": scan_n,
                "queries_per_launch": queries_per_launch, "sampled": sampled,
            },
            "index_build_s": build_s,
        }
        if latency:
            line["latency_us"] = latency
        if shard_check:
            line.update(shard_check)
        if batched:
            line["roofline"] = batched_roofline(args.batch, n_local, args.dim, scan_avg_ms, scan_n,
                                                args.batch_precision == "bf16x3")
            line["dtype"] = "bf16x3" if args.batch_precision == "bf16x3" else "f32"
        if not sharded and hybrid and not args.no_also and not args.filter:
            line["also"] = also_measurements(args, torch, nat, lib, idx, Index, synth, E, Q, T, n_terms, post, device,
                                             local_rank, K, TOPN)
        if not sharded and not args.no_cpu_baseline and not batched:
            line["cpu_baseline"] = cpu_baseline(args, E, Q, post, idf, avgdl, term_lists, out, cnt, hybrid, K, TOPN,
                                                (W_DENSE, W_BM25, WRRF_K), allowed_rows)
            line["recall_at_10"] = line["cpu_baseline"].pop("niceqa_recall_at_10")
            ok = ok and line["cpu_baseline"]["gpu_results_match_cpu"] and bool(line["recall_at_10"].get("equal", True))
        print(json.dumps(line), flush=True)
    if sharded:
        dist.barrier()
        dist.destroy_process_group()
    if not ok:
        print("[bench] PARITY FAILURE (see gpu_results_match_cpu / sharded_matches_single in the line above)",
              file=sys.stderr, flush=True)
        return 1
    return 0


def carried_traffic(n_rows, dim, queries_per_launch):
    """HBM bytes per K1 launch from the PMC counters.  NOT measured in this run (rocprofv3 --pmc needs its own
    passes): carried over from the committed collection for this shape, and labelled as such."""
    for name in ("r03_pmc_dense_scan.json", "r02_pmc_dense_scan.json"):
        path = os.path.join(REPO, "profiles", name)
        try:
            with open(path) as f:
                rec = json.load(f)
        except (OSError, ValueError):
            continue
        if rec.get("rows") == n_rows and rec.get("dim") == dim and rec.get("hbm_bytes_per_launch") is not None:
            src = ("profiles/%s: builder-run rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, FETCH_SIZE x 2 "
                   "as the gfx950 guide prescribes) on one-query launches of %s at commit %s; not collected in this run"
                   % (name, rec.get("kernel", "dense_scan_kernel"), rec.get("commit", "?")))
            return rec["hbm_bytes_per_launch"] * queries_per_launch, src
    return None, None


def batched_roofline(batch, n_rows, dim, avg_ms, launches, split):
    flop = 2.0 * batch * n_rows * dim  # 2*Q*N*D per pass
    tf = flop / (avg_ms * 1e-3) / 1e12 if launches else 0.0
    peak = BF16_MFMA_PEAK_TF / 3.0 if split else F32_MFMA_PEAK_TF  # three bf16 MFMAs per product
    return {
        "kernel": ("dense_batched_split_kernel (sampled pass) + dense_batched_split_dma_kernel (full pass, LDS-DMA "
                   "images)" if split else "dense_batched_kernel") + " (K2: sample pass + threshold + full pass)",
        "bound": "mfma",
        "achieved": tf, "peak": peak, "unit": "TFLOP/s", "frac": tf / peak, "traffic": None,
        "algorithmic_flop_per_launch": flop, "avg_launch_ms": avg_ms, "launches": launches,
        "note": ("bf16 x 3 split products (hi.hi + hi.lo + lo.hi), f32 accumulate; peak = 2.5 PF / 3; scores "
                 "within ~1e-6 of f32 (bound 3e-5), not bit-equal" if split else
                 "f32-in/f32-acc v_mfma_f32_32x32x2_f32") + "; the timed span includes the sampled-threshold "
                "pre-pass, the flop count does not; both are power-limited on this board: the shader clock settles near "
                "1.65 GHz (bf16 x 3) / 2.2 GHz (f32) under the matrix load, DESIGN.md section 3"}


def check_against_single_index(args, synth, Index, nat, lib, device, local_rank, Q, T, n_terms, got, nq, K, TOPN,
                               idf, avgdl, allow, d_allow):
    """Rank 0, outside the timed region: ONE index over the whole corpus (same blocks, same global statistics) asked
    the same queries through anrag_hybrid_search_device; ids and fused fp64 scores must equal the sharded answers."""
    import torch

    t0 = time.time()
    rows = args.rows
    E1 = synth.dense_corpus(rows, args.dim, CORPUS_SEED, device)
    post1 = synth.bm25_postings(rows, args.vocab, BM25_SEED, device)
    src1 = ((np.arange(rows, dtype=np.int64)) % 300).astype(np.uint16) if args.filter else None
    torch.cuda.synchronize()
    one = Index(local_rank)
    one.dense_load((E1.data_ptr(), rows, args.dim), source_id=src1)
    one.bm25_load(post1["indptr"], (post1["post_doc"].data_ptr(), post1["post_doc"].numel()),
                  (post1["post_tf"].data_ptr(), post1["post_tf"].numel()), idf, post1["doc_len"], avgdl,
                  synth.BM25_K1, synth.BM25_B, source_id=src1)
    out = torch.zeros((nq, TOPN, 2), dtype=torch.int64, device=device)
    cnt = torch.zeros(nq, dtype=torch.int32, device=device)
    torch.cuda.synchronize()  # torch's zero fills run on its own stream: they must not land after the library's answers
    ap = d_allow.data_ptr() if d_allow is not None else None
    for qi in range(nq):
        nat.check(lib.anrag_hybrid_search_device(one.handle, Q[qi].data_ptr(), T[qi].data_ptr(), n_terms[qi], K,
                                                 W_DENSE, W_BM25, WRRF_K, TOPN, ap, ap, out[qi].data_ptr(),
                                                 cnt[qi:].data_ptr()))
    one.sync()
    same = True
    worst = 0.0
    for qi in range(nq):
        n = int(cnt[qi].item())
        rec = out[qi, :n].cpu().numpy()
        ids, scores = got[qi]
        if ids.tolist() != rec[:, 1].tolist() or not np.array_equal(scores, rec[:, 0].copy().view(np.float64)):
            same = False
            if len(scores) == n:
                worst = max(worst, float(np.max(np.abs(scores - rec[:, 0].copy().view(np.float64)))))
    one.close()
    del E1, post1
    return {"sharded_matches_single": bool(same), "sharded_check": {
        "queries": nq, "max_abs_fused_score_diff": worst, "seconds": time.time() - t0,
        "what": "rank 0 built one index over all %d rows from the same seeded blocks and global BM25 statistics; "
                "ids and fused fp64 scores compared exactly" % rows}}


# ====================================================================== side measurements of the same run
def also_measurements(args, torch, nat, lib, idx, Index, synth, E, Q, T, n_terms, post, device, local_rank, K, TOPN):
    """C2 (100k x 768 dense, batch=1), the 8-GPU shard shape (125k rows, 8 queries per launch), C4 (256 queries per
    MFMA pass over the 1M rows, f32 and bf16x3) and K3 / tail alone -- short passes after the headline's timed
    region, every launch of the measured kernel bracketed by HIP events on its stream.  Each entry carries the
    algorithmic work per launch and the average launch time: frac = work / time / peak."""
    res = {}
    n_rows, dim = E.shape

    def k1_pass(sub, rows, group, steps):
        qn = Q.shape[0]
        outb = torch.zeros((qn, TOPN, 2), dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        def run(n):
            for i in range(0, n, group):
                qi = i % qn
                g = min(group, qn - qi)
                nat.check(lib.anrag_dense_search_device(sub.handle, Q[qi].data_ptr(), g, TOPN, None, outb[qi].data_ptr()))
            sub.sync()
        run(64)
        sub.profile(False)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(steps)
        wall = time.perf_counter() - t0
        sub.profile(True, kernels=[nat.KERNEL_DENSE_SCAN], every=1)
        sub.profile_reset()
        run(256)
        ms, n = sub.profile_read(nat.KERNEL_DENSE_SCAN)
        units = sub.profile_units(nat.KERNEL_DENSE_SCAN)
        sub.profile(False)
        per_launch = units / max(n, 1)
        byts = rows * dim * 4 * per_launch
        gbs = byts / (ms / max(n, 1) * 1e-3) / 1e9
        r = {"kernel": "dense_scan_kernel (K1)", "bound": "hbm", "rows": rows, "dim": dim,
             "queries_per_launch": per_launch, "algorithmic_bytes_per_launch": byts,
             "avg_launch_us": ms / max(n, 1) * 1e3, "launches": n, "achieved": gbs, "peak": HBM_PEAK_GBS,
             "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "queries_per_s": steps / wall,
             "us_per_query": wall / steps * 1e6}
        if group == 1:
            # One query per call: consecutive queries run on up to four overlapping scan streams (api.hip: lanes), so
            # a launch's event-bracketed duration is no longer what a query costs (four kernels share the chip while
            # it runs, and the brackets themselves serialise the lanes): the figure here is THROUGHPUT-based --
            # bytes of one pass / wall-clock per query over the un-bracketed loop above, launch boundaries included.
            gbs_w = rows * dim * 4 / (wall / steps) / 1e9
            lat = []
            for i in range(200):  # one query alone: submit, wait for its answer
                t1 = time.perf_counter()
                nat.check(lib.anrag_dense_search_device(sub.handle, Q[i % qn].data_ptr(), 1, TOPN, None,
                                                        outb[i % qn].data_ptr()))
                sub.sync()
                lat.append(time.perf_counter() - t1)
            r.update({"avg_launch_us_event_bracketed": r["avg_launch_us"], "frac_event_bracketed": r["frac"],
                      "achieved": gbs_w, "frac": gbs_w / HBM_PEAK_GBS, "avg_launch_us": wall / steps * 1e6,
                      "timing": "wall-clock over %d back-to-back single-query calls (no grouping by the caller)" % steps,
                      "single_query_alone_p50_us": float(np.median(lat)) * 1e6,
                      "single_query_alone_p99_us": float(np.percentile(lat, 99)) * 1e6})
        return r

    if dim == 768 and n_rows >= 125_000:
        for name, rows, group in (("c2_100k_x768_batch1", 100_000, 1), ("c2_100k_x768_8_per_launch", 100_000, 8),
                                  ("shard_125k_x768_8_per_launch", 125_000, 8)):
            with Index(local_rank) as sub:
                sub.dense_load((E.data_ptr(), rows, dim))  # the first `rows` rows of the same corpus
                res[name] = k1_pass(sub, rows, group, 2048)
        res["c2_100k_x768_batch1"]["what"] = "BASELINE configs[1]: dense-only top-%d at batch=1, 100k x 768" % TOPN
        res["shard_125k_x768_8_per_launch"]["what"] = ("per-rank dense leg of the 8-GPU strong-scaling run: 125k rows, "
                                                        "each query its own pass, 8 queries per launch")
    # ---- C4: 256 queries per pass over all rows on the matrix cores
    if dim % 32 == 0 and n_rows >= 65536:
        Qb, _ = synth.dense_queries_global(args.rows, dim, 256, QUERY_SEED + 1, CORPUS_SEED, device)
        outb = torch.zeros((256, TOPN, 2), dtype=torch.int64, device=device)
        flag = torch.zeros(256, dtype=torch.int32, device=device)
        ref = torch.zeros((4, TOPN, 2), dtype=torch.int64, device=device)
        # torch fills these on ITS stream; the library writes them on its own streams: without this the zero fill of
        # `ref` could land after K1's answer (it did, once the passes above it changed the timing)
        torch.cuda.synchronize()
        nat.check(lib.anrag_dense_search_device(idx.handle, Qb.data_ptr(), 4, TOPN, None, ref.data_ptr()))
        idx.sync()
        for mode in ("f32", "bf16x3"):
            idx.set_batched_precision(mode)
            def run(n):
                for _ in range(n):
                    nat.check(lib.anrag_dense_search_batch_device(idx.handle, Qb.data_ptr(), 256, TOPN, None,
                                                                  outb.data_ptr(), flag.data_ptr()))
                idx.sync()
            # the matrix pipes pull the board to its power limit and the shader clock needs a few dozen passes to
            # settle (a 6-pass measurement read 15-20 % slower than the steady state): warm up, then time 60 passes
            run(30)
            idx.profile(True, kernels=[nat.KERNEL_DENSE_BATCHED], every=1)
            idx.profile_reset()
            run(60)
            ms, n = idx.profile_read(nat.KERNEL_DENSE_BATCHED)
            idx.profile(False)
            r = batched_roofline(256, n_rows, dim, ms / max(n, 1), n, mode == "bf16x3")
            r["queries_per_s"] = 256.0 / (ms / max(n, 1) * 1e-3)
            # K2's answers for 4 of the queries against K1's: rank by rank within the 1e-4 bar (ids may swap inside a
            # near-tie under the split-precision arithmetic, whose scores sit ~1e-6 from the f32 ones)
            got_s = outb[:4, :, 0].cpu().numpy().view(np.float64)
            ref_s = ref[:, :, 0].cpu().numpy().view(np.float64)
            r["vs_k1_on_4_queries"] = {"ids_equal": bool(torch.equal(outb[:4, :, 1], ref[:, :, 1])),
                                       "max_abs_score_diff": float(np.max(np.abs(got_s - ref_s))),
                                       "within_1e-4": bool(np.max(np.abs(got_s - ref_s)) <= 1e-4),
                                       "overflow_flags": int(flag[:4].abs().sum())}
            res["c4_256x%dx%d_%s" % (n_rows, dim, mode)] = r
        idx.set_batched_precision("f32")
    # ---- K3 and the tail alone (they hide under K1 in the hybrid pipeline)
    if post is not None:
        qn = Q.shape[0]
        outk = torch.zeros((qn, K, 2), dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        df = np.diff(post["indptr"])
        touched = [int(sum(int(df[t]) for t in T[i, : n_terms[i]].cpu().tolist() if t >= 0)) for i in range(qn)]
        def run(n):
            for i in range(n):
                qi = i % qn
                nat.check(lib.anrag_bm25_search_device(idx.handle, T[qi].data_ptr(), n_terms[qi], K, None,
                                                       outk[qi].data_ptr()))
            idx.sync()
        run(qn)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(4 * qn)
        wall = time.perf_counter() - t0
        idx.profile(True, kernels=[nat.KERNEL_BM25, nat.KERNEL_SELECT], every=1)
        idx.profile_reset()
        run(2 * qn)
        ms, n = idx.profile_read(nat.KERNEL_BM25)
        ms_t, n_t = idx.profile_read(nat.KERNEL_SELECT)
        idx.profile(False)
        byts = float(np.mean(touched)) * 12.0  # SURVEY.md 8(d) as built: 4 B doc id + 8 B fp64 impact per posting
        avg_us = ms / max(n, 1) * 1e3
        gbs = byts / (avg_us * 1e-6) / 1e9
        res["k3_bm25"] = {"kernel": "bm25_kernel (K3)", "bound": "hbm (latency-dominated)", "docs": int(len(post["doc_len"])),
                          "mean_terms": float(np.mean(n_terms)), "mean_postings_touched": float(np.mean(touched)),
                          "algorithmic_bytes_per_launch": byts, "avg_launch_us": avg_us, "launches": n,
                          "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS,
                          "queries_per_s_bm25_only": 4 * qn / wall,
                          "what": "sum over query terms of df(t) x 12 B, one launch per query, idle GPU"}
        res["tail"] = {"kernel": "query_tail_kernel (list merge -> top-k records)", "bound": "latency",
                       "avg_launch_us": ms_t / max(n_t, 1) * 1e3, "launches": n_t}
        # ---- K3 with 8 queries per launch (anrag_bm25_search_group_device): the launch and the tail are paid once per group
        import ctypes as C

        PT = (C.c_void_p * qn)(*[T[i].data_ptr() for i in range(qn)])
        PO = (C.c_void_p * qn)(*[outk[i].data_ptr() for i in range(qn)])
        PN = (C.c_int32 * qn)(*[int(x) for x in n_terms])
        for per in (8, 16):  # 8: the hybrid pipeline's exchange group; 16: what a BM25-only list call launches
            gq = qn // per * per
            def run_groups(n):
                for i in range(0, n, per):
                    o = i % gq
                    nat.check(lib.anrag_bm25_search_group_device(
                        idx.handle, C.cast(C.byref(PT, o * 8), C.c_void_p), C.cast(C.byref(PN, o * 4), C.c_void_p), per, K, None,
                        C.cast(C.byref(PO, o * 8), C.c_void_p)))
                idx.sync()
            if gq >= per:
                run_groups(gq)
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                run_groups(8 * gq)
                wall8 = time.perf_counter() - t0
                idx.profile(True, kernels=[nat.KERNEL_BM25], every=1)
                idx.profile_reset()
                run_groups(2 * gq)
                ms8, n8 = idx.profile_read(nat.KERNEL_BM25)
                u8 = idx.profile_units(nat.KERNEL_BM25)
                idx.profile(False)
                us_q = ms8 / max(u8, 1) * 1e3
                res["k3_bm25_%d_per_launch" % per] = {
                    "kernel": "bm25_kernel (K3), %d queries per launch (a workgroup set per query)" % per,
                    "bound": "hbm (latency-dominated)",
                    "algorithmic_bytes_per_query": byts, "avg_launch_us": ms8 / max(n8, 1) * 1e3, "launches": n8,
                    "us_per_query_in_kernel": us_q, "achieved": byts / (us_q * 1e-6) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": byts / (us_q * 1e-6) / 1e9 / HBM_PEAK_GBS, "queries_per_s_bm25_only": 8 * gq / wall8}
    # ---- full-ranking mode for query lists (retrieval_eval.py's similarity_k = common_sections_n = 12000)
    try:
        res.update(full_ranking_measurements(args, torch, idx, Index, synth, E, Q, T, n_terms, post, device, local_rank))
    except Exception as e:  # a side measurement must not take the headline down
        res["full_ranking_error"] = repr(e)
    return res


def full_ranking_measurements(args, torch, idx, Index, synth, E, Q, T, n_terms, post, device, local_rank):
    """`anrag_rank_batch` (rank_batch.hip) against the per-query entry points it replaces, wall-clock through the C ABI
    from host operands: (1) the reference's own corpus shape, 9,609 x 384 + BM25, 2,048 hybrid queries, k = 12,000 >= N
    (every list is a full ranking: sort, not select); (2) this run's corpus (k = 12,000 << N: radix select first)."""
    from anrag.index import rank_batch

    out = {}
    KF = 12000

    def measure(name, dense_idx, bm_idx, q_host, term_lists, n_dense, dim, n_docs, df, sample, KF=KF, top_n=None):
        nq = len(term_lists)
        TN = KF if top_n is None else top_n
        legs = [dict(index=dense_idx, weight=W_DENSE, queries=q_host), dict(index=bm_idx, weight=W_BM25, term_lists=term_lists)]
        space = max(n_dense, n_docs)
        rank_batch(legs, nq, KF, WRRF_K, TN, id_space=space)  # warm-up: scratch pool, LDS attributes
        t0 = time.perf_counter()
        ids, _, cnt = rank_batch(legs, nq, KF, WRRF_K, TN, id_space=space)
        t_ids = (time.perf_counter() - t0) / nq
        expect = ids[:, 0].copy()
        t0 = time.perf_counter()
        _, _, _, ranks = rank_batch(legs, nq, KF, WRRF_K, TN, id_space=space, expect=expect, want_ids=False)
        t_rank = (time.perf_counter() - t0) / nq
        kd, kb = min(KF, n_dense), min(KF, n_docs)
        def one(i):  # the per-query path: score dump + library sort of all N, three more sorts for the fusion
            dd, _, dc = dense_idx.dense_search(q_host[i], kd)
            bd, _, bc = bm_idx.bm25_search(term_lists[i], kb)
            return dense_idx.wrrf([dd[0, :int(dc[0])], bd[:bc]], [W_DENSE, W_BM25], WRRF_K, TN)[0]
        one(0)  # its scratch buffers are allocated on first use
        t0 = time.perf_counter()
        singles = [one(i) for i in range(sample)]
        t_single = (time.perf_counter() - t0) / sample
        same = all(np.array_equal(f, ids[i, :cnt[i]]) for i, f in enumerate(singles))
        touched = float(np.mean([sum(int(df[t]) for t in tl if t >= 0) for tl in term_lists]))
        tile_group = 16 if dim * 4 * 16 <= 48 * 1024 else max(2, (48 * 1024 // (dim * 4)) & ~1)  # queries K1T holds in LDS
        if dim in (512, 768) and n_dense >= 32 * 256:
            tile_group = 32  # the matrix-core form (dense_tile_mfma.hip)
        byts = (n_dense * dim * 4 / tile_group + 2 * n_dense * 4) + (touched * 12 + 2 * n_docs * 8) + 2 * (kd + kb) * 4
        out[name] = {
            "what": "dense + BM25 ranking and weighted RRF of %d queries in one anrag_rank_batch call, similarity_k = %d, "
                    "common_sections_n = %d (retrieval_eval.py:142-143 at 12,000); host operands in, host results out"
                    % (nq, KF, TN),
            "rows": n_dense, "dim": dim, "bm25_docs": n_docs, "queries": nq,
            "us_per_query_ids_out": t_ids * 1e6, "us_per_query_rank_of_expected_only": t_rank * 1e6,
            "us_per_query_one_by_one": t_single * 1e6, "one_by_one_sample": sample,
            "speedup_vs_one_by_one": t_single / t_ids, "speedup_rank_only": t_single / t_rank,
            "ids_equal_one_by_one": bool(same), "rank_of_expected_ok": bool(np.all(ranks == 1)),
            "algorithmic_bytes_per_query": byts,
            "bound": "K1T score tiles (matrix cores at 512 / 768-d on large corpora, else VALU: DESIGN.md K4) + LDS (sorts)",
            "achieved": byts / t_rank / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": byts / t_rank / 1e9 / HBM_PEAK_GBS,
            "bytes_are": "the rank-only mode's HBM bytes per query: corpus read once per %d queries (K1T) + score tile written "
                         "and read (dense N*D*4/%d + 2*N*4; BM25 sum df*12 + 2*N*8) + the two ranked lists written and read; "
                         "wall-clock includes the host side of the call" % (tile_group, tile_group)}

    # (1) the reference's corpus shape
    n1, d1 = 9609, 384
    E1 = synth.dense_corpus(n1, d1, CORPUS_SEED + 7, device)
    Q1, _ = synth.dense_queries(E1, 2048, QUERY_SEED + 7)
    post1 = synth.bm25_postings(n1, 50_000, BM25_SEED + 7, device)
    df1 = post1["df"].cpu().numpy()
    idf1 = synth.bm25_idf(df1, n1)
    t1 = [np.asarray(t, np.int32) for t in synth.bm25_queries(post1, 128, QUERY_SEED + 8)]
    t1 = [t1[i % len(t1)] for i in range(2048)]
    torch.cuda.synchronize()
    with Index(local_rank) as c1:
        c1.dense_load((E1.data_ptr(), n1, d1))
        c1.bm25_load(post1["indptr"], (post1["post_doc"].data_ptr(), post1["post_doc"].numel()),
                     (post1["post_tf"].data_ptr(), post1["post_tf"].numel()), idf1, post1["doc_len"],
                     float(post1["total_len"]) / n1, synth.BM25_K1, synth.BM25_B)
        measure("full_ranking_9609x384_k12000", c1, c1, Q1.cpu().numpy(), t1, n1, d1, n1, df1, 16)
        # the same corpus at ONE dense query per call (top-10): stream of calls, and one query alone
        from anrag import _native as nat

        lib = nat.load_library()
        o1 = torch.zeros((64, 10, 2), dtype=torch.int64, device=device)
        torch.cuda.synchronize()
        def run1(m):
            for i in range(m):
                nat.check(lib.anrag_dense_search_device(c1.handle, Q1[i % 64].data_ptr(), 1, 10, None, o1[i % 64].data_ptr()))
            c1.sync()
        run1(256)
        t0 = time.perf_counter()
        run1(4096)
        w1 = (time.perf_counter() - t0) / 4096
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            nat.check(lib.anrag_dense_search_device(c1.handle, Q1[i % 64].data_ptr(), 1, 10, None, o1[i % 64].data_ptr()))
            c1.sync()
            lat.append(time.perf_counter() - t0)
        out["c1_9609x384_batch1"] = {
            "kernel": "dense_scan_kernel (K1)", "bound": "launch / latency (a 14.8 MB pass is 2 us of HBM time)", "rows": n1,
            "dim": d1, "us_per_query": w1 * 1e6, "queries_per_s": 1.0 / w1,
            "single_query_alone_p50_us": float(np.median(lat)) * 1e6,
            "single_query_alone_p99_us": float(np.percentile(lat, 99)) * 1e6,
            "timing": "wall-clock over 4096 back-to-back single-query calls; one query alone = call + anrag_index_sync"}
        # ... and as HYBRID queries, one per call (what the reference's app asks of this corpus): scan, BM25 kernel and
        # tail back to back on one of four lane streams
        T1 = torch.full((64, 16), -1, dtype=torch.int32, device=device)
        for i in range(64):
            T1[i, : len(t1[i])] = torch.from_numpy(t1[i][:16]).to(device)
        n1t = [min(16, len(t1[i])) for i in range(64)]
        o2 = torch.zeros((64, 10, 2), dtype=torch.int64, device=device)
        c2 = torch.zeros(64, dtype=torch.int32, device=device)
        torch.cuda.synchronize()
        def runh(m):
            for i in range(m):
                j = i % 64
                nat.check(lib.anrag_hybrid_search_device(c1.handle, Q1[j].data_ptr(), T1[j].data_ptr(), n1t[j], 25, W_DENSE,
                                                         W_BM25, WRRF_K, 10, None, None, o2[j].data_ptr(), c2[j:].data_ptr()))
            c1.sync()
        runh(256)
        t0 = time.perf_counter()
        runh(4096)
        wh = (time.perf_counter() - t0) / 4096
        lat = []
        for i in range(200):
            t0 = time.perf_counter()
            runh(1)
            lat.append(time.perf_counter() - t0)
        out["c1_9609x384_hybrid_batch1"] = {
            "kernel": "dense_scan_kernel + bm25_kernel + query_tail_kernel on one lane stream per query",
            "bound": "launch / latency", "rows": n1, "dim": d1, "us_per_query": wh * 1e6, "queries_per_s": 1.0 / wh,
            "single_query_alone_p50_us": float(np.median(lat)) * 1e6,
            "single_query_alone_p99_us": float(np.percentile(lat, 99)) * 1e6,
            "timing": "wall-clock over 4096 back-to-back hybrid calls (similarity_k 25, top 10), operands in HBM"}
    del E1, post1
    # (2) this run's corpus
    if post is not None and E.shape[0] >= 100_000:
        nb = min(64, Q.shape[0])
        nq = 256  # the run's queries four times over: a list long enough for two full score-tile launches (128 each)
        qh = np.ascontiguousarray(np.tile(Q[:nb].cpu().numpy(), ((nq + nb - 1) // nb, 1))[:nq])
        tl = [T[i % nb, : n_terms[i % nb]].cpu().numpy().astype(np.int32) for i in range(nq)]
        df = np.diff(post["indptr"])
        measure("full_ranking_%dx%d_k12000" % (E.shape[0], E.shape[1]), idx, idx, qh, tl, int(E.shape[0]), int(E.shape[1]),
                int(len(post["doc_len"])), df, 4)
        # the headline's query shape (similarity_k 25, top 10) asked as a LIST: score tiles read the corpus once per 16
        # queries; "one by one" here is the per-query entry points, the device pipeline's list form is the headline's rate
        measure("hybrid_list_%dx%d_k25" % (E.shape[0], E.shape[1]), idx, idx, qh, tl, int(E.shape[0]), int(E.shape[1]),
                int(len(post["doc_len"])), df, 8, KF=25, top_n=10)
    return out


# ====================================================================== CPU port beside it
def cpu_baseline(args, E, Q, post, idf, avgdl, term_lists, gpu_out, gpu_cnt, hybrid, K, TOPN, fusion, allowed=None):
    """The reference-shaped CPU path (oracle = port of src/search_engine.py) timed on this box's host
    cores over a bounded sample of the same queries, and used as the parity check of the GPU results."""
    from oracle import ref_bm25, ref_search

    w_dense, w_bm25, wrrf_k = fusion
    nq = max(1, min(args.cpu_queries, args.queries))
    e_host = E.cpu().numpy()
    q_host = Q[:nq].cpu().numpy()
    if hybrid:
        post_doc = post["post_doc"].cpu().numpy()
        post_tf = post["post_tf"].cpu().numpy()
    rows = list(e_host)  # the reference keeps one ndarray per DataFrame row (database_manager.py:49)
    t_ref = t_pre = 0.0
    parity_ok = True
    exact_lists = 0
    max_dscore = 0.0
    for qi in range(nq):
        t0 = time.perf_counter()
        emb = np.stack(rows)  # search_engine.py:80 -- re-materialised on every query
        sims = np.dot(q_host[qi].reshape(1, -1), emb.T).flatten()
        if allowed is not None:  # the reference filters the DataFrame first (pandas str ops, not timed here)
            sims = np.where(allowed, sims, -np.inf)
        top = ref_search.numpy_topk_idiom(sims, K)
        dense_list = top.tolist()
        t_stack_path = time.perf_counter() - t0
        t1 = time.perf_counter()
        sims2 = np.dot(q_host[qi].reshape(1, -1), e_host.T).flatten()
        ref_search.numpy_topk_idiom(sims2, K)
        t_dense_pre = time.perf_counter() - t1
        t2 = time.perf_counter()
        if hybrid:
            scores = ref_bm25.csr_get_scores(post["indptr"], post_doc, post_tf, idf, post["doc_len"], avgdl,
                                             1.7, 0.83, term_lists[qi].tolist())
            bm_list = ref_search.canonical_topk(scores, K, allowed).tolist()
            fused = ref_search.weighted_reciprocal_rank_fusion(
                [(dense_list, "dense"), (bm_list, "BM25")], {"dense": w_dense, "BM25": w_bm25}, int(wrrf_k))[:TOPN]
            want_ids = [i for i, _ in fused]
        else:
            want_ids = dense_list[:TOPN]
        t_rest = time.perf_counter() - t2
        t_ref += t_stack_path + t_rest
        t_pre += t_dense_pre + t_rest
        # parity of the GPU's answer for this query (full-size check, same inputs): the id LIST must be equal; where
        # it is not, every differing position must be a dense near-tie (the two rows' dot products within the 1e-4
        # the north star allows) and, for hybrid, the fused scores must still agree position by position
        n = int(gpu_cnt[qi].item()) if hybrid else TOPN
        rec = gpu_out[qi, :n].cpu().numpy()
        got_ids = rec[:, 1].tolist()
        if got_ids == want_ids:
            exact_lists += 1
        elif len(got_ids) != len(want_ids):
            parity_ok = False
        else:
            for g, w in zip(got_ids, want_ids):
                if g != w and not abs(float(sims[g]) - float(sims[w])) <= 1e-4:
                    parity_ok = False
        if hybrid:
            got_s = rec[:, 0].copy().view(np.float64)
            want_s = np.array([s for _, s in fused])
            d = float(np.max(np.abs(got_s - want_s))) if n == len(want_s) and n else (0.0 if n == len(want_s) else 1.0)
            max_dscore = max(max_dscore, d)
            if d > 1e-12:
                parity_ok = False
    try:
        from threadpoolctl import threadpool_info

        blas_threads = max([p.get("num_threads", 1) for p in threadpool_info()] or [1])
    except Exception:
        blas_threads = os.cpu_count() or 1
    recall = niceqa_recall()
    return {
        "niceqa_recall_at_10": recall,
        "value": nq / t_ref, "unit": "queries/s", "cores": blas_threads, "kind": "port",
        "sample": "%d of the %d benchmark queries, full %d x %d corpus on the host: reference-shaped path "
                  "(np.stack of per-row arrays per query as search_engine.py:80 does, np.dot, argpartition; CSR "
                  "restatement of BM25Okapi.get_scores; Python WRRF)" % (nq, args.queries, args.rows, args.dim),
        "value_prestacked": nq / t_pre,
        "note_prestacked": "same sample with the corpus matrix stacked once up front (removes the reference's "
                           "per-query np.stack)",
        "gpu_results_match_cpu": bool(parity_ok), "identical_id_lists": exact_lists,
        "max_abs_fused_score_diff": max_dscore,
    }


def niceqa_recall():
    """Recall@10 on data/NICEQA.csv over the stand-in corpus (anrag/niceqa.py): GPU path vs CPU reference path.
    The acceptance criterion is equality; the absolute value says little (hashed-BoW stand-in embeddings)."""
    from anrag import niceqa
    from oracle.niceqa_ref import cpu_ranked_ids

    gold = os.path.join(REPO, "tests", "golden")
    try:
        data = niceqa.load_standin(os.path.join(gold, "suggested_queries_bm25_preprocessed.json.gz"),
                                   os.path.join(gold, "NICEQA.csv"))
    except OSError as e:
        return {"error": str(e)}
    gpu = niceqa.gpu_ranked_ids(data)
    cpu = cpu_ranked_ids(data, *niceqa.encode_questions(data))
    rg, rc = niceqa.recall_at_10(data, gpu), niceqa.recall_at_10(data, cpu)
    return {"gpu": rg["recall_at_10"], "cpu_reference": rc["recall_at_10"], "equal": rg == rc,
            "questions": rg["questions"], "with_gold_chunk": rg["with_gold_chunk"],
            "identical_top10_lists": sum(a == b for a, b in zip(gpu, cpu)),
            # the rest: near-ties of the fp32 dense scores (host BLAS and the scan sum in different orders; the dense
            # bar is 1e-4) move a candidate by a rank, and with it its RRF term
            "differing_lists_with_the_same_10_ids": sum(a != b and set(a) == set(b) for a, b in zip(gpu, cpu)),
            "ids_in_one_list_only_max": max([len(set(a) ^ set(b)) // 2 for a, b in zip(gpu, cpu)] or [0]),
            "corpus": "stand-in: 9,609 shipped chunk ids x 384-d hashed-BoW embeddings (no encoder weights offline)"}


if __name__ == "__main__":
    sys.exit(main())
