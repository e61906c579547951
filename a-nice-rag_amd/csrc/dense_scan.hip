// K1 -- brute-force dense scan + fused top-k, batch = 1.
//
// Replaces src/search_engine.py:80-87 of the reference (np.stack -> np.dot -> argpartition):
// the corpus matrix already sits in HBM row-major fp32, one query streams over it once.
//
// HBM-bound by construction: N*D*4 algorithmic bytes per query, 2 flop per 4 bytes.
// Mapping (wave64):
//   * G lanes share one row (G = 64 when D % 256 == 0: a whole wave reads 1 KiB of one row per
//     global_load_dwordx4; D = 384 uses G = 32, i.e. two rows per wave-load, 512 B each);
//   * a wave works in batches of R row-groups (R*CH ~ 6 dwordx4 loads per lane), software-pipelined over
//     two register sets so that the next batch is in flight while this one is reduced; ONE 4-wave
//     workgroup per CU strides the matrix: 24-48 KiB in flight per CU measured best on MI355X
//     (7.05 TB/s at 1M x 768; 8/16 waves or deeper unrolls = 96-192 KiB in flight lose 3-6 % and starve
//     co-running kernels: profiles/r01_scan_config_sweep.txt);
//   * the query slice of each lane lives in registers (CH float4), staged once per wave;
//   * G-lane butterfly reduction, then the wave's register top-k (wave_topk.hpp);
//   * one sorted list per WAVE -> tail kernel (tail.hip).  No workgroup merge here: a tree merge through LDS kept
//     every CU away from HBM for 3-5 us per query (bitonic networks + barriers) -- 1 % of a 1M-row scan, 7 % of a
//     125k-row one -- and the tail kernel merges off the scans' stream anyway; without it the waves of a workgroup
//     never meet at a barrier and go on to the next query of a launch on their own.
//   * the 6-step lane reduction is DPP row ops (no LDS traffic in the streaming loop).
#include "common.hpp"
#include "wave_topk.hpp"
#include "dense_scan_common.hpp"

namespace anrag {

// G  lanes per row (64,32,16); dim % (4*G) == 0
// CH float4 chunks per lane per row (dim / (4*G)), compile-time so the query sits in registers
// R  row-groups in flight per wave iteration
// FILTER  rows carry a source id and an allow bitmap is applied before selection
// SCORES  write every row's score (large-k path / anrag_dense_scores) instead of selecting
// The queries of one launch: each is scanned on its own (one pass over the matrix per query, batch = 1 arithmetic
// and traffic); a workgroup goes on to the next query as soon as it has finished its share of this one -- no
// grid-wide step between queries, so launch gap, ramp and the spread of the workgroups' finishing times are paid
// once per launch instead of once per query.
struct ScanQueries {
    const float *q[kScanGroupMax];
    float *blk_s[kScanGroupMax];
    uint32_t *blk_r[kScanGroupMax];
    int32_t n;
};

// The list merge of the PREVIOUS single query of the same stream, done by one extra workgroup of this launch (the
// lists are complete: the launch that wrote them has finished).  A query submitted alone used to cost a marker on the
// scans' stream plus a merge kernel on another stream waiting for it: hipEventRecord alone is 4-5 us of stream time per
// query on this stack (scripts/exp/launch_gap.hip), a tenth of a 100k-row pass.  out == nullptr: nothing to merge.
struct DeferredTail {
    const float *blk_s;
    const uint32_t *blk_r;
    int32_t n_lists, k;
    const int64_t *doc;
    int64_t base;
    anrag_candidate *out;
};

// 4 waves: the dense half of query_tail_kernel (tail.hip) -- same merge network, same records
__device__ __forceinline__ void dense_tail_block(const DeferredTail &dt) {
    __shared__ float lf_s[4 * kListLen];
    __shared__ uint32_t lf_r[4 * kListLen];
    const int lane = lane_id(), wave = threadIdx.x / kWave;
    WaveTopK<float> tf;
    tf.init(dt.k);
    merge_lists(tf, dt.blk_s, dt.blk_r, dt.n_lists, dt.k, wave, 4);
    for (int half = 2; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            lf_s[wave * kListLen + lane] = tf.s;
            lf_r[wave * kListLen + lane] = tf.r;
        }
        __syncthreads();
        if (wave < half) tf.merge_sorted(lf_s + (wave + half) * kListLen, lf_r + (wave + half) * kListLen);
        __syncthreads();
    }
    if (wave == 0 && lane < dt.k) {
        const bool empty = tf.r == kNoRow;
        anrag_candidate c;
        c.doc = empty ? -1 : (dt.doc ? dt.doc[tf.r] : dt.base + (int64_t)tf.r);
        c.score = empty ? -__builtin_huge_val() : (double)tf.s;
        dt.out[lane] = c;
    }
}

template <int G, int CH, int R, bool FILTER, bool SCORES, int THREADS = kScanThreads>
__global__ __launch_bounds__(THREADS) void dense_scan_kernel(
    const float *__restrict__ emb, ScanQueries Q, int64_t n_rows, int32_t dim, int32_t k,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits, float *__restrict__ scores_out,
    int64_t scores_stride, DeferredTail dt) {
    constexpr int GROUPS = kWave / G;  // rows per wave-load
    int scan_blocks = gridDim.x;
    if constexpr (!SCORES && THREADS == 4 * kWave) {
        if (dt.out != nullptr) {  // the last workgroup of the launch merges the previous query's lists
            scan_blocks -= 1;
            if ((int)blockIdx.x == scan_blocks) {
                dense_tail_block(dt);
                return;
            }
        }
    }
    constexpr int RW = GROUPS * R;     // rows per wave iteration
    constexpr int WAVES = THREADS / kWave;
    __shared__ uint32_t lds_allow[FILTER ? 2048 : 1];  // 65536 source ids

    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const int sub = lane % G, grp = lane / G;
    const bool leader = sub == G - 1;
    if constexpr (FILTER) {
        for (int i = threadIdx.x; i < 2048; i += THREADS) lds_allow[i] = allow_bits[i];
        __syncthreads();
    }

    const f32x4 *__restrict__ ev = reinterpret_cast<const f32x4 *>(emb);
    const int64_t row_f4 = dim / 4;
    // The queries of the launch are ONE stream of batches for a wave: the first batch of query i+1 (the same rows as the
    // first batch of query i) is issued before the last batch of query i is reduced, so the wave never drains its
    // loads at a query boundary (a drained boundary cost one exposed memory round trip, ~2 us, per query: 3.5 % of a
    // 125k-row pass).
    // this query's slice and, when registers allow (CH <= 3: a scan wave must stay within 128 VGPRs to share its SIMD
    // with four K3 waves of 96), the next one's, fetched a whole query ahead
    constexpr bool PREFETCH_Q = CH <= 3;
    f32x4 q[CH], qn[PREFETCH_Q ? CH : 1];
    WaveTopK<float> top;
    // Long rows (D >= 3072) are streamed in two halves: a batch is HALF a row-group (6-8 loads per lane, the same
    // 12-16 in flight as the other shapes), set b0 always the first half, b1 the second, the partial dot product
    // carried between them.  Whole-row batches of 12-16 loads needed 254+ VGPRs (two sets + the query slice); halves
    // need ~130, and 24-32 loads in flight per lane were past the optimum anyway (82.6 % at 3072-d).
    constexpr int SPLIT = CH >= 12 ? 2 : 1;
    constexpr int CHB = CH / SPLIT;
    static_assert(CH % SPLIT == 0 && (SPLIT == 1 || R == 1), "long rows: one row-group per batch, two halves");
    struct Batch {
        f32x4 v[R][CHB];
        uint32_t sid[R];
    };
    auto fetch_query = [&](int32_t qi) {  // -> qn; clamped behind the last query (nobody uses that copy)
        if constexpr (PREFETCH_Q) {
            const float *__restrict__ query = Q.q[qi < Q.n ? qi : Q.n - 1];
#pragma unroll
            for (int c = 0; c < CH; ++c) qn[c] = reinterpret_cast<const f32x4 *>(query)[c * G + sub];
        }
    };
    auto next_query = [&](int32_t qi) {  // qn -> q, and the fetch of the query after it: no wait at a boundary
        if constexpr (PREFETCH_Q) {
#pragma unroll
            for (int c = 0; c < CH; ++c) q[c] = qn[c];
            fetch_query(qi + 1);
        } else {
            const float *__restrict__ query = Q.q[qi];
#pragma unroll
            for (int c = 0; c < CH; ++c) q[c] = reinterpret_cast<const f32x4 *>(query)[c * G + sub];
        }
        top.init(SCORES ? 1 : k);
    };
    // row range of this wave.  List-keeping form: rows interleaved over all waves of the grid, `step` apart.  SCORES: a
    // contiguous block per wave (see finish_row).
    constexpr int RW_ = (kWave / G) * R;
    const int64_t total_waves = (int64_t)scan_blocks * (THREADS / kWave);
    const int64_t wave_global = (int64_t)blockIdx.x * (THREADS / kWave) + threadIdx.x / kWave;
    const int64_t rows_per_wave = ((n_rows + total_waves - 1) / total_waves + RW_ - 1) / RW_ * RW_;
    const int64_t scan_base0 = SCORES ? wave_global * rows_per_wave : wave_global * RW_;
    const int64_t scan_lim = SCORES ? (scan_base0 + rows_per_wave < n_rows ? scan_base0 + rows_per_wave : n_rows) : n_rows;
    float sc_buf = 0.f;
    int64_t sc_base = scan_base0;
    auto flush = [&](int32_t qi) {  // the wave's sorted list of query qi (SCORES: the scores still parked)
        if constexpr (!SCORES) {
            Q.blk_s[qi][(blockIdx.x * WAVES + wave) * kListLen + lane] = top.s;
            Q.blk_r[qi][(blockIdx.x * WAVES + wave) * kListLen + lane] = top.r;
        } else {
            if (sc_base + lane < scan_lim) scores_out[(int64_t)qi * scores_stride + sc_base + lane] = sc_buf;
            sc_base = scan_base0;
        }
    };
    // One batch = R row-groups of this wave: R*CH dwordx4 loads per lane.  The loop is software-pipelined over
    // two register sets: the loads of batch i+1 are issued BEFORE batch i is reduced, so the wave always has
    // loads in flight (a pure-read kernel of this geometry reaches 7.0-7.4 TB/s on MI355X:
    // profiles/r01_hbm_read_ceiling.txt).
    auto issue = [&](int64_t base, Batch &bt, int half = 0) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = base + r * GROUPS + grp;
            const int64_t rc = row < n_rows ? row : n_rows - 1;  // clamp: tail lanes re-read the last row
            if constexpr (FILTER) bt.sid[r] = src[rc];           // issued ahead of the row data
            const f32x4 *p = ev + rc * row_f4 + sub + half * CHB * G;
#pragma unroll
            for (int c = 0; c < CHB; ++c) bt.v[r][c] = __builtin_nontemporal_load(p + c * G);
        }
    };
    float part[R];        // SPLIT == 2: the first half's partial dot products
    uint32_t part_sid[R];
    int32_t qi = 0;  // the query whose batch is being reduced (SCORES: query qi's scores start at qi * scores_stride)
    // SCORES: a wave owns a CONTIGUOUS block of rows [base0, lim) (the list-keeping form interleaves the waves' rows) and
    // parks its scores in one register, lane j <- row sc_base + j, stored 64 at a time as one 256-byte piece.  One 4- or
    // 8-byte store per wave and step (the interleaved form's) put a store between every two batches of loads -- stores
    // count in vmcnt like loads, so the counted waits waited for them too: 479 us against 430 us per 1M-row pass.
    auto finish_row = [&](int64_t row, float acc, uint32_t sid) {
        acc = nan_first(group_sum<G>(acc));
        bool ok = row < n_rows;
        if constexpr (FILTER) ok = ok && source_ok(lds_allow, sid);
        if constexpr (SCORES) {
            const float val = ok ? acc : neg_inf<float>();
            const int64_t row0 = row - grp;  // the row of lane group 0: wave-uniform
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) {
                const float sg = read_lane(val, g * G + G - 1);
                const int64_t rg = row0 + g;
                if (rg < scan_lim) {
                    const int slot = (int)(rg - sc_base);
                    sc_buf = lane == slot ? sg : sc_buf;
                    if (slot == kWave - 1) {
                        scores_out[(int64_t)qi * scores_stride + sc_base + lane] = sc_buf;
                        sc_base += kWave;
                    }
                }
            }
        } else {
            const uint32_t r32 = (uint32_t)row;
            top.offer_lanes(leader && ok && top.admits(acc, r32), acc, r32);
        }
    };
    auto reduce = [&](int64_t base, const Batch &bt) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CHB; ++c) acc = dot4(bt.v[r][c], q[c], acc);
            finish_row(base + r * GROUPS + grp, acc, bt.sid[r]);
        }
    };
    auto reduce_first_half = [&](const Batch &bt) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CHB; ++c) acc = dot4(bt.v[r][c], q[c], acc);
            part[r] = acc;
            part_sid[r] = bt.sid[r];
        }
    };
    auto reduce_second_half = [&](int64_t base, const Batch &bt) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            float acc = part[r];
#pragma unroll
            for (int c = 0; c < CHB; ++c) acc = dot4(bt.v[r][c], q[CHB + c], acc);
            finish_row(base + r * GROUPS + grp, acc, part_sid[r]);
        }
    };
    const int64_t step = SCORES ? RW : (int64_t)scan_blocks * WAVES * RW;
    const int64_t base0 = scan_base0;
    if (base0 >= n_rows) {  // a wave without rows (tiny corpora): empty lists
        top.init(1);
        for (int32_t qe = 0; qe < Q.n; ++qe) flush(qe);
        return;
    }
    fetch_query(0);
    next_query(0);
    Batch b0, b1;
    issue(base0, b0);
    int64_t base = base0;
    // Every step issues the NEXT batch unconditionally (behind the launch's last batch it re-reads the wave's first
    // rows: 6 loads per wave and launch that nobody uses) and only then reduces the current one.  Unconditionally,
    // because a load behind a branch leaves the compiler unsure how many loads are younger than the ones it is about
    // to use, and it then waits for (nearly) all of them: `if (next < n_rows) issue(...)` made every step wait for the
    // batch it had just issued (s_waitcnt vmcnt(2..0) instead of vmcnt(11..6)).  The rare block at a query boundary
    // (two stores, a query reload) sits behind a scalar branch after the reduce.
    if constexpr (SPLIT == 2) {
        for (;;) {
            issue(base, b1, 1);  // the second half of the same rows
            reduce_first_half(b0);
            int32_t nq = qi;
            int64_t next = base + step;
            if (next >= scan_lim) {
                next = base0;
                ++nq;
            }
            issue(next, b0, 0);  // the first half of the next rows
            reduce_second_half(base, b1);
            if (nq != qi) {
                flush(qi);
                if (nq >= Q.n) break;
                qi = nq;
                next_query(qi);
            }
            base = next;
        }
        return;
    }
    for (;;) {
        int32_t nq = qi;
        int64_t next = base + step;
        if (next >= scan_lim) {
            next = base0;
            ++nq;
        }
        issue(next, b1);
        reduce(base, b0);
        if (nq != qi) {
            flush(qi);
            if (nq >= Q.n) break;
            qi = nq;
            next_query(qi);
        }
        base = next;
        next = base + step;
        if (next >= scan_lim) {
            next = base0;
            ++nq;
        }
        issue(next, b0);
        reduce(base, b1);
        if (nq != qi) {
            flush(qi);
            if (nq >= Q.n) break;
            qi = nq;
            next_query(qi);
        }
        base = next;
    }
}

// Any dim: one wave per row, scalar strided loads, query from global (L2).  Correctness path for
// odd dimensions (the golden vectors use D = 8); not a performance path.
__global__ __launch_bounds__(kScanThreads) void dense_scan_topk_generic_kernel(
    const float *__restrict__ emb, const float *__restrict__ query, int64_t n_rows, int32_t dim, int32_t k,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits, float *__restrict__ blk_score,
    uint32_t *__restrict__ blk_row, float *__restrict__ scores_out) {
    __shared__ uint32_t lds_allow[2048];
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const bool filtered = allow_bits != nullptr;
    if (filtered) {
        for (int i = threadIdx.x; i < 2048; i += kScanThreads) lds_allow[i] = allow_bits[i];
        __syncthreads();
    }
    WaveTopK<float> top;
    top.init(k > 0 ? k : 1);
    const int64_t total_waves = (int64_t)gridDim.x * kScanWaves;
    for (int64_t row = (int64_t)blockIdx.x * kScanWaves + wave; row < n_rows; row += total_waves) {
        const float *p = emb + row * dim;
        float acc = 0.f;
        for (int c = lane; c < dim; c += kWave) acc = __builtin_fmaf(p[c], query[c], acc);
        acc = nan_first(group_sum<kWave>(acc));  // total lands in lane 63
        bool ok = true;
        if (filtered) ok = source_ok(lds_allow, src[row]);
        if (scores_out != nullptr && lane == kWave - 1) scores_out[row] = ok ? acc : neg_inf<float>();
        if (k > 0) top.offer_lanes(lane == kWave - 1 && ok && top.admits(acc, (uint32_t)row), acc, (uint32_t)row);
    }
    if (k > 0) {
        blk_score[(blockIdx.x * kScanWaves + wave) * kListLen + lane] = top.s;
        blk_row[(blockIdx.x * kScanWaves + wave) * kListLen + lane] = top.r;
    }
}

// fp64 query (the reference's text path: the embedding API returns float64, and np.dot then promotes the fp32 matrix and
// scores in fp64, src/search_engine.py:157, :129): every row's dot product accumulated in fp64 from the fp32 rows
// (exact in fp64) and the fp64 query -- all N scores; selection is the score-array sort (sort_select.hip).  One wave
// per row, 16-byte loads when the rows allow it.  HBM-bound like K1 but not tuned like it: this path answers one
// query behind a remote embedding call.
__global__ __launch_bounds__(kScanThreads) void dense_scores_f64_kernel(
    const float *__restrict__ emb, const double *__restrict__ query, int64_t n_rows, int32_t dim,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits, double *__restrict__ scores_out) {
    __shared__ uint32_t lds_allow[2048];
    const int lane = lane_id();
    const int wave = threadIdx.x / kWave;
    const bool filtered = allow_bits != nullptr;
    if (filtered) {
        for (int i = threadIdx.x; i < 2048; i += kScanThreads) lds_allow[i] = allow_bits[i];
        __syncthreads();
    }
    const int64_t total_waves = (int64_t)gridDim.x * kScanWaves;
    const bool vec = dim % 4 == 0;
    for (int64_t row = (int64_t)blockIdx.x * kScanWaves + wave; row < n_rows; row += total_waves) {
        const float *p = emb + row * dim;
        double acc = 0.0;
        if (vec) {
            for (int c = lane * 4; c < dim; c += kWave * 4) {
                const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4 *>(p + c));
                acc = __builtin_fma((double)v.x, query[c], acc);
                acc = __builtin_fma((double)v.y, query[c + 1], acc);
                acc = __builtin_fma((double)v.z, query[c + 2], acc);
                acc = __builtin_fma((double)v.w, query[c + 3], acc);
            }
        } else {
            for (int c = lane; c < dim; c += kWave) acc = __builtin_fma((double)p[c], query[c], acc);
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) acc += __shfl_xor(acc, off, kWave);
        acc = nan_first(acc);
        bool ok = true;
        if (filtered) ok = source_ok(lds_allow, src[row]);
        if (lane == 0) scores_out[row] = ok ? acc : neg_inf<double>();
    }
}

int launch_dense_scores_f64(anrag_index *idx, hipStream_t st, const double *d_query, const uint32_t *d_allow_bits,
                            double *d_scores_out) {
    const uint32_t *allow = (idx->d_dense_src != nullptr) ? d_allow_bits : nullptr;
    const int64_t need = (idx->n_rows + kScanWaves - 1) / kScanWaves;
    const int64_t cap = (int64_t)idx->n_cus * 8;  // 32 waves per CU: latency hiding by occupancy, no software pipeline
    LaunchTimer t(idx, ANRAG_KERNEL_DENSE_SCAN, st);
    dense_scores_f64_kernel<<<(unsigned)(need < cap ? need : cap), kScanThreads, 0, st>>>(
        idx->d_emb, d_query, idx->n_rows, idx->dim, idx->d_dense_src, allow, d_scores_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

int dense_scan_grid(const anrag_index *idx) {
    int grid = idx->n_cus < kMaxScanBlocks ? idx->n_cus : kMaxScanBlocks;
    // small corpora: no more workgroups than there are wave-iterations of work
    const int64_t need = (idx->n_rows + kScanWaves * 4 - 1) / (kScanWaves * 4);
    if (need < grid) grid = (int)(need > 0 ? need : 1);
    return grid;
}

// VGPRs of the scan kernel this index's queries run (the filtered top-k variant: the larger one).  K3 sizes its
// workgroups so that they fit NEXT TO a scan wave on every SIMD (bm25.hip).
int dense_scan_vgprs(const anrag_index *idx) {
    int regs = 128;
    const bool known = scan_dispatch(idx->dim, [&](auto shape) {
        using S = decltype(shape);
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(
                                            &dense_scan_kernel<S::kG, S::kCH, S::kR, true, false>)) == hipSuccess)
            regs = attr.numRegs;
    });
    if (!known) {
        hipFuncAttributes attr;
        if (hipFuncGetAttributes(&attr, reinterpret_cast<const void *>(&dense_scan_topk_generic_kernel)) == hipSuccess)
            regs = attr.numRegs;
    }
    return regs;
}

int launch_dense_scan(anrag_index *idx, hipStream_t st, const float *d_query, int32_t k,
                      const uint32_t *d_allow_bits, float *d_scores_out, int set) {
    return launch_dense_scan_group(idx, st, &d_query, 1, k, d_allow_bits, d_scores_out, &set, 0);
}

// n <= kScanGroupMax queries in ONE launch, query i into block-list set sets[i] -- or, with d_scores_out, every score
// of query i to d_scores_out + i * scores_stride (the full-ranking tile, rank_batch.hip)
bool dense_scan_has_shape(const anrag_index *idx) {
    return scan_dispatch(idx->dim, [](auto) {});
}

int launch_dense_scan_group(anrag_index *idx, hipStream_t st, const float *const *d_queries, int32_t n_queries, int32_t k,
                            const uint32_t *d_allow_bits, float *d_scores_out, const int *sets, int64_t scores_stride,
                            const PendingTail *pending) {
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= kScanGroupMax, "scan group of %d queries", n_queries);
    ANRAG_REQUIRE(!pending || (!d_scores_out && dense_scan_has_shape(idx)), "deferred list merge: shaped scan kernels only");
    DeferredTail dt{};
    if (pending) {
        dt.blk_s = idx->d_blk_score_f32 + (int64_t)pending->set * kMaxScanLists * kListLen;
        dt.blk_r = idx->d_blk_row_a + (int64_t)pending->set * kMaxScanLists * kListLen;
        dt.n_lists = dense_scan_lists(idx);
        dt.k = pending->k;
        dt.doc = idx->d_dense_doc;
        dt.base = idx->dense_doc_base;
        dt.out = pending->out;
    }
    ANRAG_REQUIRE(!d_scores_out || n_queries == 1 || scores_stride >= idx->n_rows, "score tile rows overlap");
    const int64_t n = idx->n_rows;
    ScanQueries Q;
    Q.n = n_queries;
    for (int i = 0; i < kScanGroupMax; ++i) {
        const int j = i < n_queries ? i : 0;
        Q.q[i] = d_queries[j];
        Q.blk_s[i] = idx->d_blk_score_f32 + (int64_t)sets[j] * kMaxScanLists * kListLen;
        Q.blk_r[i] = idx->d_blk_row_a + (int64_t)sets[j] * kMaxScanLists * kListLen;
    }
    const float *d_query = d_queries[0];
    float *blk_s = Q.blk_s[0];
    uint32_t *blk_r = Q.blk_r[0];
    const int d = idx->dim;
    const int grid = dense_scan_grid(idx);
    const uint32_t *allow = (idx->d_dense_src != nullptr) ? d_allow_bits : nullptr;
    {
        LaunchTimer t(idx, ANRAG_KERNEL_DENSE_SCAN, st, n_queries);
        const bool done = scan_dispatch(d, [&](auto shape) {
            using S = decltype(shape);
#define ANRAG_SCAN(F, SC)                                                                                  \
    dense_scan_kernel<S::kG, S::kCH, S::kR, F, SC><<<grid + (pending ? 1 : 0), kScanThreads, 0, st>>>(        \
        idx->d_emb, Q, idx->n_rows, idx->dim, k, idx->d_dense_src, allow, d_scores_out, scores_stride, dt)
            if (d_scores_out) {
                if (allow) ANRAG_SCAN(true, true); else ANRAG_SCAN(false, true);
            } else {
                if (allow) ANRAG_SCAN(true, false); else ANRAG_SCAN(false, false);
            }
#undef ANRAG_SCAN
        });
        if (!done) {  // odd dimensions: one launch per query
            (void)d_query; (void)blk_s; (void)blk_r;
            for (int i = 0; i < n_queries; ++i)
                dense_scan_topk_generic_kernel<<<grid, kScanThreads, 0, st>>>(
                    idx->d_emb, Q.q[i], n, d, k, idx->d_dense_src, allow, Q.blk_s[i], Q.blk_r[i],
                    d_scores_out ? d_scores_out + (int64_t)i * scores_stride : nullptr);
        }
        ANRAG_HIP(hipGetLastError());
    }
    return ANRAG_OK;
}

int launch_dense_topk(anrag_index *idx, hipStream_t st, const float *d_query, int32_t k,
                      const uint32_t *d_allow_bits, anrag_candidate *d_out, float *d_scores_out) {
    int rc = launch_dense_scan(idx, st, d_query, k, d_allow_bits, d_scores_out, 0);
    if (rc || k <= 0) return rc;
    return launch_tail(idx, st, 0, /*dense*/ true, /*bm25*/ false, k, kTailCandidates, 0, 0, 0, 0, d_out, nullptr);
}

}  // namespace anrag
