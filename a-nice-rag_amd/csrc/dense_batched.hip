// K2 -- batched queries: S = E . Q^T on the fp32 matrix cores + exact top-k, for Q up to 256 per pass.
//
// No reference counterpart (the reference is batch = 1 only, src/search_engine.py:77-81; the oracle is a
// loop of the single-query path).  BASELINE.json config "1M x 768 dense, batch=256 queries (MFMA
// batched-GEMM path)".  Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-for-bit an
// fmaf chain (k order permuted), so scores stay within the 1e-4 bar of the fp32 reference; no bf16/fp16
// rounding of the corpus.  2*Q*N*D = 3.93e11 flop per 256-query pass vs 157.3 TFLOP/s dense f32 peak:
// MFMA-bound (the 3.07 GB corpus is read once per pass: 1.2 TB/s at peak rate).
//
// Exact top-k without materialising the Q x N score matrix (1 GB at 256 x 1M):
//   1. SAMPLE pass   the same GEMM over every (N/S)-th corpus row; each lane keeps the best two allowed scores of
//                    its rows per query column (dense_batched_common.hpp), a workgroup per query takes the k-th
//                    best of those, tau_q.  The k-th best over ALL rows is >= the k-th best over a subset, so
//                    tau_q is a valid lower bound.
//   2. FILTER pass   the full GEMM; the epilogue appends (score,row) with score >= tau_q to the
//                    query's candidate list (one global atomic per survivor; ~N*k/S survivors per query).
//   3. SELECT        a wave per query ranks its survivors (score desc, row asc) -> k records.
//   A query whose list overflows its capacity is flagged (count = -1) and redone by K1.
//
// Tiling (wave64): workgroup tile = 128 corpus rows x QW queries, BK = 32, 2 waves per SIMD on the CU either as
// one 8-wave workgroup (QW = 256) or as two 4-wave workgroups (QW = 128) -- BatchGeom below.  Corpus rows are the
// MFMA A operand (rows of D), queries the B operand (columns of D), so a LANE owns one query column and its
// threshold.  Wave w: rows (w&1)*64.., queries (w>>1)*64.. -> 2x2 tiles of 32x32 = 64 accumulators.  LDS images
// [row][32 k + 4 pad] fp32 (pad breaks the 128-B row stride for ds_read_b128), double buffered (110 / 74 KB per
// workgroup); global -> registers -> LDS staging two k-steps ahead.  Each lane reads FOUR
// consecutive k per ds_read_b128 and feeds them to four MFMAs: MFMA s of group g sums k = 8g+s (lane
// half 0) and k = 8g+4+s (lane half 1) -- a permutation of k, identical on the A and B side.
#include "dense_batched_common.hpp"

namespace anrag {

constexpr int kLdk = kBK + 4;    // padded LDS row (floats)
// QW = queries per workgroup.  256: one 8-wave workgroup per CU (2 waves per SIMD), the whole pass against each
// corpus tile -- fewest staged bytes per MFMA, best for full passes.  128: 4-wave workgroups, two per CU; a pass of
// <= 128 queries does half the matrix work instead of multiplying zero padding (measured, 1M x 768: 256 queries
// 3.24 ms with QW=256 vs 3.5 ms with 2 x QW=128; 128 queries 1.75 ms with QW=128).
template <int QW>
struct BatchGeom {
    static constexpr int kThreads = QW * 2;                 // waves = 2 (row halves) x QW/64 (query columns)
    static constexpr int kBufFloats = (kBM + QW) * kLdk;
    static constexpr int kStageBytes = 2 * kBufFloats * 4;
    // survivor slices (dense_batched_common.hpp): 4 KB per wave for the one-workgroup-per-CU shape, 1 KB per wave
    // where two workgroups share the CU's 160 KB
    static constexpr int kSurvSlice = QW == 256 ? 4096 : 1024;
    static constexpr int kSurvEntries = QW == 256 ? 448 : 112;
    static constexpr int kLdsBytes = kStageBytes + (kThreads / 64) * kSurvSlice;
    static constexpr int kStageE = kBM * 8 / kThreads;      // float4 per thread per k-step, corpus tile
    static constexpr int kStageQ = QW * 8 / kThreads;       // = 4
};

template <int QW, bool SAMPLE, bool FILTER>
__global__ __launch_bounds__(BatchGeom<QW>::kThreads, 2) void dense_batched_kernel(
    const float *__restrict__ emb, const float *__restrict__ queries /* [kBQ][dim], zero padded */, int32_t n_qblocks,
    int64_t n_rows /* corpus rows */, int32_t dim, int32_t nq, int64_t n_work /* rows this pass visits */,
    int64_t stride /* SAMPLE: corpus row = work row * stride */, const float *__restrict__ tau,
    float *__restrict__ sample_scores /* [kBQ][tiles * 4 slots][2] */, int32_t *__restrict__ cnt, Cand32 *__restrict__ cand,
    int32_t cap, const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits) {
    using Geo = BatchGeom<QW>;
    constexpr int kBatchThreads = Geo::kThreads, kBufFloats = Geo::kBufFloats, kBQW = QW;
    constexpr int NE = Geo::kStageE, NQ = Geo::kStageQ;
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rw = wave & 1, qw = wave >> 1;
    // the source filter is consulted for survivors only (rare), straight from the L2-resident bitmap: keeping it
    // in LDS would push two workgroups past the CU's 160 KB
    const int ksteps = dim / kBK;
    const int64_t n_tiles = (n_work + kBM - 1) / kBM;
    // workgroup -> (query block, first corpus tile); tiles are strided by the workgroups of the same query block
    const int qblock = blockIdx.x % n_qblocks;
    const int64_t first_tile = blockIdx.x / n_qblocks;
    const int64_t tile_step = gridDim.x / n_qblocks;
    const int qbase = qblock * kBQW;
    const int64_t my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + tile_step - 1) / tile_step : 0;
    const int64_t total = my_tiles * ksteps;

    // this lane's thresholds: query columns qw*64 + tj*32 + l31
    float my_tau[2] = {0.f, 0.f};
    if constexpr (!SAMPLE) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {  // padding queries (zero rows of the block) never keep a score
            const int q = qbase + qw * 64 + tj * 32 + l31;
            my_tau[tj] = q < nq ? tau[q] : 3.0e38f;
        }
    }
    int surv_fill = 0;  // wave-uniform: entries in this wave's survivor slice
    unsigned char *slice = reinterpret_cast<unsigned char *>(lds) + Geo::kStageBytes + wave * Geo::kSurvSlice;

    // Staging cursor: (tile, k-step) of the NEXT load_stage call, advanced incrementally -- a 64-bit divide and
    // six 64-bit address multiplies per k-step were ~10 % of the loop before.
    f32x4 stage_e[NE], stage_q[NQ];
    int64_t ld_tile = first_tile;
    int ld_ks = 0;
    const float *pe[NE];
    const float *pq[NQ];
    auto point_rows = [&]() {
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int f = tid + i * kBatchThreads;
            int64_t wr = ld_tile * kBM + (f >> 3);
            if (wr >= n_work) wr = n_work - 1;
            const int64_t row = SAMPLE ? wr * stride : wr;
            pe[i] = emb + row * dim + (f & 7) * 4;
        }
    };
#pragma unroll
    for (int i = 0; i < NQ; ++i) {
        const int f = tid + i * kBatchThreads;
        pq[i] = queries + (int64_t)(qbase + (f >> 3)) * dim + (f & 7) * 4;
    }
    point_rows();
    // one staged float4 (item j < NE: corpus tile, else query tile): global -> register, register -> LDS
    auto load_item = [&](int j) {
        if (j < NE) stage_e[j] = *reinterpret_cast<const f32x4 *>(pe[j] + ld_ks * kBK);
        else stage_q[j - NE] = *reinterpret_cast<const f32x4 *>(pq[j - NE] + ld_ks * kBK);
    };
    auto advance_cursor = [&]() {
        if (++ld_ks == ksteps) {
            ld_ks = 0;
            ld_tile += tile_step;
            point_rows();
        }
    };
    auto store_item = [&](int j, int buf) {
        float *es = lds + buf * kBufFloats;
        const int f = tid + (j < NE ? j : j - NE) * kBatchThreads;
        float *dst = (j < NE ? es : es + kBM * kLdk) + (f >> 3) * kLdk + (f & 7) * 4;
        if (j < NE) {
            *reinterpret_cast<f32x4 *>(dst) = stage_e[j];
        } else {
            // query rows are stored with every group of four k rotated by two: the MFMA of k-slot s then takes A from
            // register s and B from register (s + 2) & 3 of their (4-aligned) fragment tuples.  Three functionally
            // identical builds of this kernel ran 3.09 / 3.50 / 3.73 ms depending only on the registers the compiler
            // picked for the fragments (A and B of every MFMA in the same VGPR bank in all three); with this pairing
            // every build since has run 3.12-3.15 ms.  Why is not established (a bare MFMA loop does not care:
            // scripts/exp/mfma_f32_banks.hip).
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const f32x4 x = stage_q[j - NE];
            *reinterpret_cast<f32x2 *>(dst) = f32x2{x[2], x[3]};
            *reinterpret_cast<f32x2 *>(dst + 2) = f32x2{x[0], x[1]};
        }
    };
    auto load_stage = [&]() {
#pragma unroll
        for (int j = 0; j < NE + NQ; ++j) load_item(j);
        advance_cursor();
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int j = 0; j < NE + NQ; ++j) store_item(j, buf);
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    // Staging pipeline, two steps deep: the registers always hold step it+1 while step it is being multiplied.
    // During step it they are written to the OTHER LDS buffer (free since the barrier that closed step it-1) and
    // re-loaded with step it+2.  Both are dealt out ONE instruction at a time between this step's MFMAs: a wave
    // issues in order, and the two waves of a SIMD leave every barrier together, so a block of staging code at
    // the top of the step left the matrix pipe idle for its whole length.
    if (total > 0) {
        load_stage();
        store_stage(0);
        if (total > 1) load_stage();
    }
    __syncthreads();
    int64_t cur_tile = first_tile;
    int cur_ks = 0;
    for (int64_t it = 0; it < total; ++it) {
        const int buf = (int)(it & 1);
        // (stores and loads run on the last steps too -- the cursor clamps to the last row, nobody reads the
        // buffer: a branch around them would make the compiler wait for vmcnt(0) at every slot)
        const float *es = lds + buf * kBufFloats + (rw * 64 + l31) * kLdk + lh * 4;
        const float *qs = lds + buf * kBufFloats + kBM * kLdk + (qw * 64 + l31) * kLdk + lh * 4;
        // operand fragments of k-group g+1 are read from LDS while the 16 MFMAs of group g issue (separate
        // registers: no write-after-read wait on the fragments the matrix pipe is still consuming)
        f32x4 fa[2][2], fb[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            fa[0][t] = *reinterpret_cast<const f32x4 *>(es + t * 32 * kLdk);
            fb[0][t] = *reinterpret_cast<const f32x4 *>(qs + t * 32 * kLdk);
        }
#pragma unroll
        for (int g = 0; g < kBK / 8; ++g) {
            const int cur = g & 1, nxt = cur ^ 1;
            if (g + 1 < kBK / 8) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    fa[nxt][t] = *reinterpret_cast<const f32x4 *>(es + t * 32 * kLdk + (g + 1) * 8);
                    fb[nxt][t] = *reinterpret_cast<const f32x4 *>(qs + t * 32 * kLdk + (g + 1) * 8);
                }
            }
            __builtin_amdgcn_sched_barrier(0);  // keep the prefetch ahead of this group's MFMAs
#pragma unroll
            for (int s = 0; s < 4; ++s) {
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[cur][ti][s], fb[cur][tj][(s + 2) & 3], acc[ti][tj], 0, 0, 0);
                // staging slots: (0,0..1) write last step's registers to LDS, (0,2..3) refill them -- as early in
                // the step as the order store -> load allows, so a load has almost a whole step to land
                if (g == 0) {
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int j = 0; j < NE + NQ; ++j) {
                        if (s < 2 && j % 2 == s) store_item(j, buf ^ 1);
                        if (s >= 2 && j % 2 == s - 2) load_item(j);
                    }
                    if (s == 3) advance_cursor();
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (++cur_ks == ksteps) {
            batched_tile_epilogue<SAMPLE, FILTER, 2, Geo::kSurvEntries>(acc, cur_tile, rw, qw, qbase, l31, lh, my_tau,
                                                                        n_work, stride, nq, sample_scores, cnt, cand, cap,
                                                                        src, allow_bits, slice, surv_fill);
            cur_ks = 0;
            cur_tile += tile_step;
        }
        __syncthreads();
    }
    if constexpr (!SAMPLE) {
        if (surv_fill) flush_survivors(slice, Geo::kSurvEntries, surv_fill, cnt, cand);
    }
}

// tau[q] = k-th best sampled score of query q (or -inf when fewer than k allowed rows were sampled)
// One workgroup of 8 waves per query: the k-th best score of its sample row.  Each wave takes every eighth chunk of
// 1,024 scores with sixteen independent loads per lane in flight (one wave and one load at a time, this kernel cost as
// much as half the sampled pass), the eight lists meet through LDS.
__global__ __launch_bounds__(kThrWaves * 64) void batched_threshold_kernel(const float *__restrict__ sample_scores,
                                                                           int64_t n_sample, int32_t k,
                                                                           float *__restrict__ tau,
                                                                           int32_t *__restrict__ cnt) {
    __shared__ float lds_s[kThrWaves * kListLen];
    __shared__ uint32_t lds_r[kThrWaves * kListLen];
    const int q = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveTopK<float> top;
    top.init(k);
    const float *s = sample_scores + (int64_t)q * n_sample;
    constexpr int kDeep = 16;  // independent loads per lane: the scan is a chain of HBM / L2 round trips otherwise
    for (int64_t i0 = (int64_t)wave * (kDeep * 64); i0 < n_sample; i0 += kThrWaves * kDeep * 64) {
        float v[kDeep];
#pragma unroll
        for (int j = 0; j < kDeep; ++j) {
            const int64_t i = i0 + j * 64 + lane;
            v[j] = i < n_sample ? s[i] : neg_inf<float>();
        }
        // the best of each lane's sixteen first: one sorted offer lifts the threshold so far that the other fifteen
        // rounds rarely pass admits() (offering sixteen full rounds to an empty list is sixteen sort-and-merge networks)
        float m = v[0];
        int jm = 0;
#pragma unroll
        for (int j = 1; j < kDeep; ++j)
            if (v[j] > m) {
                m = v[j];
                jm = j;
            }
        const uint32_t im = (uint32_t)(i0 + jm * 64 + lane);  // rows are only a tie-break here: the sample index
        top.offer_lanes(m > neg_inf<float>() && top.admits(m, im), m, im);
#pragma unroll
        for (int j = 0; j < kDeep; ++j) {
            const uint32_t i = (uint32_t)(i0 + j * 64 + lane);
            top.offer_lanes(j != jm && v[j] > neg_inf<float>() && top.admits(v[j], i), v[j], i);
        }
    }
    block_merge(top, lds_s, lds_r, kThrWaves);
    if (threadIdx.x == 0) {
        tau[q] = top.thr_r == kNoRow ? neg_inf<float>() : top.thr_s;
        cnt[q] = 0;
    }
}

// rank one query's survivors -> k records; count = -1 flags an overflowed list (caller redoes the query with K1)
__global__ __launch_bounds__(64) void batched_select_kernel(const Cand32 *__restrict__ cand,
                                                            const int32_t *__restrict__ cnt, int32_t cap, int32_t k,
                                                            const int64_t *__restrict__ doc_of_row, int64_t doc_base,
                                                            anrag_candidate *__restrict__ out,
                                                            int32_t *__restrict__ out_flag) {
    const int q = blockIdx.x, lane = threadIdx.x;
    const int32_t n = cnt[q];
    WaveTopK<float> top;
    top.init(k);
    const Cand32 *c = cand + (int64_t)q * cap;
    const int32_t m = n < cap ? n : cap;
    for (int32_t i0 = 0; i0 < m; i0 += kWave) {
        const int32_t i = i0 + lane;
        Cand32 v;
        v.score = neg_inf<float>();
        v.row = kNoRow;
        if (i < m) v = c[i];
        top.offer_lanes(i < m && top.admits(v.score, v.row), v.score, v.row);
    }
    if (lane < k) {
        anrag_candidate r;
        const bool empty = top.r == kNoRow;
        r.score = empty ? -__builtin_huge_val() : (double)top.s;
        r.doc = empty ? -1 : (doc_of_row ? doc_of_row[top.r] : doc_base + (int64_t)top.r);
        out[(int64_t)q * k + lane] = r;
    }
    if (lane == 0) out_flag[q] = n > cap ? -1 : 0;
}

// ------------------------------------------------------------------ host side
bool batched_path_applies(const anrag_index *idx, int32_t n_queries, int32_t k) {
    return n_queries >= 16 && k <= ANRAG_FUSED_K_MAX && idx->dim % kBK == 0 && idx->n_rows >= 65536 &&
           idx->n_rows < 0xffffffffLL;  // survivor records carry 32-bit rows
}

static int ensure_batched_workspace(anrag_index *idx, int64_t n_sample /* floats per query */) {
    if (!idx->d_bq) {
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_bq), (size_t)kBQ * idx->dim * sizeof(float)));
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_btau), kBQ * sizeof(float)));
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_bcnt), kBQ * sizeof(int32_t)));
        ANRAG_HIP(hipMemset(idx->d_bcnt, 0, kBQ * sizeof(int32_t)));  // the counters of padding queries are never reset
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_bflag), kBQ * sizeof(int32_t)));
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_bcand), (size_t)kBQ * kCandCap * sizeof(Cand32)));
        idx->hbm_bytes += (int64_t)kBQ * idx->dim * 4 + (int64_t)kBQ * kCandCap * 8;
    }
    if (idx->bsample_cap < n_sample) {
        if (idx->d_bsample) (void)counted_free(idx->d_bsample);
        idx->d_bsample = nullptr;
        idx->bsample_cap = 0;
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_bsample), (size_t)kBQ * n_sample * sizeof(float)));
        idx->bsample_cap = n_sample;
    }
    return ANRAG_OK;
}

void free_batched(anrag_index *idx) {
    void *ptrs[] = {idx->d_bq,    idx->d_btau,    idx->d_bcnt,  idx->d_bflag,
                    idx->d_bcand, idx->d_bsample, idx->d_bq_hi, idx->d_bq_lo, idx->d_split_img, idx->d_bq_img};
    for (void *p : ptrs)
        if (p) (void)counted_free(p);
    idx->d_bq = idx->d_btau = idx->d_bsample = nullptr;
    idx->d_bcnt = idx->d_bflag = nullptr;
    idx->d_bcand = idx->d_bq_hi = idx->d_bq_lo = idx->d_split_img = idx->d_bq_img = nullptr;
    idx->hbm_bytes -= idx->split_img_bytes;
    idx->split_img_bytes = 0;
    idx->bsample_cap = 0;
}

template <int QW>
static int batched_attrs(int device) {  // per device, under a mutex (common.hpp: ensure_dynamic_lds)
    using Geo = BatchGeom<QW>;
    for (const void *f : {reinterpret_cast<const void *>(&dense_batched_kernel<QW, true, false>),
                          reinterpret_cast<const void *>(&dense_batched_kernel<QW, true, true>),
                          reinterpret_cast<const void *>(&dense_batched_kernel<QW, false, false>),
                          reinterpret_cast<const void *>(&dense_batched_kernel<QW, false, true>)})
        if (int rc = ensure_dynamic_lds(device, f, Geo::kLdsBytes)) return rc;
    return ANRAG_OK;
}

// workgroups for a pass over `rows` corpus rows in geometry QW: one per tile and query block, capped at what the
// chip holds at 8 waves per CU
template <int QW>
static unsigned batched_grid(const anrag_index *idx, int64_t rows, int n_qblocks) {
    const int wg_per_cu = 512 / BatchGeom<QW>::kThreads;
    const int64_t tiles = (rows + kBM - 1) / kBM;
    const int64_t per_block = (int64_t)wg_per_cu * idx->n_cus / n_qblocks;
    return (unsigned)((tiles < per_block ? tiles : per_block) * n_qblocks);
}

template <int QW>
static int batched_passes(anrag_index *idx, hipStream_t st, int32_t nq, int32_t k, int64_t n_sample, int64_t stride,
                          const uint32_t *allow, Cand32 *cand) {
    using Geo = BatchGeom<QW>;
    using GeoS = BatchGeom<128>;
    int rc;
    if ((rc = batched_attrs<QW>(idx->device)) || (rc = batched_attrs<128>(idx->device))) return rc;
    const int64_t n = idx->n_rows;
    const int dim = idx->dim;
    const int n_qblocks = (nq + QW - 1) / QW;  // 1
    // The sampled pass is a few thousand rows = a few dozen tiles: its duration is ONE tile's latency, whatever the
    // geometry's throughput.  It always runs in the 128-query geometry (half the matrix work per workgroup and
    // k-step, one wave per SIMD): ~50 us instead of ~100 us for 256 queries.
    const int sq = (nq + 127) / 128;
    if (allow)
        dense_batched_kernel<128, true, true><<<batched_grid<128>(idx, n_sample, sq), GeoS::kThreads, GeoS::kLdsBytes, st>>>(
            idx->d_emb, idx->d_bq, sq, n, dim, nq, n_sample, stride, nullptr, idx->d_bsample, nullptr, nullptr, 0,
            idx->d_dense_src, allow);
    else
        dense_batched_kernel<128, true, false><<<batched_grid<128>(idx, n_sample, sq), GeoS::kThreads, GeoS::kLdsBytes, st>>>(
            idx->d_emb, idx->d_bq, sq, n, dim, nq, n_sample, stride, nullptr, idx->d_bsample, nullptr, nullptr, 0,
            nullptr, nullptr);
    batched_threshold_kernel<<<nq, kThrWaves * 64, 0, st>>>(idx->d_bsample, sample_floats_per_query(n_sample, kBM), k,
                                                             idx->d_btau, idx->d_bcnt);
    if (allow)
        dense_batched_kernel<QW, false, true><<<batched_grid<QW>(idx, n, n_qblocks), Geo::kThreads, Geo::kLdsBytes, st>>>(
            idx->d_emb, idx->d_bq, n_qblocks, n, dim, nq, n, 1, idx->d_btau, nullptr, idx->d_bcnt, cand, kCandCap,
            idx->d_dense_src, allow);
    else
        dense_batched_kernel<QW, false, false><<<batched_grid<QW>(idx, n, n_qblocks), Geo::kThreads, Geo::kLdsBytes, st>>>(
            idx->d_emb, idx->d_bq, n_qblocks, n, dim, nq, n, 1, idx->d_btau, nullptr, idx->d_bcnt, cand, kCandCap, nullptr,
            nullptr);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

// One pass of up to 256 queries (device pointers); d_out: nq x k records, d_flag: nq ints (0 ok, -1 redo).
int launch_dense_batched(anrag_index *idx, hipStream_t st, const float *d_queries, int32_t nq, int32_t k,
                         const uint32_t *d_allow_bits, anrag_candidate *d_out, int32_t *d_flag) {
    ANRAG_REQUIRE(nq >= 1 && nq <= kBQ, "a batched pass takes 1..%d queries", kBQ);
    const int64_t n = idx->n_rows;
    const int dim = idx->dim;
    const uint32_t *allow = idx->d_dense_src ? d_allow_bits : nullptr;
    // Sample size.  A score survives the full pass when it reaches the k-th best of the S sampled rows: N*k/S expected
    // survivors per query.  The sampled pass costs ~1 ns per row (beyond the ~64k rows that merely fill the CUs), a
    // survivor ~0.8 ns x 256 queries: the sum is smallest near S = sqrt(128 N k) (36k rows, ~280 survivors per query at
    // 1M x k=10; a fixed N*k/S = cap/4 = 2,048 made the survivors 6 % of the f32 pass and 30 % of the split one).  The
    // f32 pass costs ~3 ns per row and its survivors less since they go through LDS: sqrt(24 N k) (one workgroup per CU
    // and query block at 1M x k=10: 16,384 rows, 3.12 -> 3.09 ms per pass against two per CU).
    // Never fewer rows than keep the expected survivors under cap/4.
    int64_t n_sample = (int64_t)__builtin_sqrt((idx->batched_split ? 128.0 : 24.0) * (double)n * (double)k);
    const int64_t floor_rows = (4 * n * (int64_t)k + kCandCap - 1) / kCandCap;
    if (n_sample < floor_rows) n_sample = floor_rows;
    if (n_sample < 4096) n_sample = 4096;
    if (n_sample > n) n_sample = n;
    // a round of tiles (one per CU) costs the same whether the tiles are all there or not: fill the round, as long as the
    // sample stays a small part of the corpus
    // (f32: the sampled pass runs the 128-query geometry, two workgroups per CU when there are more than 128 queries --
    // a "round" is then one workgroup per CU per query block = half as many rows)
    const int64_t round_rows = idx->batched_split ? (int64_t)idx->n_cus * 256 : (int64_t)idx->n_cus * kBM / (nq > 128 ? 2 : 1);
    const int64_t filled = (n_sample + round_rows - 1) / round_rows * round_rows;
    if (filled <= n / 8) n_sample = filled;
    const int64_t stride = n / n_sample;
    int rc = ensure_batched_workspace(idx, sample_floats_per_query(n_sample, idx->batched_split ? 256 : kBM));
    if (rc) return rc;
    // zero-padded query block
    ANRAG_HIP(hipMemsetAsync(idx->d_bq, 0, (size_t)kBQ * dim * sizeof(float), st));
    ANRAG_HIP(hipMemcpyAsync(idx->d_bq, d_queries, (size_t)nq * dim * sizeof(float), hipMemcpyDeviceToDevice, st));
    Cand32 *cand = reinterpret_cast<Cand32 *>(idx->d_bcand);
    {
        LaunchTimer t(idx, ANRAG_KERNEL_DENSE_BATCHED, st);
        int rc2 = idx->batched_split ? batched_passes_split(idx, st, nq, k, n_sample, stride, allow, cand)
                  : nq > 128       ? batched_passes<256>(idx, st, nq, k, n_sample, stride, allow, cand)
                                   : batched_passes<128>(idx, st, nq, k, n_sample, stride, allow, cand);
        if (rc2) return rc2;
    }
    {
        LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
        batched_select_kernel<<<nq, 64, 0, st>>>(reinterpret_cast<const Cand32 *>(idx->d_bcand), idx->d_bcnt, kCandCap, k,
                                                 idx->d_dense_doc, idx->dense_doc_base, d_out, d_flag ? d_flag : idx->d_bflag);
        ANRAG_HIP(hipGetLastError());
    }
    return ANRAG_OK;
}

}  // namespace anrag
