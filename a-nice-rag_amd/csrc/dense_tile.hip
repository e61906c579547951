// K1T: see the kernel's comment.  Its own translation unit because it is built with -fno-slp-vectorize (Makefile): left to
// itself the compiler pairs the FMA chains of two rows into v_pk_fma_f32 and pays for each pair with register copies
// (1,355 v_mov_b32 for 384 v_pk_fma_f32 in the 768-d kernel) -- 2.4x the instructions of the plain chains.
#include "common.hpp"
#include "wave_topk.hpp"
#include "dense_scan_common.hpp"

// (m0 carries v_writelane's lane select below: nothing else in this file uses it -- no LDS-DMA, no s_movrel, no messages)
#pragma clang diagnostic ignored "-Winline-asm"

namespace anrag {

// K1T -- score TILES for lists of queries (rank_batch.hip: full ranking, src/retrieval_eval.py:142-143): every score of n
// queries against every row, row-stationary.  The scan above streams the matrix once PER QUERY (that is batch = 1, the
// reference's shape); asked for the scores of 8 queries it read the matrix 8 times, and on the evaluation corpus (9,609
// rows: ~10 rows per wave) each of the 8 passes was one exposed memory round trip: 18 us per launch for 15 MB.  Here a
// wave takes a batch of rows into registers ONCE and runs all n queries over it; the queries sit in LDS (n * dim * 4
// bytes, staged per workgroup), each lane re-reading its own CH float4 slices of query i -- conflict-free 16-byte LDS
// reads.  HBM traffic per query falls by n; what bounds the kernel is VALU: 4*CH FMAs + the lane reduction per
// (row-group, query).  Per row the arithmetic is the scan's, operation for operation (the same per-lane FMA chain over
// the same slices, the same reduction tree): tile scores are bit-identical to K1's, so the ranking of a query list equals
// the ranking of each query alone.
// A wave owns a contiguous block of rows and parks query i's scores in register sc[i] (lane j <- row sc_base + j), stored
// 64 at a time: 256-byte pieces.
constexpr int kTileGroupMax = 16;  // queries per launch: 16 parking registers; n * dim * 4 <= 48 KB of LDS

// queries in LDS at a time: 16 parking registers at most, and n * dim * 4 <= 48 KB (+ 8 KB of filter bits: inside the
// 64 KB a launch gets without asking), dim = 4 * G * CH
template <int G, int CH>
constexpr int tile_group() { return (3072 / (G * CH) < kTileGroupMax ? 3072 / (G * CH) : kTileGroupMax) & ~1; }  // pairs

// group_sum (dense_scan_common.hpp) with its two row_bcast steps as ONE instruction each.  The compiler turns
// `v + update_dpp(0, v, row_bcast, row_mask)` into v_mov 0 / v_mov_dpp / v_add (it has no identity fold for float adds under
// a partial row mask); `v_add_f32_dpp v, v, v row_bcast:15 row_mask:0xa` is the same sum in the rows the mask enables and
// leaves the others alone -- same operands, same rounding, 4 instructions fewer per reduced value.  (The s_nops: a DPP read
// of a VGPR needs two wait states behind the VALU write, a v_readlane one; the hazard recognizer does not look inside asm.)
template <int G>
__device__ __forceinline__ float tile_group_sum(float v) {
    if constexpr (G >= 2) v = dpp_add<0xB1, 0xF>(v);
    if constexpr (G >= 4) v = dpp_add<0x4E, 0xF>(v);
    if constexpr (G >= 8) v = dpp_add<0x141, 0xF>(v);
    if constexpr (G >= 16) v = dpp_add<0x140, 0xF>(v);
    if constexpr (G == 32)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\ts_nop 0" : "+v"(v));
    if constexpr (G == 64)
        asm("s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
            "s_nop 1\n\tv_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\ts_nop 0"
            : "+v"(v));
    return v;
}

// ... of N independent values, step by step across all of them (N - 1 instructions between a write and the DPP read of it:
// no wait states to pad when N >= 3).  Same tree per value.
template <int G, int N>
__device__ __forceinline__ void tile_group_sums(float (&v)[N]) {
    if constexpr (N != 4 && N != 2) {
#pragma unroll
        for (int j = 0; j < N; ++j) v[j] = tile_group_sum<G>(v[j]);
    } else {
#define ANRAG_STEP(CTRL)                                      \
    _Pragma("unroll") for (int j = 0; j < N; ++j) v[j] = dpp_add<CTRL, 0xF>(v[j]);
        if constexpr (G >= 2) { ANRAG_STEP(0xB1) }
        if constexpr (G >= 4) { ANRAG_STEP(0x4E) }
        if constexpr (G >= 8) { ANRAG_STEP(0x141) }
        if constexpr (G >= 16) { ANRAG_STEP(0x140) }
#undef ANRAG_STEP
        if constexpr (N == 4) {
            if constexpr (G >= 32)
                asm("s_nop 1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_bcast:15 row_mask:0xa bank_mask:0xf"
                    : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            if constexpr (G >= 64)
                asm("v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_f32_dpp %2, %2, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_f32_dpp %3, %3, %3 row_bcast:31 row_mask:0xc bank_mask:0xf"
                    : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]));
            if constexpr (G >= 32) asm volatile("s_nop 0" ::: "memory");  // v_readlane behind a VALU write
        } else {
            if constexpr (G >= 32)
                asm("s_nop 1\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n\t"
                    "s_nop 0"
                    : "+v"(v[0]), "+v"(v[1]));
            if constexpr (G >= 64)
                asm("s_nop 0\n\t"
                    "v_add_f32_dpp %0, %0, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "v_add_f32_dpp %1, %1, %1 row_bcast:31 row_mask:0xc bank_mask:0xf\n\t"
                    "s_nop 0"
                    : "+v"(v[0]), "+v"(v[1]));
        }
    }
}

template <int CH>
constexpr int tile_r() { return CH >= 6 ? 1 : (CH >= 3 ? 2 : (CH == 2 ? 4 : 8)); }  // a power of two: 64 % (R * GROUPS) == 0

template <int G, int CH, int R, bool FILTER>
__global__ __launch_bounds__(kScanThreads) void dense_tile_kernel(
    const float *__restrict__ emb, const float *__restrict__ queries, int64_t q_stride, int32_t n_q,
    int64_t n_rows, int32_t dim, const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits,
    float *__restrict__ scores_out, int64_t scores_stride) {
    constexpr int GROUPS = kWave / G, RW = GROUPS * R, NQ = tile_group<G, CH>();
    static_assert(NQ >= 2 && NQ % 2 == 0, "the queries of a group are computed in pairs");
    static_assert(kWave % RW == 0, "a parked block of 64 scores ends at a batch boundary");
    extern __shared__ __attribute__((aligned(16))) float tile_lds[];
    __shared__ uint32_t lds_allow[FILTER ? 2048 : 1];
    f32x4 *q_lds = reinterpret_cast<f32x4 *>(tile_lds);
    const int lane = lane_id();
    const int sub = lane % G, grp = lane / G;
    constexpr int64_t row_f4 = G * CH;  // dim == 4 * G * CH: LDS offsets of the query slices are immediates

    const int64_t total_waves = (int64_t)gridDim.x * kScanWaves;
    // (through readfirstlane: the row range, the parking slots and every branch on them are scalar then)
    const int64_t wave_global = (int64_t)blockIdx.x * kScanWaves + __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t rows_per_wave = ((n_rows + total_waves - 1) / total_waves + RW - 1) / RW * RW;
    const int64_t base0 = wave_global * rows_per_wave;
    const bool has_rows = base0 < n_rows;  // (a wave without rows still stages queries and meets the barriers)
    const int64_t lim = base0 + rows_per_wave < n_rows ? base0 + rows_per_wave : n_rows;

    const f32x4 *__restrict__ ev = reinterpret_cast<const f32x4 *>(emb);
    struct Batch {
        f32x4 v[R][CH];
        uint32_t sid[R];
    };
    auto issue = [&](int64_t base, Batch &bt) {
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const int64_t row = base + r * GROUPS + grp;
            const int64_t rc = row < n_rows ? row : n_rows - 1;  // clamp: tail lanes re-read the last row
            if constexpr (FILTER) bt.sid[r] = src[rc];
            const f32x4 *p = ev + rc * row_f4 + sub;
#pragma unroll
            for (int c = 0; c < CH; ++c) bt.v[r][c] = __builtin_nontemporal_load(p + c * G);
        }
    };
    Batch b0, b1;
    if (has_rows) issue(base0, b0);  // in flight while the first queries are staged
    if constexpr (FILTER)
        for (int i = threadIdx.x; i < 2048; i += kScanThreads) lds_allow[i] = allow_bits[i];

    for (int32_t g0 = 0; g0 < n_q; g0 += NQ) {
        const int32_t n_g = n_q - g0 < NQ ? n_q - g0 : NQ;
        if (g0 > 0) __syncthreads();  // every wave is done with the previous group's slices
        // queries in PAIRS, element-interleaved: float4 j of queries (A, B) as (A.x, B.x, A.y, B.y) at [j] and
        // (A.z, B.z, A.w, B.w) at [row_f4 + j] of the pair's block -- the operand layout of the packed FMAs below
        for (int32_t p = 0; 2 * p < n_g; ++p) {
            const int32_t ia = g0 + 2 * p, ib = 2 * p + 1 < n_g ? ia + 1 : ia;  // an odd group's last pair: (A, A)
            const f32x4 *qa = reinterpret_cast<const f32x4 *>(queries + (int64_t)ia * q_stride);
            const f32x4 *qb = reinterpret_cast<const f32x4 *>(queries + (int64_t)ib * q_stride);
            for (int32_t j = threadIdx.x; j < row_f4; j += kScanThreads) {
                const f32x4 a = qa[j], b = qb[j];
                q_lds[2 * p * row_f4 + j] = f32x4{a.x, b.x, a.y, b.y};
                q_lds[(2 * p + 1) * row_f4 + j] = f32x4{a.z, b.z, a.w, b.w};
            }
        }
        __syncthreads();
        if (!has_rows) continue;
        float *__restrict__ out = scores_out + (int64_t)g0 * scores_stride;
        float sc[NQ];
#pragma unroll
        for (int i = 0; i < NQ; ++i) sc[i] = 0.f;
        int64_t sc_base = base0;
        auto store_parked = [&]() {
            if (sc_base + lane < lim) {
#pragma unroll
                for (int i = 0; i < NQ; ++i)
                    if (i < n_g) out[(int64_t)i * scores_stride + sc_base + lane] = nan_first(sc[i]);  // NaN ranks first
            }
        };
        auto reduce = [&](int64_t base, const Batch &bt) {
            // (rows at or past n_rows are computed like any other: they land in lanes store_parked never stores)
            bool ok[R];
            if constexpr (FILTER) {
#pragma unroll
                for (int r = 0; r < R; ++r) ok[r] = source_ok(lds_allow, bt.sid[r]);
            }
            const int slot0 = (int)(base - sc_base);  // the lane that parks the batch's first row
            // Straight-line over the NQ / 2 query pairs of the group: behind the group's last query the slots hold whatever
            // the LDS holds (computed, never stored) -- a branch per query turned the loop into a jump table and cost 40
            // VGPRs.  Pair i+1's slices are read from LDS while pair i is computed.
            // Two queries per FMA: v_pk_fma_f32 with the row element broadcast to both halves (op_sel) and the pair's
            // elements side by side -- each half is the scan's chain for its query, element by element in the same order.
            f32x4 q[2][2 * CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                q[0][2 * c] = q_lds[c * G + sub];
                q[0][2 * c + 1] = q_lds[row_f4 + c * G + sub];
            }
#pragma unroll
            for (int i = 0; i < NQ / 2; ++i) {
                if (i + 1 < NQ / 2) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) {
                        q[(i + 1) & 1][2 * c] = q_lds[2 * (i + 1) * row_f4 + c * G + sub];
                        q[(i + 1) & 1][2 * c + 1] = q_lds[(2 * (i + 1) + 1) * row_f4 + c * G + sub];
                    }
                }
                // The FMA chains of the batch's R row-groups and the 2 * R reductions behind them are independent of each
                // other: they are written out interleaved, instruction by instruction, so that a wave always has an
                // instruction whose operands are ready (a chain on its own waits out the VALU latency at every step,
                // and a DPP read needs two wait states behind the write: the dependent form was 25 % s_nop).
                f32x2 acc2[R];
#pragma unroll
                for (int r = 0; r < R; ++r) acc2[r] = f32x2{0.f, 0.f};
#pragma unroll
                for (int c = 0; c < CH; ++c) {
                    const f32x4 q01 = q[i & 1][2 * c], q23 = q[i & 1][2 * c + 1];
                    if constexpr (R == 2) {
                        const f32x4 a0 = bt.v[0][c], a1 = bt.v[1][c];
                        asm("v_pk_fma_f32 %0, %2, %6, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                            "v_pk_fma_f32 %1, %4, %6, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                            "v_pk_fma_f32 %0, %2, %7, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %1, %4, %7, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %0, %3, %8, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                            "v_pk_fma_f32 %1, %5, %8, %1 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                            "v_pk_fma_f32 %0, %3, %9, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                            "v_pk_fma_f32 %1, %5, %9, %1 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                            : "+v"(acc2[0]), "+v"(acc2[1])
                            : "v"(a0.xy), "v"(a0.zw), "v"(a1.xy), "v"(a1.zw), "v"(q01.xy), "v"(q01.zw), "v"(q23.xy),
                              "v"(q23.zw));
                    } else {
#pragma unroll
                        for (int r = 0; r < R; ++r) {
                            const f32x4 a = bt.v[r][c];
                            asm("v_pk_fma_f32 %0, %1, %3, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                                "v_pk_fma_f32 %0, %1, %4, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]\n\t"
                                "v_pk_fma_f32 %0, %2, %5, %0 op_sel:[0,0,0] op_sel_hi:[0,1,1]\n\t"
                                "v_pk_fma_f32 %0, %2, %6, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]"
                                : "+v"(acc2[r])
                                : "v"(a.xy), "v"(a.zw), "v"(q01.xy), "v"(q01.zw), "v"(q23.xy), "v"(q23.zw));
                        }
                    }
                }
                asm volatile("s_nop 1" ::: "memory");  // the reduction's DPP reads: two wait states behind the last FMA
                float red[2 * R];  // value 2 * r + h: row-group r, query 2 * i + h
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    red[2 * r] = acc2[r].x;
                    red[2 * r + 1] = acc2[r].y;
                }
                tile_group_sums<G, 2 * R>(red);
#pragma unroll
                for (int r = 0; r < R; ++r) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        float acc = red[2 * r + h];
                        if constexpr (FILTER) acc = ok[r] ? acc : neg_inf<float>();
                        // the leader lane's sum into lane `slot` of the parking register: readlane + writelane (scalar slot)
#pragma unroll
                        for (int g = 0; g < GROUPS; ++g) {
                            const int sum_bits = __builtin_amdgcn_readlane(__float_as_int(acc), g * G + G - 1);
                            // (lane select through m0: one SGPR operand per VALU instruction on gfx9)
                            asm("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0"
                                : "+v"(sc[2 * i + h])
                                : "s"(sum_bits), "s"(slot0 + r * GROUPS + g)
                                : "m0");
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (or the scheduler hoists every pair's LDS reads: 300+ VGPRs)
            }
            if (slot0 + RW == kWave) {
                store_parked();
                sc_base += kWave;
            }
        };
        // two register sets, the next batch issued before this one is reduced (behind the wave's last batch: its
        // first rows again, unconditionally -- see the scan's loop; they are the next group's first batch)
        for (int64_t base = base0;;) {
            int64_t next = base + RW;
            issue(next < lim ? next : base0, b1);
            reduce(base, b0);
            if (next >= lim) {  // b1 holds the first batch again
                b0 = b1;
                break;
            }
            base = next;
            next = base + RW;
            issue(next < lim ? next : base0, b0);
            reduce(base, b1);
            if (next >= lim) break;  // b0 holds the first batch again
            base = next;
        }
        if (sc_base < lim) store_parked();
    }
}

// Queries one K1T launch takes at this index's dimension (0: no tile kernel for it -- generic dimensions and rows of
// 3,072+ floats, whose batches are half rows: launch_dense_scan_group's per-query score passes serve those).
int dense_tile_group_max(const anrag_index *idx) {
    int n = 0;
    scan_dispatch(idx->dim, [&](auto shape) {
        using S = decltype(shape);
        if constexpr (S::kCH < 12) n = tile_group<S::kG, S::kCH>();
    });
    return n;
}

// Scores of n queries (d_queries + i * q_stride floats) against every row: query i's to d_scores_out + i * scores_stride.
// One launch; inside it the queries go through LDS in groups of dense_tile_group_max().
int launch_dense_tile(anrag_index *idx, hipStream_t st, const float *d_queries, int64_t q_stride, int32_t n_queries,
                      const uint32_t *d_allow_bits, float *d_scores_out, int64_t scores_stride) {
    {  // rows of 256 / 512 / 768 floats, corpora that fill the chip: the matrix-core form (dense_tile_mfma.hip), same bits
        static const int mfma = [] { const char *e = getenv("ANRAG_TILE_MFMA"); return e ? atoi(e) : 1; }();  // 0: measurements
        const int64_t br = dense_tile_mfma_has_shape(idx) ? dense_tile_mfma_block_rows(idx) : 1;
        if (mfma && dense_tile_mfma_has_shape(idx) && idx->n_rows >= br * (int64_t)idx->n_cus &&
            scores_stride >= (idx->n_rows + br - 1) / br * br && scores_stride % 4 == 0)
            return launch_dense_tile_mfma(idx, st, d_queries, q_stride, n_queries, d_allow_bits, d_scores_out, scores_stride);
    }
    const int group = dense_tile_group_max(idx);
    ANRAG_REQUIRE(group > 0, "no tile kernel for dimension %d", idx->dim);
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= kTileLaunchMax, "tile launch of %d queries (at most %d)", n_queries,
                  kTileLaunchMax);
    ANRAG_REQUIRE(n_queries == 1 || scores_stride >= idx->n_rows, "score tile rows overlap");
    const uint32_t *allow = (idx->d_dense_src != nullptr) ? d_allow_bits : nullptr;
    static const int per_cu = [] {
        const char *e = getenv("ANRAG_TILE_WGS_PER_CU");  // measurements only
        const int v = e ? atoi(e) : 0;
        return v >= 1 && v <= 8 ? v : 3;
    }();
    const size_t lds = (size_t)group * idx->dim * 4;  // every slot is read, also behind the launch's last query
    LaunchTimer t(idx, ANRAG_KERNEL_DENSE_SCAN, st, n_queries);
    scan_dispatch(idx->dim, [&](auto shape) {
        using S = decltype(shape);
        if constexpr (S::kCH < 12) {
            constexpr int R = tile_r<S::kCH>();
            constexpr int RW = (kWave / S::kG) * R;
            // small corpora: a wave per batch of rows (they stay in its registers for every query group of the launch);
            // large ones: per_cu workgroups per CU
            const int64_t need = (idx->n_rows + kScanWaves * RW - 1) / (kScanWaves * RW);
            const int64_t cap = (int64_t)idx->n_cus * per_cu;
            const unsigned grid = (unsigned)(need < cap ? (need > 0 ? need : 1) : cap);
            if (allow)
                dense_tile_kernel<S::kG, S::kCH, R, true><<<grid, kScanThreads, lds, st>>>(
                    idx->d_emb, d_queries, q_stride, n_queries, idx->n_rows, idx->dim, idx->d_dense_src, allow,
                    d_scores_out, scores_stride);
            else
                dense_tile_kernel<S::kG, S::kCH, R, false><<<grid, kScanThreads, lds, st>>>(
                    idx->d_emb, d_queries, q_stride, n_queries, idx->n_rows, idx->dim, idx->d_dense_src, allow,
                    d_scores_out, scores_stride);
        }
    });
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
