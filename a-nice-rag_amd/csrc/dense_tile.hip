// K1T: see the kernel's comment.  Its own translation unit because it is built with -fno-slp-vectorize (Makefile): left to
// itself the compiler pairs the FMA chains of two rows into v_pk_fma_f32 and pays for each pair with register copies
// (1,355 v_mov_b32 for 384 v_pk_fma_f32 in the 768-d kernel) -- 2.4x the instructions of the plain chains.
#include "common.hpp"
#include "wave_topk.hpp"
#include "dense_scan_common.hpp"

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
constexpr int tile_group() { return 3072 / (G * CH) < kTileGroupMax ? 3072 / (G * CH) : kTileGroupMax; }

template <int CH>
constexpr int tile_r() { return CH >= 6 ? 1 : (CH >= 3 ? 2 : (CH == 2 ? 4 : 8)); }  // a power of two: 64 % (R * GROUPS) == 0

template <int G, int CH, int R, bool FILTER>
__global__ __launch_bounds__(kScanThreads) void dense_tile_kernel(
    const float *__restrict__ emb, const float *__restrict__ queries, int64_t q_stride, int32_t n_q,
    int64_t n_rows, int32_t dim, const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits,
    float *__restrict__ scores_out, int64_t scores_stride) {
    constexpr int GROUPS = kWave / G, RW = GROUPS * R, NQ = tile_group<G, CH>();
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
        for (int32_t i = 0; i < n_g; ++i) {
            const f32x4 *qv = reinterpret_cast<const f32x4 *>(queries + (int64_t)(g0 + i) * q_stride);
            for (int32_t j = threadIdx.x; j < row_f4; j += kScanThreads) q_lds[i * row_f4 + j] = qv[j];
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
                    if (i < n_g) out[(int64_t)i * scores_stride + sc_base + lane] = sc[i];
            }
        };
        auto reduce = [&](int64_t base, const Batch &bt) {
            bool ok[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                ok[r] = base + r * GROUPS + grp < n_rows;
                if constexpr (FILTER) ok[r] = ok[r] && source_ok(lds_allow, bt.sid[r]);
            }
            const int slot0 = (int)(base - sc_base);  // the lane that parks the batch's first row
            // Straight-line over the NQ slots of the group: behind the group's last query the slots hold whatever the LDS
            // holds (computed, never stored) -- a branch per query turned the loop into a jump table and cost 40 VGPRs.
            // Query i+1's slices are read from LDS while query i is computed.
            f32x4 q[2][CH];
#pragma unroll
            for (int c = 0; c < CH; ++c) q[0][c] = q_lds[c * G + sub];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                if (i + 1 < NQ) {
#pragma unroll
                    for (int c = 0; c < CH; ++c) q[(i + 1) & 1][c] = q_lds[(i + 1) * row_f4 + c * G + sub];
                }
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    float acc = 0.f;
#pragma unroll
                    for (int c = 0; c < CH; ++c) acc = dot4(bt.v[r][c], q[i & 1][c], acc);
                    acc = nan_first(group_sum<G>(acc));
                    const float val = ok[r] ? acc : neg_inf<float>();
#pragma unroll
                    for (int g = 0; g < GROUPS; ++g) {
                        const float sg = read_lane(val, g * G + G - 1);
                        sc[i] = lane == slot0 + r * GROUPS + g ? sg : sc[i];  // rows past lim: lanes nobody stores
                    }
                }
                __builtin_amdgcn_sched_barrier(0);  // (or the scheduler hoists every query's LDS reads: 300+ VGPRs)
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
