// K1T on the matrix cores, bit for bit the scan's score (dense_scan.hip) -- for row-major corpora of 512 or 768 dimensions
// (the scan's shapes <64, 2> and <64, 3>: G = 64 lanes per row, CH float4 chunks per lane).
//
// K1's score of (row, query) is: per lane s of 64, a chain acc = fma(e, q, acc) over the lane's CH float4s of the row
// (columns (c * 64 + s) * 4 + 0..3, c = 0..CH-1, acc starts at 0), then a balanced binary tree over the 64 lanes in lane
// order.  v_mfma_f32_16x16x4_f32 IS such a chain: D[i][j] = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C)))), one rounding
// per step (scripts/exp/mfma_chain_order.hip: 1,024,000 outputs against host fmaf chains, magnitudes 2^-75 .. 2^60, exact
// cancellations: 0 differ).  So for a tile of 16 rows x 16 queries, one accumulator D_s PER LANE SLICE s (64 of them, 4
// registers each: the whole accumulation file), one MFMA per (chunk c, slice s) with A = the rows' float4 (c, s) and B = the
// queries' -- that is every lane's chain of K1, for 256 (row, query) pairs at once at the matrix pipe's rate (64 flop per
// clock and SIMD: twice the VALU's) and with no lane reduction to pay: the tree is 63 plain vector adds over the
// accumulators (left + right = the scan's own + partner: IEEE addition commutes).
//   VALU K1T (dense_tile.hip): 328 cycles per 2 rows x 2 queries = 82 per pair; here 3 x 64 MFMAs x 32 cycles + ~500 VALU
//   per 256 pairs = ~32 per pair, and the corpus is read once per 32 queries instead of once per 16.
// Workgroup: 8 waves = 2 row tiles x 2 query tiles x 2 halves of the 64 slices (two waves per SIMD: one's tree and waits
// hide under the other's MFMAs; the halves meet through LDS, lower + upper) over a block of 32 rows; the launch's 32 queries stay in LDS for the
// whole kernel (padded rows: a 16 x 4 operand read touches every bank exactly twice), the rows go through LDS one
// 256-column chunk at a time, fetched a whole block ahead into registers.
#include "common.hpp"
#include "wave_topk.hpp"
#include "dense_scan_common.hpp"

namespace anrag {

constexpr int kMfmaTileQueries = 32;  // per workgroup pass (2 query tiles of 16)
// rows per block: 2 row tiles of 16 when a row's 64 slices are split over two waves (G = 64), 4 when a wave takes all 32
// slices of its tile (G = 32: the scan's shape for 384-d rows, two rows per wave there) -- 2,048 float4 per chunk either way
template <int G>
constexpr int mfma_tile_rows() { return G == 64 ? 32 : 64; }

constexpr int kMfmaThreads = 512;  // 8 waves: (row tile, query tile, half of the slices)

template <int G, int CH, bool FILTER>
__global__ __launch_bounds__(kMfmaThreads) void dense_tile_mfma_kernel(
    const float *__restrict__ emb, const float *__restrict__ queries, int64_t q_stride, int32_t n_q, int64_t n_rows,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits, float *__restrict__ scores_out,
    int64_t scores_stride) {
    static_assert(G == 64 || G == 32, "the scan's shapes <64, CH> and <32, CH>");
    constexpr int DIM = 4 * G * CH, T = kMfmaThreads, ROWS = mfma_tile_rows<G>(), HALVES = G / 32, RT = ROWS / 16;
    constexpr int QSTR = DIM + 4, ASTR = 4 * G + 4;  // floats; both = 4 (mod 32): conflict-free 16-lane operand reads
    extern __shared__ __attribute__((aligned(16))) float mt_lds[];
    float *Qs = mt_lds;                                    // [32][QSTR], k-major inside a chunk
    float *As = mt_lds + kMfmaTileQueries * QSTR;          // [ROWS][ASTR]
    f32x4 *Xs = reinterpret_cast<f32x4 *>(As + ROWS * ASTR);  // [4 tiles][64 lanes]: the upper half's partial sums (G = 64)
    uint32_t *lds_allow = reinterpret_cast<uint32_t *>(Xs + 4 * 64);   // [2048] with a filter
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int rt = wave % RT, qt = (wave / RT) & 1, half = wave / (2 * RT);  // 8 waves = RT row tiles x 2 query tiles x HALVES
    const int q0 = blockIdx.y * kMfmaTileQueries;
    const f32x4 *__restrict__ ev = reinterpret_cast<const f32x4 *>(emb);
    constexpr int ROW_F4 = DIM / 4, CHUNK_F4 = G;  // float4s per row / per row and chunk

    // the launch's queries: resident for the whole kernel (a query past the launch's last: the last one again, never stored)
    for (int g = tid; g < kMfmaTileQueries * ROW_F4; g += T) {
        const int q = g / ROW_F4, f = g % ROW_F4;
        const int qq = q0 + q < n_q ? q0 + q : n_q - 1;
        const f32x4 v = reinterpret_cast<const f32x4 *>(queries + (int64_t)qq * q_stride)[f];
        // k-major inside a chunk: element k of float4 (c, s) at [c][k][s], so that a lane's operands for FOUR consecutive
        // slices are one 16-byte read
        float *d = &Qs[q * QSTR + (f / G) * 4 * G + (f % G)];
        d[0] = v.x;
        d[G] = v.y;
        d[2 * G] = v.z;
        d[3 * G] = v.w;
    }
    if constexpr (FILTER)
        for (int i = tid; i < 2048; i += T) lds_allow[i] = allow_bits[i];

    const int64_t n_blocks = (n_rows + ROWS - 1) / ROWS;
    // A block's rows come in CH chunks of 32 x 64 float4 = 4 float4 per thread and chunk, row-major (1 KB runs: coalesced);
    // chunk c of the NEXT block is fetched into register set c right after chunk c of this block has gone to LDS: a whole
    // block (CH steps of MFMAs, ~2 us) ahead of its use -- more than an HBM round trip under load.  The fetches are
    // UNCONDITIONAL (behind the last block: the last block again) and the register sets are static: a load behind a branch
    // makes the compiler wait for every outstanding load at the next use, i.e. for the chunk just issued -- one exposed HBM
    // round trip per step, the pipe a quarter busy.
    const int64_t my_blocks = blockIdx.x < n_blocks ? (n_blocks - blockIdx.x + gridDim.x - 1) / gridDim.x : 0;
    if (my_blocks == 0) return;  // (the whole workgroup)
    f32x4 stage[CH][4];
    // thread t takes float4 t % G of rows t / G + u * (T / G), u = 0..3: one base address, constant offsets (per-u row and LDS
    // addresses kept across the loop were what spilled)
    constexpr int RSTEP = T / G;
    const int r0 = tid / G, f0 = tid % G;
    auto fetch = [&](int64_t blk, int c, f32x4 (&st)[4]) {
        const int64_t row0 = blk * ROWS + r0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            int64_t row = row0 + u * RSTEP;
            row = row < n_rows ? row : n_rows - 1;  // the last block's tail re-reads the last row
            st[u] = __builtin_nontemporal_load(ev + row * ROW_F4 + c * CHUNK_F4 + f0);
        }
    };
    float *put_base = As + r0 * ASTR + f0;
    auto put = [&](const f32x4 (&st)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            float *d = put_base + u * RSTEP * ASTR;  // k-major, as the queries
            d[0] = st[u].x;
            d[G] = st[u].y;
            d[2 * G] = st[u].z;
            d[3 * G] = st[u].w;
        }
    };
    int order = 0;
#pragma unroll
    for (int c = 0; c < CH; ++c) {
        // The loop's counted waits (s_waitcnt vmcnt) assume set 0's loads are the oldest outstanding ones.  The first
        // block's fetches are therefore completed one set after the other (left alone the compiler interleaved them,
        // and every first put of a block then waited for the chunks issued a step before): three round trips, once.
        fetch(blockIdx.x + order, c, stage[c]);
        // (the next set's addresses depend on a 0 produced BEHIND this wait: nothing else keeps loads of read-only,
        // unaliased memory from moving across it)
        if (c + 1 < CH) asm volatile("s_waitcnt vmcnt(0)\n\tv_mov_b32 %0, 0" : "=v"(order) : : "memory");
    }
    const int li = lane & 15, lk = lane >> 4;  // operand lane layout: A[i = li][k = lk], B[k = lk][j = li]
    constexpr int HS = 32, NG = HS / 4;          // slices per wave (a half of the row's at G = 64) and groups of four
    const f32x4 *a_base = reinterpret_cast<const f32x4 *>(As + (rt * 16 + li) * ASTR + lk * G) + half * NG;
    const f32x4 *b_base = reinterpret_cast<const f32x4 *>(Qs + (qt * 16 + li) * QSTR + lk * G) + half * NG;
    for (int64_t bi = 0; bi < my_blocks; ++bi) {
        const int64_t blk = blockIdx.x + bi * gridDim.x;
        const int64_t nxt = blockIdx.x + (bi + 1 < my_blocks ? bi + 1 : bi) * gridDim.x;
        f32x4 acc[HS];
#pragma unroll
        for (int c = 0; c < CH; ++c) {
            __syncthreads();  // the previous chunk's operand reads are done (first pass: the queries are in place)
            put(stage[c]);
            __builtin_amdgcn_sched_barrier(0);  // (the new loads behind the waits of the old ones)
            fetch(nxt, c, stage[c]);
            __syncthreads();
            // operands of four slices per 16-byte read, fetched two groups (8 MFMAs) ahead of their use; the barriers pin
            // that order (left alone the compiler read each pair right in front of its two MFMAs: one exposed LDS round
            // trip per 64 cycles of work).  The first chunk's MFMAs take the literal 0 as C: no accumulators to clear.
            const f32x4 *bq = b_base + c * G;
            constexpr int AH = G == 64 ? 2 : 1;  // groups of operands in flight ahead of the MFMAs (registers: 8 each)
            f32x4 ab[AH + 1], bb[AH + 1];
#pragma unroll
            for (int g = 0; g < AH; ++g) {
                ab[g] = a_base[g];
                bb[g] = bq[g];
            }
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                if (g + AH < NG) {
                    ab[(g + AH) % (AH + 1)] = a_base[g + AH];
                    bb[(g + AH) % (AH + 1)] = bq[g + AH];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int t = 0; t < 4; ++t)
                    acc[4 * g + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(
                        ab[g % (AH + 1)][t], bb[g % (AH + 1)][t], c == 0 ? f32x4{0.f, 0.f, 0.f, 0.f} : acc[4 * g + t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        // the scan's lane tree over this half's 32 slices (neighbours first), then lower half + upper half
#pragma unroll
        for (int w = 1; w < HS; w <<= 1)
#pragma unroll
            for (int s = 0; s < HS; s += 2 * w) acc[s] = acc[s] + acc[s + w];
        f32x4 out = acc[0];
        if constexpr (HALVES == 2) {
            if (half == 1) Xs[(rt * 2 + qt) * 64 + lane] = out;
            __syncthreads();
        }
        if (half == 0) {
            if constexpr (HALVES == 2) out = out + Xs[(rt * 2 + qt) * 64 + lane];
            // lane holds rows (lane / 16) * 4 + 0..3 of the tile for query lane % 16: four consecutive rows of its score row
            const int q = q0 + qt * 16 + li;
            const int64_t row = blk * ROWS + rt * 16 + lk * 4;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                float x = nan_first(out[v]);  // NaN ranks first (carried as +inf)
                if constexpr (FILTER) {
                    const int64_t r = row + v < n_rows ? row + v : n_rows - 1;
                    x = source_ok(lds_allow, src[r]) ? x : neg_inf<float>();
                }
                out[v] = x;
            }
            if (q < n_q) *reinterpret_cast<f32x4 *>(&scores_out[(int64_t)q * scores_stride + row]) = out;
        }
    }
}

// (The kernel is written for G = 32 as well -- the scan's <32, 3> for 384-d rows: a wave then takes all 32 slices of its tile,
// four row tiles per block, no exchange -- but that instantiation does not fit: 8-14 registers spill at 256, inside the loop,
// and spill reloads turn the counted waits into full ones.  Not instantiated; 384-d corpora take the VALU form.)
static int mfma_shape_g(int dim) { return dim == 512 || dim == 768 ? 64 : 0; }

size_t dense_tile_mfma_lds(int dim, bool filter) {
    const int g = mfma_shape_g(dim), rows = g == 64 ? 32 : 64;
    return ((size_t)kMfmaTileQueries * (dim + 4) + (size_t)rows * (4 * g + 4)) * 4 + 4 * 64 * 16 + (filter ? 2048 * 4 : 0);
}

// the dimensions this kernel takes: the scan's shapes <64, 2> and <64, 3> (the others: the VALU form, dense_tile.hip;
// 256-d rows are scanned as <32, 2> and 1,024-d and wider ones leave no room for 32 resident queries)
bool dense_tile_mfma_has_shape(const anrag_index *idx) { return mfma_shape_g(idx->dim) != 0; }
int dense_tile_mfma_block_rows(const anrag_index *idx) { return mfma_shape_g(idx->dim) == 64 ? 32 : 64; }

int launch_dense_tile_mfma(anrag_index *idx, hipStream_t st, const float *d_queries, int64_t q_stride, int32_t n_queries,
                           const uint32_t *d_allow_bits, float *d_scores_out, int64_t scores_stride) {
    ANRAG_REQUIRE(dense_tile_mfma_has_shape(idx), "no matrix-core tile kernel for dimension %d", idx->dim);
    ANRAG_REQUIRE(n_queries >= 1 && n_queries <= kTileLaunchMax, "tile launch of %d queries (at most %d)", n_queries,
                  kTileLaunchMax);
    const int rows = dense_tile_mfma_block_rows(idx);
    ANRAG_REQUIRE(scores_stride >= (idx->n_rows + rows - 1) / rows * rows && scores_stride % 4 == 0,
                  "score tile rows too short for whole row blocks");
    const uint32_t *allow = (idx->d_dense_src != nullptr) ? d_allow_bits : nullptr;
    const int64_t n_blocks = (idx->n_rows + rows - 1) / rows;
    const dim3 grid((unsigned)(n_blocks < idx->n_cus ? n_blocks : idx->n_cus),
                    (unsigned)((n_queries + kMfmaTileQueries - 1) / kMfmaTileQueries));
    const size_t lds = dense_tile_mfma_lds(idx->dim, allow != nullptr);
    int rc;
    LaunchTimer t(idx, ANRAG_KERNEL_DENSE_SCAN, st, n_queries);
#define ANRAG_MT(G_, CH_, F_)                                                                                                \
    do {                                                                                                                     \
        if ((rc = ensure_dynamic_lds(idx->device, reinterpret_cast<const void *>(&dense_tile_mfma_kernel<G_, CH_, F_>),      \
                                     (int)lds)))                                                                             \
            return rc;                                                                                                       \
        dense_tile_mfma_kernel<G_, CH_, F_><<<grid, kMfmaThreads, lds, st>>>(                                                 \
            idx->d_emb, d_queries, q_stride, n_queries, idx->n_rows, idx->d_dense_src, allow, d_scores_out, scores_stride);  \
    } while (0)
    if (idx->dim == 512) {
        if (allow) ANRAG_MT(64, 2, true); else ANRAG_MT(64, 2, false);
    } else {
        if (allow) ANRAG_MT(64, 3, true); else ANRAG_MT(64, 3, false);
    }
#undef ANRAG_MT
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
