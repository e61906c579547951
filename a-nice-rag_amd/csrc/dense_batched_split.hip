// K2, split-precision arithmetic (opt-in: anrag_set_batched_precision(idx, 1)) -- the same pass structure as
// dense_batched.hip (sampled threshold, filtered full pass, per-query select), but the products run on the
// bf16 matrix cores at 16x the f32 MFMA rate:
//
//     x = hi(x) + lo(x),  hi = bf16(x),  lo = bf16(x - hi)          (both round-to-nearest-even)
//     a . b  ~=  hi(a).hi(b) + hi(a).lo(b) + lo(a).hi(b)             three v_mfma_f32_32x32x16_bf16, f32 accumulate
//
// The dropped lo.lo term and the rounding of lo are each <= 2^-16 |a_i||b_i| per product, so for the unit-norm
// rows and queries of this path |error| <= ~3e-5 * sum|a_i b_i| <= 3e-5 (Cauchy-Schwarz) -- inside the 1e-4 bar
// BASELINE.json sets for the dense side (measured: < 1e-6), but NOT the bit-level f32 result: that is why it is
// opt-in and the default stays the exact f32 MFMA kernel.
//
// Two kernels.  The SAMPLED pass (a few per cent of the rows, strided) reads the fp32 corpus and splits each tile
// while it is staged to LDS -- dense_batched_split_kernel, first below.  The FULL pass streams the corpus from a
// second copy kept as ready-to-copy bf16 hi / lo images (+4 bytes per element of HBM, built on the first
// split-precision pass after a load) with LDS-DMA and ping-pong waves -- dense_batched_split_dma_kernel, further
// down, with its own notes.  Both accumulate every score in the same order (per 16-wide k-step: lo.hi, hi.lo, hi.hi),
// so a sampled row scores bit-identically in both passes and the sampled threshold is an exact lower bound.
//
// Sampled-pass tiling: 8 waves, workgroup tile 256 corpus rows x 256 queries, wave w -> rows (w&1)*128.., queries
// (w>>1)*64..: 4 x 2 accumulator tiles of 32 x 32, 48 MFMAs per wave and k-step of 32.  LDS images are bf16
// [row][32 k] = 64-byte rows whose four 16-byte chunks are XOR-swizzled with (row >> 2) & 3, which makes the
// ds_read_b128 of one 8-element fragment per lane (lane l -> row l&31, k = 8*(l>>5) .. +8) conflict-free
// without padding; four images per buffer (corpus hi / lo, query hi / lo), double buffered: 128 KB.
#include "dense_batched_common.hpp"

namespace anrag {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));

constexpr int kSplitThreads = 512;
constexpr int kSM = 256;                                   // corpus rows per workgroup tile
constexpr int kRowB = kBK * 2;                             // 64 bytes per LDS row (32 bf16)
constexpr int kImgE = kSM * kRowB;                         // one corpus image
constexpr int kImgQ = kBQ * kRowB;                         // one query image
constexpr int kSplitBuf = 2 * kImgE + 2 * kImgQ;           // hi + lo of both
constexpr int kSplitLdsBytes = 2 * kSplitBuf;              // double buffered

// byte offset of 16-byte chunk `c` (0..3) of LDS row `row`
__device__ __forceinline__ int swz(int row, int c) { return row * kRowB + ((c ^ ((row >> 2) & 3)) << 4); }

__global__ void split_queries_kernel(const float *__restrict__ q, int64_t n, __bf16 *__restrict__ hi,
                                     __bf16 *__restrict__ lo) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float x = q[i];
    const __bf16 h = (__bf16)x;
    hi[i] = h;
    lo[i] = (__bf16)(x - (float)h);
}

template <bool SAMPLE, bool FILTER>
__global__ __launch_bounds__(kSplitThreads, 2) void dense_batched_split_kernel(
    const float *__restrict__ emb, const __bf16 *__restrict__ q_hi, const __bf16 *__restrict__ q_lo, int32_t dim,
    int32_t nq, int64_t n_work, int64_t stride, const float *__restrict__ tau, float *__restrict__ sample_scores,
    int32_t *__restrict__ cnt, Cand32 *__restrict__ cand, int32_t cap, const uint16_t *__restrict__ src,
    const uint32_t *__restrict__ allow_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rw = wave & 1, qw = wave >> 1;
    const int ksteps = dim / kBK;
    const int64_t n_tiles = (n_work + kSM - 1) / kSM;
    const int64_t first_tile = blockIdx.x, tile_step = gridDim.x;
    const int64_t my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + tile_step - 1) / tile_step : 0;
    const int64_t total = my_tiles * ksteps;

    float my_tau[2] = {0.f, 0.f};
    if constexpr (!SAMPLE) {
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) my_tau[tj] = tau[qw * 64 + tj * 32 + l31];
    }

    // staging: corpus 256 rows x 8 float4 = 2048 float4 -> 4 per thread; each query image 256 rows x 4 chunks of
    // 16 B = 1024 -> 2 per thread per image
    f32x4 st_e[4];
    bf16x8 st_qh[2], st_ql[2];
    int64_t ld_tile = first_tile;
    int ld_ks = 0;
    const float *pe[4];
    auto point_rows = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int f = tid + i * kSplitThreads;
            int64_t wr = ld_tile * kSM + (f >> 3);
            if (wr >= n_work) wr = n_work - 1;
            const int64_t row = SAMPLE ? wr * stride : wr;
            pe[i] = emb + row * dim + (f & 7) * 4;
        }
    };
    point_rows();
    // staged items: 0..3 corpus float4 (split into hi / lo while they are written to LDS), 4..5 query chunks
    auto load_corpus = [&](int i) { st_e[i] = *reinterpret_cast<const f32x4 *>(pe[i] + ld_ks * kBK); };
    auto load_query = [&](int i) {
        const int f = tid + i * kSplitThreads;  // query row f>>2, 8-element chunk f&3
        const int64_t off = (int64_t)(f >> 2) * dim + ld_ks * kBK + (f & 3) * 8;
        st_qh[i] = *reinterpret_cast<const bf16x8 *>(q_hi + off);
        st_ql[i] = *reinterpret_cast<const bf16x8 *>(q_lo + off);
    };
    auto advance_cursor = [&]() {
        if (++ld_ks == ksteps) {
            ld_ks = 0;
            ld_tile += tile_step;
            point_rows();
        }
    };
    auto store_corpus = [&](int i, int buf) {
        unsigned char *base = lds + buf * kSplitBuf;
        const int f = tid + i * kSplitThreads;
        const f32x4 x = st_e[i];
        const bf16x4 h = __builtin_convertvector(x, bf16x4);
        const f32x4 r = x - __builtin_convertvector(h, f32x4);
        const bf16x4 l = __builtin_convertvector(r, bf16x4);
        const int c4 = f & 7;  // float4 index inside the 32-float row: 4 bf16 = half a 16-byte chunk
        const int off = swz(f >> 3, c4 >> 1) + (c4 & 1) * 8;
        *reinterpret_cast<bf16x4 *>(base + off) = h;
        *reinterpret_cast<bf16x4 *>(base + kImgE + off) = l;
    };
    auto store_query = [&](int i, int buf) {
        unsigned char *base = lds + buf * kSplitBuf;
        const int f = tid + i * kSplitThreads;
        const int off = swz(f >> 2, f & 3);
        *reinterpret_cast<bf16x8 *>(base + 2 * kImgE + off) = st_qh[i];
        *reinterpret_cast<bf16x8 *>(base + 2 * kImgE + kImgQ + off) = st_ql[i];
    };
    auto load_stage = [&]() {
#pragma unroll
        for (int i = 0; i < 4; ++i) load_corpus(i);
#pragma unroll
        for (int i = 0; i < 2; ++i) load_query(i);
        advance_cursor();
    };
    auto store_stage = [&](int buf) {
#pragma unroll
        for (int i = 0; i < 4; ++i) store_corpus(i, buf);
#pragma unroll
        for (int i = 0; i < 2; ++i) store_query(i, buf);
    };

    f32x16 acc[4][2];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    // Staging is dealt out between the MFMAs of a step, one item per slot (a slot = the three MFMAs of one
    // accumulator tile): the registers holding step it+1 are written to the other LDS buffer in the first six slots,
    // the loads of step it+2 follow at once.  A wave issues in order and the waves of the workgroup leave every
    // barrier together, so staging as a block at the top of the step idled the matrix pipe (dense_batched.hip has
    // the measurements).  Stores and loads also run on the last steps: the cursor clamps to the last row and
    // nobody reads the buffer, and a branch around them would force vmcnt(0) waits.
    if (total > 0) {
        load_stage();
        store_stage(0);
        load_stage();
    }
    __syncthreads();
    int64_t cur_tile = first_tile;
    int cur_ks = 0;
    for (int64_t it = 0; it < total; ++it) {
        const int buf = (int)(it & 1);
        const unsigned char *base = lds + buf * kSplitBuf;
#pragma unroll
        for (int s = 0; s < kBK / 16; ++s) {  // two k = 16 sub-steps per staged tile
            bf16x8 ah[4], al[4], bh[2], bl[2];
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const int off = swz(rw * 128 + t * 32 + l31, 2 * s + lh);
                ah[t] = *reinterpret_cast<const bf16x8 *>(base + off);
                al[t] = *reinterpret_cast<const bf16x8 *>(base + kImgE + off);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int off = swz(qw * 64 + t * 32 + l31, 2 * s + lh);
                bh[t] = *reinterpret_cast<const bf16x8 *>(base + 2 * kImgE + off);
                bl[t] = *reinterpret_cast<const bf16x8 *>(base + 2 * kImgE + kImgQ + off);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    // small cross terms first, the dominant hi.hi product last
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[ti], bh[tj], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ti], bl[tj], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[ti], bh[tj], acc[ti][tj], 0, 0, 0);
                    const int slot = s * 8 + ti * 2 + tj;  // 0..15
                    if (slot < 10) {
                        __builtin_amdgcn_sched_barrier(0);
                        if (slot < 4) store_corpus(slot, buf ^ 1);
                        else if (slot < 6) store_query(slot - 4, buf ^ 1);
                        else if (slot == 6) { load_corpus(0); load_corpus(1); }
                        else if (slot == 7) { load_corpus(2); load_corpus(3); }
                        else if (slot == 8) load_query(0);
                        else { load_query(1); advance_cursor(); }
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
        if (++cur_ks == ksteps) {
            int no_fill = 0;
            batched_tile_epilogue<SAMPLE, FILTER, 4>(acc, cur_tile, rw, qw, 0, l31, lh, my_tau, n_work, stride, nq,
                                                     sample_scores, cnt, cand, cap, src, allow_bits, nullptr, no_fill);
            cur_ks = 0;
            cur_tile += tile_step;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ the full pass: LDS-DMA staging, ping-pong waves
// The full pass streams the whole corpus once, so it gets its own kernel (the sampled pass above reads a few per cent
// of the rows, strided, and stays on the register-staged kernel).  What is different:
//
//  * The corpus is kept a second time as READY-TO-COPY IMAGES (built on the first split-precision pass after a load,
//    +4 bytes per element of HBM): for every tile of 256 rows and every k-step of 16, an 8 KB hi image and an 8 KB lo
//    image in exactly the (swizzled) byte order the fragment reads want.  A stage is then 16 KB of corpus + 16 KB of
//    query images copied HBM/L2 -> LDS by global_load_lds (LDS-DMA): no staging registers, no VALU split, no ds_write.
//  * Four 32 KB stages in LDS, three in flight.
//  * PING-PONG: waves 0-3 (one per SIMD) and waves 4-7 run half a step apart -- while one set issues its 24 MFMAs
//    the other reads its fragments for the next step (and pushes its share of the copies), then they swap at a
//    barrier.  The matrix pipe always has one wave feeding it and no wave needs a second fragment set (128
//    accumulator + 48 fragment registers).  scripts/exp/k2dma_bench.hip holds the stand-alone measurements: without
//    copies this loop is within 4 % of an MFMA-only loop; with them the kernel sits on the board's POWER limit (the
//    shader clock falls to ~1.6 GHz: every schedule variant lands on the same 0.96 ms per 1M x 768 pass).
//  * The tile epilogue looks at 16 scores with 16 compares and takes the per-score path only for an accumulator
//    tile that holds a survivor (or a NaN); survivors go through a per-wave LDS slice (dense_batched_common.hpp).
constexpr int kDK = 16;                         // k per stage
constexpr int kDImg = kSM * kDK * 2;            // one image: 256 rows x 16 bf16 = 8 KB
constexpr int kDStage = 4 * kDImg;              // corpus hi, corpus lo, query hi, query lo
constexpr int kDSlots = 4;
constexpr int kDmaLdsBytes = kDSlots * kDStage;  // 128 KB

// byte offset of 16-byte half `h` (k 0..7 / 8..15) of image row `row`: the two halves swap every 8 rows, so that the
// 64 lanes of a ds_read_b128 (lane l -> row l & 31, half l >> 5) cover all banks
__device__ __forceinline__ int dswz(int row, int h) { return row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }

// one thread per (row of a tile, 8-element chunk): 32 bytes of f32 in, 16 bytes into the hi and the lo image
__global__ void split_images_kernel(const float *__restrict__ x, int64_t n_rows, int32_t dim,
                                    unsigned char *__restrict__ img) {
    const int chunks = dim / 8;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t row = i / chunks;
    const int c = (int)(i - row * chunks);
    const int64_t n_tiles = (n_rows + kSM - 1) / kSM;
    if (row >= n_tiles * kSM) return;
    bf16x8 h, l;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = row < n_rows ? x[row * dim + c * 8 + j] : 0.f;
        const __bf16 hv = (__bf16)v;
        h[j] = hv;
        l[j] = (__bf16)(v - (float)hv);
    }
    const int64_t tile = row / kSM;
    const int r = (int)(row - tile * kSM);
    unsigned char *blk = img + (tile * (dim / kDK) + (c >> 1)) * (int64_t)(2 * kDImg);
    *reinterpret_cast<bf16x8 *>(blk + dswz(r, c & 1)) = h;
    *reinterpret_cast<bf16x8 *>(blk + kDImg + dswz(r, c & 1)) = l;
}

constexpr int kSurvEntries = 448;               // survivors per wave slice: 448 x 9 bytes = 4,032
constexpr int kSurvSlice = 4096;
constexpr int kDmaLdsTotal = kDmaLdsBytes + 8 * kSurvSlice;  // 160 KB: all of a CU's LDS

// TJ = accumulator tiles per wave along the queries: 2 = all 256 queries of the block, 1 = the first 128 (a pass of
// <= 128 queries then does half the matrix work and copies half the query images instead of multiplying padding).
template <bool FILTER, int TJ>
__global__ __launch_bounds__(kSplitThreads, 2) void dense_batched_split_dma_kernel(
    const unsigned char *__restrict__ e_img, const unsigned char *__restrict__ q_img, int32_t ksteps, int32_t nq,
    int64_t n_work, const float *__restrict__ tau, int32_t *__restrict__ cnt, Cand32 *__restrict__ cand, int32_t cap,
    const uint16_t *__restrict__ src, const uint32_t *__restrict__ allow_bits) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5;
    const int rw = wave & 1, qw = wave >> 1;
    const int64_t n_tiles = (n_work + kSM - 1) / kSM;
    const int64_t first_tile = blockIdx.x, tile_step = gridDim.x;
    const int64_t my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + tile_step - 1) / tile_step : 0;
    const int64_t total = my_tiles * ksteps;
    if (total == 0) return;

    float my_tau[TJ];  // padding queries (zero rows of the block) never keep a score
#pragma unroll
    for (int tj = 0; tj < TJ; ++tj) {
        const int q = qw * (TJ * 32) + tj * 32 + l31;
        my_tau[tj] = q < nq ? tau[q] : 3.0e38f;
    }
    int surv_fill = 0;  // wave-uniform: entries in this wave's survivor slice
    unsigned char *slice = lds + kDmaLdsBytes + wave * kSurvSlice;

    // copies: waves 0-3 bring the 16 KB of corpus images of a stage (4 KB = four 1 KB instructions each), waves 4-7
    // the 16 KB of query images (the same 48 x 16 KB for every tile: they come from L2).  Past the last stage the
    // cursor stays on the last tile: the copies land in a slot nobody reads, and a branch around them would break the
    // counted vmcnt waits.
    int64_t ld_tile = first_tile;
    int ld_ks = 0;
    // per stage a corpus wave issues 4 instructions of 1 KB; a query wave 4 (TJ = 2: a quarter of the 16 KB of query
    // images) or 2 (TJ = 1: rows 0..127 are the first half of each image; waves 4, 5 take the hi image's, 6, 7 the lo's)
    constexpr int QI = 2 * TJ;
    auto issue_stage = [&](int slot) {
        const int y = wave - 4;
        const int q_off = TJ == 2 ? y * 4096 : (y >> 1) * kDImg + (y & 1) * 2048;
        const unsigned char *g = wave < 4 ? e_img + (ld_tile * ksteps + ld_ks) * (int64_t)(2 * kDImg) + wave * 4096
                                          : q_img + (int64_t)ld_ks * (2 * kDImg) + q_off;
        unsigned char *d = lds + slot * kDStage + (wave < 4 ? wave * 4096 : 2 * kDImg + q_off);
        if (wave < 4) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16,
                                                 (__attribute__((address_space(3))) void *)(d + i * 1024), 16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < QI; ++i)
                __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16,
                                                 (__attribute__((address_space(3))) void *)(d + i * 1024), 16, 0, 0);
        }
        if (++ld_ks == ksteps) {
            ld_ks = 0;
            if (ld_tile + tile_step < n_tiles) ld_tile += tile_step;
        }
    };

    f32x16 acc[4][TJ];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;

    struct Frags {
        bf16x8 ah[4], al[4], bh[TJ], bl[TJ];
    };
    auto read_frags = [&](int slot, Frags &f) {
        const unsigned char *base = lds + slot * kDStage;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = dswz(rw * 128 + t * 32 + l31, lh);
            f.ah[t] = *reinterpret_cast<const bf16x8 *>(base + off);
            f.al[t] = *reinterpret_cast<const bf16x8 *>(base + kDImg + off);
        }
#pragma unroll
        for (int t = 0; t < TJ; ++t) {
            const int off = dswz(qw * (TJ * 32) + t * 32 + l31, lh);
            f.bh[t] = *reinterpret_cast<const bf16x8 *>(base + 2 * kDImg + off);
            f.bl[t] = *reinterpret_cast<const bf16x8 *>(base + 3 * kDImg + off);
        }
    };
    // kind-major: the three products of one accumulator are eight MFMAs apart; small cross terms first
    auto mfmas = [&](const Frags &f) {
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj)
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
    };
    int64_t cur_tile = first_tile;
    int cur_ks = 0;
    auto step_done = [&]() {
        if (++cur_ks < ksteps) return;
        cur_ks = 0;
        batched_tile_epilogue<false, FILTER, 4, kSurvEntries, TJ>(acc, cur_tile, rw, qw, 0, l31, lh, my_tau, n_work, 1, nq,
                                                              nullptr, cnt, cand, cap, src, allow_bits, slice, surv_fill);
        cur_tile += tile_step;
    };

    // The compiler moves MFMAs (register-only) across an s_barrier freely: fence the scheduler around each one.
#define ANRAG_K2_BAR()                          \
    do {                                        \
        __builtin_amdgcn_sched_barrier(0);      \
        __builtin_amdgcn_s_barrier();           \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
    issue_stage(0);
    issue_stage(1);
    issue_stage(2);
    if (wave < 4 || TJ == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    ANRAG_K2_BAR();
    // Stage c lives in slot c & 3.  Waves 0-3 read it in phase 2c and multiply in phase 2c+1; waves 4-7 read it in
    // phase 2c+1 and multiply in phase 2c+2.  Slot (c+3) & 3 = slot of stage c-1 is free from phase 2c on (its last
    // readers, waves 4-7, finished in phase 2c-1).  Every wave waits for its own copies of stage c+1 (vmcnt(8): two
    // later stages may be in flight) before the barrier that ends phase 2c+1.
    Frags f;
    if (wave < 4) {
        for (int64_t c = 0; c < total; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            read_frags((int)(c & 3), f);
            issue_stage((int)((c + 3) & 3));
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            ANRAG_K2_BAR();
            mfmas(f);
            step_done();
            asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
            ANRAG_K2_BAR();
        }
    } else {
        for (int64_t c = 0; c < total; ++c) {
            __builtin_amdgcn_sched_barrier(0);
            if (c) {
                mfmas(f);
                step_done();
            }
            ANRAG_K2_BAR();
            read_frags((int)(c & 3), f);
            issue_stage((int)((c + 3) & 3));
            if constexpr (TJ == 2) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
            ANRAG_K2_BAR();
        }
        mfmas(f);
        step_done();
    }
#undef ANRAG_K2_BAR
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the trailing copies must not outlive the workgroup's LDS
    if (surv_fill) flush_survivors(slice, kSurvEntries, surv_fill, cnt, cand);
}

__global__ void batched_threshold_kernel(const float *, int64_t, int32_t, float *, int32_t *);  // dense_batched.hip

int batched_passes_split(anrag_index *idx, hipStream_t st, int32_t nq, int32_t k, int64_t n_sample, int64_t stride,
                         const uint32_t *allow, Cand32 *cand) {
    // > 64 KB of dynamic LDS: asked for per device, under a mutex (common.hpp: ensure_dynamic_lds)
    for (const void *f : {reinterpret_cast<const void *>(&dense_batched_split_kernel<true, false>),
                          reinterpret_cast<const void *>(&dense_batched_split_kernel<true, true>)})
        if (int rc = ensure_dynamic_lds(idx->device, f, kSplitLdsBytes)) return rc;
    for (const void *f : {reinterpret_cast<const void *>(&dense_batched_split_dma_kernel<false, 2>),
                          reinterpret_cast<const void *>(&dense_batched_split_dma_kernel<true, 2>),
                          reinterpret_cast<const void *>(&dense_batched_split_dma_kernel<false, 1>),
                          reinterpret_cast<const void *>(&dense_batched_split_dma_kernel<true, 1>)})
        if (int rc = ensure_dynamic_lds(idx->device, f, kDmaLdsTotal)) return rc;
    const int64_t n = idx->n_rows;
    const int dim = idx->dim;
    const int64_t qelems = (int64_t)kBQ * dim;
    const int64_t n_tiles = (n + kSM - 1) / kSM;
    if (!idx->d_split_img) {  // first split-precision pass after a load: the corpus images (free_batched drops them)
        const int64_t bytes = n_tiles * kSM * (int64_t)dim * 4;
        ANRAG_HIP(counted_malloc(&idx->d_split_img, (size_t)bytes));
        idx->split_img_bytes = bytes;
        idx->hbm_bytes += bytes;
        const int64_t items = n_tiles * kSM * (dim / 8);
        split_images_kernel<<<(unsigned)((items + 255) / 256), 256, 0, st>>>(
            idx->d_emb, n, dim, static_cast<unsigned char *>(idx->d_split_img));
    }
    if (!idx->d_bq_img) ANRAG_HIP(counted_malloc(&idx->d_bq_img, (size_t)qelems * 4));
    split_images_kernel<<<(unsigned)((kBQ * (dim / 8) + 255) / 256), 256, 0, st>>>(
        idx->d_bq, kBQ, dim, static_cast<unsigned char *>(idx->d_bq_img));
    if (!idx->d_bq_hi) {
        ANRAG_HIP(counted_malloc(&idx->d_bq_hi, (size_t)qelems * 2));
        ANRAG_HIP(counted_malloc(&idx->d_bq_lo, (size_t)qelems * 2));
    }
    __bf16 *qh = static_cast<__bf16 *>(idx->d_bq_hi), *ql = static_cast<__bf16 *>(idx->d_bq_lo);
    split_queries_kernel<<<(unsigned)((qelems + 255) / 256), 256, 0, st>>>(idx->d_bq, qelems, qh, ql);
    auto grid_for = [&](int64_t rows) {
        const int64_t tiles = (rows + kSM - 1) / kSM;
        return (unsigned)(tiles < idx->n_cus ? tiles : idx->n_cus);
    };
    if (allow)
        dense_batched_split_kernel<true, true><<<grid_for(n_sample), kSplitThreads, kSplitLdsBytes, st>>>(
            idx->d_emb, qh, ql, dim, nq, n_sample, stride, nullptr, idx->d_bsample, nullptr, nullptr, 0, idx->d_dense_src,
            allow);
    else
        dense_batched_split_kernel<true, false><<<grid_for(n_sample), kSplitThreads, kSplitLdsBytes, st>>>(
            idx->d_emb, qh, ql, dim, nq, n_sample, stride, nullptr, idx->d_bsample, nullptr, nullptr, 0, nullptr, nullptr);
    batched_threshold_kernel<<<nq, kThrWaves * 64, 0, st>>>(idx->d_bsample, sample_floats_per_query(n_sample, kSM), k,
                                                             idx->d_btau, idx->d_bcnt);
    const unsigned char *e_img = static_cast<const unsigned char *>(idx->d_split_img);
    const unsigned char *q_img = static_cast<const unsigned char *>(idx->d_bq_img);
#define ANRAG_SPLIT_FULL(F, TJ_)                                                                                      \
    dense_batched_split_dma_kernel<F, TJ_><<<grid_for(n), kSplitThreads, kDmaLdsTotal, st>>>(                          \
        e_img, q_img, dim / kDK, nq, n, idx->d_btau, idx->d_bcnt, cand, kCandCap, F ? idx->d_dense_src : nullptr,      \
        F ? allow : nullptr)
    if (nq <= 128) {
        if (allow) ANRAG_SPLIT_FULL(true, 1); else ANRAG_SPLIT_FULL(false, 1);
    } else {
        if (allow) ANRAG_SPLIT_FULL(true, 2); else ANRAG_SPLIT_FULL(false, 2);
    }
#undef ANRAG_SPLIT_FULL
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
