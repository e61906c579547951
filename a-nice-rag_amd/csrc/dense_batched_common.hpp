// Shared by the two K2 kernels (dense_batched.hip: exact f32 MFMA; dense_batched_split.hip: bf16 x 3 split).
#pragma once
#include "common.hpp"
#include "wave_topk.hpp"

namespace anrag {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kBM = 128;  // corpus rows per workgroup tile
constexpr int kBQ = 256;  // queries per pass
constexpr int kBK = 32;   // k per staging step
constexpr int32_t kCandCap = 8192;
constexpr int kThrWaves = 8;  // batched_threshold_kernel: waves per query

struct Cand32 {
    float score;
    uint32_t row;
};

// Survivors of the full pass (score >= the query's threshold) are rare -- a few per wave and tile -- so a wave parks
// them in its own slice of LDS (no atomics: the wave is the only writer, positions come from a ballot) and sends them
// to the per-query candidate lists in HBM in one go, when the slice fills up and at the end of the kernel: the global
// atomics of a whole slice are in flight together instead of one round trip per score in the middle of the tile loop
// (which cost 6 % of the f32 pass and 30 % of the split-precision one).  A slice of `entries` survivors takes
// entries * 9 bytes: scores, rows, query numbers (< 256).
struct SurvBuf {
    float *score;
    uint32_t *row;
    uint8_t *q;
};
__device__ __forceinline__ SurvBuf surv_buf(unsigned char *slice, int entries) {
    return {reinterpret_cast<float *>(slice), reinterpret_cast<uint32_t *>(slice + entries * 4), slice + entries * 8};
}

// (Few, narrow arguments on purpose, here and below: one argument more than fit the argument registers went through
// the stack, which turns on scratch memory for the whole kernel -- the f32 full pass ran 20 % slower for it.)
// flush_survivors_body is inlined into tile_survivors, so that function stays a LEAF: a non-inlined function that
// calls another one has to save its return address on the stack, and that alone gave every full-pass kernel a 16-byte
// private segment (scratch set up for the whole launch).  tests/test_build_resources.py holds all K2 kernels to 0.
static __device__ __forceinline__ void flush_survivors_body(unsigned char *slice, int entries, int n,
                                                            int32_t *__restrict__ cnt, Cand32 *__restrict__ cand) {
    constexpr int32_t cap = kCandCap;
    const SurvBuf b = surv_buf(slice, entries);
    for (int i = lane_id(); i < n; i += kWave) {
        const int q = b.q[i];
        const int pos = atomicAdd(&cnt[q], 1);
        if ((uint32_t)pos < (uint32_t)cap) {
            Cand32 c;
            c.score = b.score[i];
            c.row = b.row[i];
            cand[(int64_t)q * cap + pos] = c;
        }
    }
    // atomics and stores count in vmcnt like the LDS-DMA copies of the split-precision kernel: drain them, so that its
    // counted waits keep meaning "my copies of the stage two back have landed"
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
}
static __device__ __noinline__ void flush_survivors(unsigned char *slice, int entries, int n, int32_t *__restrict__ cnt,
                                                    Cand32 *__restrict__ cand) {
    flush_survivors_body(slice, entries, n, cnt, cand);
}

// The per-score path for one 32 x 32 accumulator tile that holds at least one survivor (lane = query column q -- -1
// for a padding query: a NaN / inf row scores NaN against the zero rows of the query block too -- with threshold
// tau_q; register r = row row0 + (r&3) + 8*(r>>2), row0 includes 4 * lane half).  Returns the wave's new fill.
template <bool FILTER, int ENTRIES>
__device__ __noinline__ int tile_survivors(f32x16 v, uint32_t row0, int q, float tau_q, uint32_t n_work,
                                           unsigned char *slice, int fill, int32_t *__restrict__ cnt,
                                           Cand32 *__restrict__ cand, const uint16_t *__restrict__ src,
                                           const uint32_t *__restrict__ allow_bits) {
    const SurvBuf b = surv_buf(slice, ENTRIES);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        float s = v[r];
        if (__builtin_amdgcn_ballot_w64(!(s < tau_q)) == 0) continue;  // !(s < tau): survivors and NaNs
        const uint32_t wr = row0 + (r & 3) + 8 * (r >> 2);
        bool ok = !(s < tau_q) && wr < n_work && q >= 0;
        if constexpr (FILTER) {
            if (ok) ok = source_ok(allow_bits, src[wr]);
        }
        if (s != s) s = __builtin_huge_valf();  // NaN ranks first, as in K1 (and in numpy)
        const unsigned long long m = __builtin_amdgcn_ballot_w64(ok);
        if (ok) {
            const int at = fill + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0));
            b.score[at] = s;
            b.row[at] = wr;
            b.q[at] = (uint8_t)q;
        }
        fill += __builtin_popcountll(m);
        if (fill > ENTRIES - kWave) {
            flush_survivors_body(slice, ENTRIES, fill, cnt, cand);
            fill = 0;
        }
    }
    return fill;
}

// Epilogue of one (64*TI) x QW tile held as TI x 2 MFMA accumulators per wave: D[corpus row][query], lane = query
// column (a wave owns TJ x 32 of them), 16 rows per register set (32x32 C/D map: row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  SAMPLE: store each lane's
// two best scores per query column; otherwise 16 compares per accumulator tile, and the per-score path above only for a tile
// that holds a survivor (my_tau must be +huge for padding queries).  Zeroes the accumulators.
template <bool SAMPLE, bool FILTER, int TI = 2, int ENTRIES = 448, int TJ = 2>
__device__ __forceinline__ void batched_tile_epilogue(f32x16 (&acc)[TI][TJ], int64_t tile, int rw, int qw, int qbase,
                                                      int l31, int lh, const float (&my_tau)[TJ], int64_t n_work,
                                                      int64_t stride, int32_t nq, float *__restrict__ sample_scores,
                                                      int32_t *__restrict__ cnt, Cand32 *__restrict__ cand, int32_t cap,
                                                      const uint16_t *__restrict__ src,
                                                      const uint32_t *__restrict__ allow_bits, unsigned char *slice,
                                                      int &fill) {
    // a workgroup tile is 2 (row halves) x TI x 32 corpus rows: TI = 2 -> 128 rows (kBM), TI = 4 -> 256 rows
    if constexpr (SAMPLE) {
        // A lane holds TI x 16 sampled rows of each of its two query columns: it keeps the BEST TWO of them and writes
        // those -- sample_scores[query][slot][2], slot = (tile, row half, lane half).  The threshold is the k-th best
        // of a query's slots: still the score of a real row with k - 1 real rows ahead of it, i.e. a lower bound on the
        // k-th best of the corpus; and hardly lower than the k-th best of ALL sampled rows (it differs only when three
        // of the k best sampled rows share one lane's 16 TI rows).  Writing every score (16.7M four-byte stores scattered
        // over 256 query rows, then read back by the threshold kernel) was the larger part of the pre-pass.
        const int64_t n_tiles = (n_work + 2 * TI * 32 - 1) / (2 * TI * 32);
        // which of the lane's TI x 16 rows count (inside the sample, allowed source): ONE bit per row, worked out before
        // the query columns are walked -- a row's source does not depend on the column.  (Asking per column doubled the
        // source loads, and the filtered sampled pass of the split kernel spilled 118 registers over them; ti is NOT
        // unrolled here so that at most 16 source loads are in flight -- acc is not touched in this loop.)
        static_assert(TI * 16 <= 64, "one bit per row of the lane");
        unsigned long long okbits = 0;
#pragma unroll 1
        for (int ti = 0; ti < TI; ++ti) {
            uint32_t bits = 0;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t wr = tile * (2 * TI * 32) + rw * (TI * 32) + ti * 32 + 4 * lh + (r & 3) + 8 * (r >> 2);
                bool ok = wr < n_work;
                if constexpr (FILTER) {
                    if (ok) ok = source_ok(allow_bits, src[wr * stride]);
                }
                bits |= (ok ? 1u : 0u) << r;
            }
            okbits |= (unsigned long long)bits << (ti * 16);
        }
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj) {
            float b1 = neg_inf<float>(), b2 = neg_inf<float>();
#pragma unroll
            for (int ti = 0; ti < TI; ++ti)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    float s = acc[ti][tj][r];
                    if (s != s) s = __builtin_huge_valf();  // NaN ranks first, as in K1 (and in numpy)
                    const bool ok = (okbits >> (ti * 16 + r)) & 1ull;
                    if (!ok) s = neg_inf<float>();
                    const float lo = s < b1 ? s : b1;  // the smaller of (s, b1) competes for second place
                    b1 = s < b1 ? b1 : s;
                    b2 = lo < b2 ? b2 : lo;
                }
            const int q = qbase + qw * (TJ * 32) + tj * 32 + l31;
            const int64_t slot = (tile * 2 + rw) * 2 + lh;
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            *reinterpret_cast<f32x2 *>(sample_scores + ((int64_t)q * n_tiles * 4 + slot) * 2) = f32x2{b1, b2};
        }
    } else {
#pragma unroll
        for (int ti = 0; ti < TI; ++ti)
#pragma unroll
            for (int tj = 0; tj < TJ; ++tj) {
                const int q = qbase + qw * (TJ * 32) + tj * 32 + l31;
                const int64_t row0 = tile * (2 * TI * 32) + rw * (TI * 32) + ti * 32 + 4 * lh;
                bool hit = false;
#pragma unroll
                for (int r = 0; r < 16; ++r) hit |= !(acc[ti][tj][r] < my_tau[tj]);
                if (__builtin_amdgcn_ballot_w64(hit) != 0)
                    fill = tile_survivors<FILTER, ENTRIES>(acc[ti][tj], (uint32_t)row0, q < nq ? q : -1, my_tau[tj],
                                                           (uint32_t)n_work, slice, fill, cnt, cand, src, allow_bits);
            }
    }
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < TJ; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
}

// floats per query the sampled pass of tile height `tile_rows` writes for n_sample rows (two per slot, above)
inline int64_t sample_floats_per_query(int64_t n_sample, int tile_rows) {
    return (n_sample + tile_rows - 1) / tile_rows * 8;
}

// dense_batched_split.hip
int batched_passes_split(anrag_index *idx, hipStream_t st, int32_t nq, int32_t k, int64_t n_sample, int64_t stride,
                         const uint32_t *allow, Cand32 *cand);

}  // namespace anrag
