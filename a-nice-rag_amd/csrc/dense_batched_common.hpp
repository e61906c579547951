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

struct Cand32 {
    float score;
    uint32_t row;
};

// Epilogue of one (64*TI) x QW tile held as TI x 2 MFMA accumulators per wave: D[corpus row][query], lane = query
// column, 16 rows per register set (32x32 C/D map: row = (r&3) + 8*(r>>2) + 4*(lane>>5)).  SAMPLE: store every
// score of the sampled rows; otherwise append survivors (score >= this query's threshold) to its candidate list.
// Zeroes the accumulators.
template <bool SAMPLE, bool FILTER, int TI = 2>
__device__ __forceinline__ void batched_tile_epilogue(f32x16 (&acc)[TI][2], int64_t tile, int rw, int qw, int qbase,
                                                      int l31, int lh, const float (&my_tau)[2], int64_t n_work,
                                                      int64_t stride, int32_t nq, float *__restrict__ sample_scores,
                                                      int32_t *__restrict__ cnt, Cand32 *__restrict__ cand, int32_t cap,
                                                      const uint16_t *__restrict__ src,
                                                      const uint32_t *__restrict__ allow_bits) {
    // a workgroup tile is 2 (row halves) x TI x 32 corpus rows: TI = 2 -> 128 rows (kBM), TI = 4 -> 256 rows
#pragma unroll
    for (int ti = 0; ti < TI; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) {
            const int q = qbase + qw * 64 + tj * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int64_t wr = tile * (2 * TI * 32) + rw * (TI * 32) + ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                float s = acc[ti][tj][r];
                if (s != s) s = __builtin_huge_valf();  // NaN ranks first, as in K1 (and in numpy)
                if constexpr (SAMPLE) {
                    if (wr < n_work) {
                        bool ok = true;
                        if constexpr (FILTER) ok = source_ok(allow_bits, src[wr * stride]);
                        sample_scores[(int64_t)q * n_work + wr] = ok ? s : neg_inf<float>();
                    }
                } else {
                    if (s >= my_tau[tj] && wr < n_work && q < nq) {
                        bool ok = true;
                        if constexpr (FILTER) ok = source_ok(allow_bits, src[wr]);
                        if (ok) {
                            const int pos = atomicAdd(&cnt[q], 1);
                            if (pos < cap) {
                                Cand32 c;
                                c.score = s;
                                c.row = (uint32_t)wr;
                                cand[(int64_t)q * cap + pos] = c;
                            }
                        }
                    }
                }
                acc[ti][tj][r] = 0.f;
            }
        }
}

// dense_batched_split.hip
int batched_passes_split(anrag_index *idx, hipStream_t st, int32_t nq, int32_t k, int64_t n_sample, int64_t stride,
                         const uint32_t *allow, Cand32 *cand);

}  // namespace anrag
