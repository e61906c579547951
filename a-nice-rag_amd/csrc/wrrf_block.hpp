// Weighted reciprocal-rank fusion of <= blockDim.x entries held in LDS, by one workgroup (device function
// shared by wrrf.hip's small kernel and the query tail kernels).  Semantics and exactness: wrrf.hip header.
#pragma once
#include "common.hpp"

namespace anrag {

__device__ __forceinline__ bool wrrf_outranks(double sj, int32_t j, double si, int32_t i) {
    return sj > si || (sj == si && j < i);
}

// s_id[i] / s_c[i]: doc id (< 0 = padding) and contribution w_l * (1 / (k + rank)) of entry i, i < m <= blockDim.x,
// already written and followed by a barrier.  s_score / s_owner / s_distinct: scratch.  Writes min(top_n, distinct)
// records to `out` and that number to *out_count.  All threads of the workgroup must call.
__device__ __forceinline__ void wrrf_in_block(const int64_t *s_id, const double *s_c, double *s_score,
                                              int32_t *s_owner, int32_t *s_distinct, int32_t m, int32_t top_n,
                                              anrag_candidate *__restrict__ out, int32_t *__restrict__ out_count) {
    const int i = threadIdx.x;
    if (i == 0) *s_distinct = 0;
    __syncthreads();
    bool owner = false;
    double score = 0.0;
    int64_t id = -1;
    if (i < m) {
        id = s_id[i];
        if (id >= 0) {
            owner = true;
            for (int j = 0; j < i; ++j)
                if (s_id[j] == id) {
                    owner = false;  // an earlier entry owns this id
                    break;
                }
            if (owner)
                for (int j = i; j < m; ++j)
                    if (s_id[j] == id) score = score + s_c[j];  // the dict update's additions, in its order
        }
        s_score[i] = score;
        s_owner[i] = owner ? 1 : 0;
        if (owner) atomicAdd(s_distinct, 1);
    }
    __syncthreads();
    if (owner) {
        int32_t pos = 0;
        for (int j = 0; j < m; ++j)
            if (s_owner[j] && wrrf_outranks(s_score[j], j, score, i)) ++pos;
        if (pos < top_n) {
            anrag_candidate r;
            r.score = score;
            r.doc = id;
            out[pos] = r;
        }
    }
    if (i == 0 && out_count) *out_count = *s_distinct < top_n ? *s_distinct : top_n;
}

}  // namespace anrag
