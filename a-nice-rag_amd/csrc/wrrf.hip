// K5 -- weighted reciprocal-rank fusion + top-n, on document ids.
//
// Replaces SearchEngine.weighted_reciprocal_rank_fusion (src/search_engine.py:21-34) and the caller's
// truncation to common_sections_n (src/query_rag_retrieval.py:360-362).  Reference semantics, kept
// bit-for-bit in fp64 (library built with -ffp-contract=off, fp64 division correctly rounded):
//     for each list l in order, for rank = 1.. :  score[id] = score[id] + w_l * (1 / (k + rank))
//     result = stable sort by score descending  (equal scores keep first-insertion order)
// The device form is all-pairs, not a hash map: entry i finds the first entry with its id (that entry
// owns the id) and the owner re-adds every matching contribution in entry order -- the same additions
// in the same order as the dict update -- then an owner's output position is the number of owners that
// outrank it.  Latency-bound; M = sum of list lengths is 2*k <= 128 on the hybrid hot path (one
// workgroup, everything in LDS) and a few 10^4 in retrieval_eval's full-ranking mode (grid form).
#include "common.hpp"
#include "wrrf_block.hpp"

namespace anrag {

constexpr int kWrrfSmall = 1024;
constexpr int kWrrfAllPairsMax = 4096;  // above: sort-based form (sort_select.hip)

struct WrrfLists {
    int32_t n;
    int32_t off[ANRAG_WRRF_MAX_LISTS + 1];
    double w[ANRAG_WRRF_MAX_LISTS];
};

__device__ __forceinline__ void wrrf_entry(const int64_t *ids, const anrag_candidate *cands, const WrrfLists &L,
                                           int32_t i, double k, int64_t &id, double &contrib) {
    id = cands ? cands[i].doc : ids[i];
    int l = 0;
    while (l + 1 < L.n && i >= L.off[l + 1]) ++l;
    const int32_t rank = i - L.off[l] + 1;
    contrib = id < 0 ? 0.0 : L.w[l] * (1.0 / (k + (double)rank));
}

__device__ __forceinline__ bool outranks(double sj, int32_t j, double si, int32_t i) {
    return wrrf_outranks(sj, j, si, i);
}

// M <= 1024: one workgroup, one thread per entry.
__global__ __launch_bounds__(kWrrfSmall) void wrrf_small_kernel(const int64_t *__restrict__ ids,
                                                                const anrag_candidate *__restrict__ cands,
                                                                WrrfLists L, int32_t m, double k, int32_t top_n,
                                                                anrag_candidate *__restrict__ out,
                                                                int32_t *__restrict__ out_count) {
    __shared__ int64_t s_id[kWrrfSmall];
    __shared__ double s_c[kWrrfSmall];
    __shared__ double s_score[kWrrfSmall];
    __shared__ int32_t s_owner[kWrrfSmall];
    __shared__ int32_t s_distinct;
    const int i = threadIdx.x;
    int64_t id = -1;
    double c = 0.0;
    if (i < m) wrrf_entry(ids, cands, L, i, k, id, c);
    s_id[i] = id;
    s_c[i] = c;
    __syncthreads();
    wrrf_in_block(s_id, s_c, s_score, s_owner, &s_distinct, m, top_n, out, out_count);
}

// ---- grid form for long lists (full-ranking evaluation mode)
__global__ void wrrf_contrib_kernel(const int64_t *__restrict__ ids, const anrag_candidate *__restrict__ cands,
                                    WrrfLists L, int32_t m, double k, int64_t *__restrict__ w_ids,
                                    double *__restrict__ w_contrib, int32_t *__restrict__ count) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i == 0) *count = 0;
    if (i >= m) return;
    int64_t id;
    double c;
    wrrf_entry(ids, cands, L, i, k, id, c);
    w_ids[i] = id;
    w_contrib[i] = c;
}

__global__ __launch_bounds__(256) void wrrf_sum_kernel(const int64_t *__restrict__ w_ids,
                                                       const double *__restrict__ w_contrib, int32_t m,
                                                       double *__restrict__ w_score, int32_t *__restrict__ w_owner,
                                                       int32_t *__restrict__ count) {
    __shared__ int64_t t_id[1024];
    __shared__ double t_c[1024];
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const int64_t id = i < m ? w_ids[i] : -1;
    bool owner = id >= 0;
    double score = 0.0;
    for (int32_t j0 = 0; j0 < m; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            t_id[t] = j0 + t < m ? w_ids[j0 + t] : -2;
            t_c[t] = j0 + t < m ? w_contrib[j0 + t] : 0.0;
        }
        __syncthreads();
        if (owner) {
            const int32_t n = m - j0 < 1024 ? m - j0 : 1024;
            for (int32_t t = 0; t < n; ++t)
                if (t_id[t] == id) {
                    if (j0 + t < i) {
                        owner = false;  // an earlier entry owns this id
                        break;
                    }
                    score = score + t_c[t];
                }
        }
    }
    if (i < m) {
        w_score[i] = score;
        w_owner[i] = owner ? 1 : 0;
        if (owner) atomicAdd(count, 1);
    }
}

__global__ __launch_bounds__(256) void wrrf_rank_kernel(const int64_t *__restrict__ w_ids,
                                                        const double *__restrict__ w_score,
                                                        const int32_t *__restrict__ w_owner, int32_t m, int32_t top_n,
                                                        anrag_candidate *__restrict__ out) {
    __shared__ double t_s[1024];
    __shared__ int32_t t_o[1024];
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    const bool owner = i < m && w_owner[i];
    const double score = owner ? w_score[i] : 0.0;
    int32_t pos = 0;
    for (int32_t j0 = 0; j0 < m; j0 += 1024) {
        __syncthreads();
        for (int t = threadIdx.x; t < 1024; t += 256) {
            t_s[t] = j0 + t < m ? w_score[j0 + t] : 0.0;
            t_o[t] = j0 + t < m ? w_owner[j0 + t] : 0;
        }
        __syncthreads();
        if (owner) {
            const int32_t n = m - j0 < 1024 ? m - j0 : 1024;
            for (int32_t t = 0; t < n; ++t)
                if (t_o[t] && outranks(t_s[t], j0 + t, score, i)) ++pos;
        }
    }
    if (owner && pos < top_n) {
        anrag_candidate r;
        r.score = score;
        r.doc = w_ids[i];
        out[pos] = r;
    }
}

void free_wrrf_scratch(anrag_index *idx) {
    void *ptrs[] = {idx->d_w_ids, idx->d_w_in, idx->d_w_contrib, idx->d_w_score, idx->d_w_first, idx->d_w_out,
                    idx->d_w_count, idx->d_w_blob};
    idx->d_w_blob = nullptr;
    idx->w_blob_bytes = 0;
    for (void *p : ptrs)
        if (p) (void)counted_free(p);
    idx->d_w_ids = idx->d_w_in = nullptr;
    idx->d_w_contrib = idx->d_w_score = nullptr;
    idx->d_w_first = idx->d_w_count = nullptr;
    idx->d_w_out = nullptr;
    idx->wrrf_cap = 0;
}

int ensure_wrrf_scratch(anrag_index *idx, int64_t m) {
    if (idx->wrrf_cap >= m && idx->d_w_count) return ANRAG_OK;
    free_wrrf_scratch(idx);
    const int64_t cap = m < 4096 ? 4096 : m;
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_ids), (size_t)cap * 8));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_in), (size_t)cap * 8));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_contrib), (size_t)cap * 8));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_score), (size_t)cap * 8));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_first), (size_t)cap * 4));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_out), (size_t)cap * sizeof(anrag_candidate)));
    ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_w_count), 64));
    idx->wrrf_cap = cap;
    return ANRAG_OK;
}

int launch_wrrf(anrag_index *idx, hipStream_t st, const int64_t *d_ids, const anrag_candidate *d_cands,
                const int32_t *h_off, const double *h_weight, int32_t n_lists, double k, int32_t top_n,
                anrag_candidate *d_out, int32_t *d_count) {
    ANRAG_REQUIRE(n_lists >= 1 && n_lists <= ANRAG_WRRF_MAX_LISTS, "n_lists %d out of range [1, %d]", n_lists,
                  ANRAG_WRRF_MAX_LISTS);
    WrrfLists L;
    L.n = n_lists;
    for (int l = 0; l <= n_lists; ++l) L.off[l] = h_off[l];
    for (int l = 0; l < n_lists; ++l) L.w[l] = h_weight[l];
    const int32_t m = h_off[n_lists];
    LaunchTimer t(idx, ANRAG_KERNEL_WRRF, st);
    if (m <= kWrrfSmall) {
        int threads = 64;  // one thread per entry: 50 entries on the hybrid hot path
        while (threads < m) threads <<= 1;
        wrrf_small_kernel<<<1, threads, 0, st>>>(d_ids, d_cands, L, m, k, top_n, d_out, d_count);
        ANRAG_HIP(hipGetLastError());
        return ANRAG_OK;
    }
    int rc = ensure_wrrf_scratch(idx, m);
    if (rc) return rc;
    const unsigned blocks = (unsigned)((m + 255) / 256);
    wrrf_contrib_kernel<<<blocks, 256, 0, st>>>(d_ids, d_cands, L, m, k, idx->d_w_ids, idx->d_w_contrib, d_count);
    if (m > kWrrfAllPairsMax)  // long lists: three radix sorts instead of M^2 comparisons
        return wrrf_sorted(idx, st, idx->d_w_ids, idx->d_w_contrib, m, top_n, d_out, d_count);
    wrrf_sum_kernel<<<blocks, 256, 0, st>>>(idx->d_w_ids, idx->d_w_contrib, m, idx->d_w_score, idx->d_w_first, d_count);
    wrrf_rank_kernel<<<blocks, 256, 0, st>>>(idx->d_w_ids, idx->d_w_score, idx->d_w_first, m, top_n, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
