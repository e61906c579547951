// K4 (large k) -- full ranking of a score array, for k > ANRAG_FUSED_K_MAX.
//
// retrieval_eval.py runs 7 of its 9 configurations with similarity_k = 12000 (> corpus size, i.e. a
// full ranking, src/retrieval_eval.py:142-143) to report mean/median/max rank.  K1 / K3 then write every
// score (filtered rows as -inf) and this file sorts them: rocPRIM's LSD radix sort on the fp32 / fp64
// keys, descending, with the row as payload.  LSD radix sort is stable and the payload enters in row
// order, so equal scores come out row-ascending -- the same (score desc, row asc) rule as the fused path.
#include <cstring>

#include <rocprim/device/device_radix_sort.hpp>
#include <rocprim/iterator/counting_iterator.hpp>

#include <algorithm>

#include "common.hpp"

namespace anrag {

template <typename S>
__global__ void emit_sorted_kernel(const S *__restrict__ keys, const uint32_t *__restrict__ rows, int64_t k,
                                   const int64_t *__restrict__ doc_of_row, int64_t doc_base,
                                   anrag_candidate *__restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const S s = keys[i];
    anrag_candidate c;
    if (s == -__builtin_huge_val()) {  // filtered-out row
        c.score = -__builtin_huge_val();
        c.doc = -1;
    } else {
        const uint32_t r = rows[i];
        c.score = (double)s;
        c.doc = doc_of_row ? doc_of_row[r] : doc_base + (int64_t)r;
    }
    out[i] = c;
}

// Sort n scores descending, leave the first k as candidates in idx->d_sort_buf's candidate region.
template <typename S>
static int sort_scores(anrag_index *idx, hipStream_t st, const S *d_scores, int64_t n, int64_t k,
                       const int64_t *doc_of_row, int64_t doc_base, anrag_candidate **d_out) {
    const int64_t need = n * (int64_t)(sizeof(S) + sizeof(uint32_t)) + k * (int64_t)sizeof(anrag_candidate) + 256;
    if (idx->sort_buf_bytes < need) {
        if (idx->d_sort_buf) (void)counted_free(idx->d_sort_buf);
        idx->d_sort_buf = nullptr;
        idx->sort_buf_bytes = 0;
        ANRAG_HIP(counted_malloc(&idx->d_sort_buf, (size_t)need));
        idx->sort_buf_bytes = need;
    }
    char *base = static_cast<char *>(idx->d_sort_buf);
    anrag_candidate *cands = reinterpret_cast<anrag_candidate *>(base);
    S *keys_out = reinterpret_cast<S *>(base + ((k * sizeof(anrag_candidate) + 255) / 256) * 256);
    uint32_t *rows_out = reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(keys_out) + n * sizeof(S));
    rocprim::counting_iterator<uint32_t> rows_in(0);
    size_t tmp = 0;
    ANRAG_HIP(rocprim::radix_sort_pairs_desc(nullptr, tmp, d_scores, keys_out, rows_in, rows_out, (size_t)n, 0,
                                             8 * sizeof(S), st));
    if ((int64_t)tmp > idx->sort_tmp_bytes) {
        if (idx->d_sort_tmp) (void)counted_free(idx->d_sort_tmp);
        idx->d_sort_tmp = nullptr;
        idx->sort_tmp_bytes = 0;
        ANRAG_HIP(counted_malloc(&idx->d_sort_tmp, tmp));
        idx->sort_tmp_bytes = (int64_t)tmp;
    }
    {
        LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
        ANRAG_HIP(rocprim::radix_sort_pairs_desc(idx->d_sort_tmp, tmp, d_scores, keys_out, rows_in, rows_out,
                                                 (size_t)n, 0, 8 * sizeof(S), st));
        emit_sorted_kernel<S><<<(unsigned)((k + 255) / 256), 256, 0, st>>>(keys_out, rows_out, k, doc_of_row, doc_base,
                                                                           cands);
        ANRAG_HIP(hipGetLastError());
    }
    *d_out = cands;
    return ANRAG_OK;
}

template <typename T>
static int fetch_ranked(hipStream_t st, const anrag_candidate *d_cands, int64_t have, int32_t k, int64_t *out_doc,
                        T *out_score, int32_t *out_count) {
    std::vector<anrag_candidate> h((size_t)have);
    ANRAG_HIP(hipMemcpyAsync(h.data(), d_cands, (size_t)have * sizeof(anrag_candidate), hipMemcpyDeviceToHost, st));
    ANRAG_HIP(hipStreamSynchronize(st));
    int32_t cnt = 0;
    for (int64_t i = 0; i < k; ++i) {
        if (i < have && h[i].doc >= 0) {
            out_doc[i] = h[i].doc;
            out_score[i] = (T)h[i].score;
            ++cnt;
        } else {
            out_doc[i] = -1;
            out_score[i] = -std::numeric_limits<T>::infinity();
        }
    }
    *out_count = cnt;
    return ANRAG_OK;
}

int dense_search_large_k(anrag_index *idx, hipStream_t st, const float *h_queries, int32_t n_queries, int32_t k,
                         const uint32_t *d_allow_bits, int64_t *out_doc, float *out_score, int32_t *out_count) {
    const int64_t n = idx->n_rows;
    if (!idx->d_scores_f32) {
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_scores_f32), (size_t)n * sizeof(float)));
        idx->hbm_bytes += n * 4;
    }
    const int64_t have = std::min<int64_t>(k, n);
    for (int32_t qi = 0; qi < n_queries; ++qi) {
        ANRAG_HIP(hipMemcpyAsync(idx->d_query, h_queries + (int64_t)qi * idx->dim, (size_t)idx->dim * sizeof(float),
                                 hipMemcpyHostToDevice, st));
        int rc = launch_dense_topk(idx, st, idx->d_query, 0, d_allow_bits, nullptr, idx->d_scores_f32);
        if (rc) return rc;
        anrag_candidate *d_c = nullptr;
        if ((rc = sort_scores<float>(idx, st, idx->d_scores_f32, n, have, idx->d_dense_doc, idx->dense_doc_base, &d_c)))
            return rc;
        if ((rc = fetch_ranked<float>(st, d_c, have, k, out_doc + (int64_t)qi * k, out_score + (int64_t)qi * k,
                                      out_count + qi)))
            return rc;
    }
    return ANRAG_OK;
}

// fp64 query: fp64 score array + the same sort, for any k (one query at a time: the reference's text path)
int dense_search_f64(anrag_index *idx, hipStream_t st, const double *h_query, int32_t k, const uint32_t *d_allow_bits,
                     int64_t *out_doc, double *out_score, int32_t *out_count) {
    const int64_t n = idx->n_rows;
    if (!idx->d_dense_scores_f64) {
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_dense_scores_f64), (size_t)n * sizeof(double)));
        idx->hbm_bytes += n * 8;
    }
    if (!idx->d_query_f64) ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_query_f64), 65536 * sizeof(double)));
    ANRAG_HIP(hipMemcpyAsync(idx->d_query_f64, h_query, (size_t)idx->dim * sizeof(double), hipMemcpyHostToDevice, st));
    int rc = launch_dense_scores_f64(idx, st, idx->d_query_f64, d_allow_bits, idx->d_dense_scores_f64);
    if (rc) return rc;
    const int64_t have = std::min<int64_t>(k, n);
    anrag_candidate *d_c = nullptr;
    if ((rc = sort_scores<double>(idx, st, idx->d_dense_scores_f64, n, have, idx->d_dense_doc, idx->dense_doc_base, &d_c)))
        return rc;
    return fetch_ranked<double>(st, d_c, have, k, out_doc, out_score, out_count);
}

int bm25_search_large_k(anrag_index *idx, hipStream_t st, const int32_t *d_terms, int32_t n_terms, int32_t k,
                        const uint32_t *d_allow_bits, int64_t *out_doc, double *out_score, int32_t *out_count) {
    const int64_t n = idx->n_docs;
    if (!idx->d_scores_f64) {
        ANRAG_HIP(counted_malloc(reinterpret_cast<void **>(&idx->d_scores_f64), (size_t)n * sizeof(double)));
        idx->hbm_bytes += n * 8;
        idx->bm25_hbm_bytes += n * 8;
    }
    const int64_t have = std::min<int64_t>(k, n);
    int rc = launch_bm25(idx, st, d_terms, n_terms, 0, d_allow_bits, nullptr, idx->d_scores_f64);
    if (rc) return rc;
    anrag_candidate *d_c = nullptr;
    if ((rc = sort_scores<double>(idx, st, idx->d_scores_f64, n, have, idx->d_bm25_doc, idx->bm25_doc_base, &d_c)))
        return rc;
    return fetch_ranked<double>(st, d_c, have, k, out_doc, out_score, out_count);
}

// ------------------------------------------------------------------ K5, long lists (full-ranking mode)
// Weighted RRF over M = a few 10^4 entries (two 12,000-id lists in retrieval_eval's configurations).  The
// reference's dict update and stable sort (src/search_engine.py:21-34) as three stable radix sorts:
//   1. entries by id (payload: entry index)            -> each id's entries are adjacent, in entry order
//   2. one thread per group head adds the group's contributions in that order (the dict's `+=` sequence, same
//      fp64 bits), remembers the first entry index (= dict insertion order)
//   3. groups by first entry index, then (stable) by score descending -> `sorted(..., reverse=True)`'s order
// All sorts run over M slots (unused slots carry first = 2^32-1, score = -inf), so nothing syncs the host.
// The all-pairs grid form this replaces above 4,096 entries took 3.7 ms for 2 x 9,609 ids.
__global__ void wrrf_fill_kernel(int32_t m, uint32_t *__restrict__ g_first, double *__restrict__ g_score,
                                 int64_t *__restrict__ g_id) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    g_first[i] = 0xFFFFFFFFu;
    g_score[i] = -__builtin_huge_val();
    g_id[i] = -1;
}

__global__ void wrrf_group_kernel(const int64_t *__restrict__ s_id, const uint32_t *__restrict__ s_entry,
                                  const double *__restrict__ contrib, int32_t m, uint32_t *__restrict__ g_first,
                                  double *__restrict__ g_score, int64_t *__restrict__ g_id,
                                  int32_t *__restrict__ count) {
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= m) return;
    const int64_t id = s_id[p];
    if (id < 0 || (p > 0 && s_id[p - 1] == id)) return;  // not a group head (negative ids carry nothing)
    double score = 0.0;
    for (int32_t q = p; q < m && s_id[q] == id; ++q) score = score + contrib[s_entry[q]];
    const int32_t slot = atomicAdd(count, 1);
    g_first[slot] = s_entry[p];
    g_score[slot] = score;
    g_id[slot] = id;
}

__global__ void wrrf_gather_kernel(const uint32_t *__restrict__ order, const double *__restrict__ g_score, int32_t m,
                                   double *__restrict__ key) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < m) key[i] = g_score[order[i]];
}

__global__ void wrrf_emit_kernel(const uint32_t *__restrict__ order, const double *__restrict__ g_score,
                                 const int64_t *__restrict__ g_id, int32_t m, int32_t top_n,
                                 anrag_candidate *__restrict__ out) {
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m || i >= top_n) return;
    const uint32_t slot = order[i];
    if (g_id[slot] < 0) return;  // past the last distinct id
    anrag_candidate r;
    r.score = g_score[slot];
    r.doc = g_id[slot];
    out[i] = r;
}

// d_ids / d_contrib: the M entries in list order (wrrf_contrib_kernel); *d_count must be 0 on entry (that kernel
// clears it) and ends as the number of distinct ids.
int wrrf_sorted(anrag_index *idx, hipStream_t st, const int64_t *d_ids, const double *d_contrib, int32_t m,
                int32_t top_n, anrag_candidate *d_out, int32_t *d_count) {
    const int64_t per = 8 + 4 + 4 + 8 + 8 + 4 + 4 + 8 + 8 + 4;  // the arrays carved below
    const int64_t need = (int64_t)m * per + 16 * 256;
    if (idx->w_blob_bytes < need) {
        if (idx->d_w_blob) (void)counted_free(idx->d_w_blob);
        idx->d_w_blob = nullptr;
        idx->w_blob_bytes = 0;
        ANRAG_HIP(counted_malloc(&idx->d_w_blob, (size_t)need));
        idx->w_blob_bytes = need;
    }
    char *at = static_cast<char *>(idx->d_w_blob);
    auto carve = [&](size_t bytes) {
        char *p = at;
        at += (bytes + 255) / 256 * 256;
        return p;
    };
    int64_t *s_id = reinterpret_cast<int64_t *>(carve((size_t)m * 8));
    uint32_t *s_entry = reinterpret_cast<uint32_t *>(carve((size_t)m * 4));
    uint32_t *g_first = reinterpret_cast<uint32_t *>(carve((size_t)m * 4));
    double *g_score = reinterpret_cast<double *>(carve((size_t)m * 8));
    int64_t *g_id = reinterpret_cast<int64_t *>(carve((size_t)m * 8));
    uint32_t *f_sorted = reinterpret_cast<uint32_t *>(carve((size_t)m * 4));
    uint32_t *order1 = reinterpret_cast<uint32_t *>(carve((size_t)m * 4));
    double *key2 = reinterpret_cast<double *>(carve((size_t)m * 8));
    double *key2_sorted = reinterpret_cast<double *>(carve((size_t)m * 8));
    uint32_t *order2 = reinterpret_cast<uint32_t *>(carve((size_t)m * 4));
    rocprim::counting_iterator<uint32_t> iota(0);
    size_t t1 = 0, t2 = 0, t3 = 0;
    ANRAG_HIP(rocprim::radix_sort_pairs(nullptr, t1, d_ids, s_id, iota, s_entry, (size_t)m, 0, 64, st));
    ANRAG_HIP(rocprim::radix_sort_pairs(nullptr, t2, g_first, f_sorted, iota, order1, (size_t)m, 0, 32, st));
    ANRAG_HIP(rocprim::radix_sort_pairs_desc(nullptr, t3, key2, key2_sorted, order1, order2, (size_t)m, 0, 64, st));
    const size_t tmp = std::max(t1, std::max(t2, t3));
    if ((int64_t)tmp > idx->sort_tmp_bytes) {
        if (idx->d_sort_tmp) (void)counted_free(idx->d_sort_tmp);
        idx->d_sort_tmp = nullptr;
        idx->sort_tmp_bytes = 0;
        ANRAG_HIP(counted_malloc(&idx->d_sort_tmp, tmp));
        idx->sort_tmp_bytes = (int64_t)tmp;
    }
    const unsigned blocks = (unsigned)((m + 255) / 256);
    size_t sz = tmp;
    ANRAG_HIP(rocprim::radix_sort_pairs(idx->d_sort_tmp, sz, d_ids, s_id, iota, s_entry, (size_t)m, 0, 64, st));
    wrrf_fill_kernel<<<blocks, 256, 0, st>>>(m, g_first, g_score, g_id);
    wrrf_group_kernel<<<blocks, 256, 0, st>>>(s_id, s_entry, d_contrib, m, g_first, g_score, g_id, d_count);
    sz = tmp;
    ANRAG_HIP(rocprim::radix_sort_pairs(idx->d_sort_tmp, sz, g_first, f_sorted, iota, order1, (size_t)m, 0, 32, st));
    wrrf_gather_kernel<<<blocks, 256, 0, st>>>(order1, g_score, m, key2);
    sz = tmp;
    ANRAG_HIP(rocprim::radix_sort_pairs_desc(idx->d_sort_tmp, sz, key2, key2_sorted, order1, order2, (size_t)m, 0, 64, st));
    wrrf_emit_kernel<<<blocks, 256, 0, st>>>(order2, g_score, g_id, m, top_n, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
