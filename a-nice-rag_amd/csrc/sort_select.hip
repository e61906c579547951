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
        if (idx->d_sort_buf) (void)hipFree(idx->d_sort_buf);
        idx->d_sort_buf = nullptr;
        idx->sort_buf_bytes = 0;
        ANRAG_HIP(hipMalloc(&idx->d_sort_buf, (size_t)need));
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
        if (idx->d_sort_tmp) (void)hipFree(idx->d_sort_tmp);
        idx->d_sort_tmp = nullptr;
        idx->sort_tmp_bytes = 0;
        ANRAG_HIP(hipMalloc(&idx->d_sort_tmp, tmp));
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
        ANRAG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_scores_f32), (size_t)n * sizeof(float)));
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

int bm25_search_large_k(anrag_index *idx, hipStream_t st, const int32_t *d_terms, int32_t n_terms, int32_t k,
                        const uint32_t *d_allow_bits, int64_t *out_doc, double *out_score, int32_t *out_count) {
    const int64_t n = idx->n_docs;
    if (!idx->d_scores_f64) {
        ANRAG_HIP(hipMalloc(reinterpret_cast<void **>(&idx->d_scores_f64), (size_t)n * sizeof(double)));
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

}  // namespace anrag
