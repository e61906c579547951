// Query tail -- everything after the two streaming kernels, in ONE launch off the scan's critical path.
//
// K1 (dense scan) leaves one sorted 64-entry list per WAVE, K3 (BM25) one per workgroup, in HBM.  This kernel
//   1. merges the dense lists (waves 0-3) and the BM25 partition lists (waves 4-7) into per-modality top-k
//      (the reference's argpartition+argsort, src/search_engine.py:83-87 / :236-243, finished),
//   2. maps local rows to doc ids,
//   3. either writes both candidate lists (the per-shard all-gather payload) or fuses them: weighted RRF +
//      top-n (search_engine.py:21-34 + query_rag_retrieval.py:360-362).
// It runs on the secondary / fusion stream while the NEXT query's scan owns the primary stream, so none of
// its latency (a few dependent HBM round trips) is on the throughput path: "combine in the next kernel's
// prologue" rather than a last-workgroup merge inside the scan (measured: that put ~20 us of one-workgroup
// work at the end of every scan).
#include "common.hpp"
#include "wave_topk.hpp"
#include "wrrf_block.hpp"

namespace anrag {

constexpr int kTailThreads = 512;  // 8 waves: 4 per modality

// one workgroup per query of a group (blockIdx.x): the tails of the queries that shared a scan / K3 launch share one too
struct TailQuery {
    const float *d_score;     // dense block lists
    const uint32_t *d_row;
    const double *b_score;    // BM25 partition lists
    const uint32_t *b_row;
    int32_t n_dense;          // 0 = no dense leg
    int32_t n_bm25;           // 0 = no BM25 leg
    anrag_candidate *out;
    int32_t *count;
};
struct TailGroup {
    TailQuery q[kScanGroupMax];
    const int64_t *dense_doc;
    int64_t dense_base;
    const int64_t *bm25_doc;
    int64_t bm25_base;
    int32_t k, mode, top_n;
    double w_dense, w_bm25, wrrf_k;
};
struct TailArgs : TailQuery {
    const int64_t *dense_doc;
    int64_t dense_base;
    const int64_t *bm25_doc;
    int64_t bm25_base;
    int32_t k, mode, top_n;
    double w_dense, w_bm25, wrrf_k;
};

template <bool GROUPED>
__global__ __launch_bounds__(kTailThreads) void query_tail_kernel(TailGroup grp) {
    TailArgs a;
    static_cast<TailQuery &>(a) = grp.q[GROUPED ? (int)blockIdx.x : 0];  // one query: fixed kernarg offsets
    a.dense_doc = grp.dense_doc;
    a.dense_base = grp.dense_base;
    a.bm25_doc = grp.bm25_doc;
    a.bm25_base = grp.bm25_base;
    a.k = grp.k;
    a.mode = grp.mode;
    a.top_n = grp.top_n;
    a.w_dense = grp.w_dense;
    a.w_bm25 = grp.w_bm25;
    a.wrrf_k = grp.wrrf_k;
    __shared__ float lf_s[4 * kListLen];
    __shared__ uint32_t lf_r[4 * kListLen];
    __shared__ double ld_s[4 * kListLen];
    __shared__ uint32_t ld_r[4 * kListLen];
    __shared__ int64_t s_id[2 * kListLen];
    __shared__ double s_c[2 * kListLen];
    __shared__ double s_score[2 * kListLen];
    __shared__ int32_t s_owner[2 * kListLen];
    __shared__ int32_t s_distinct;
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int group = wave >> 2, gw = wave & 3;  // group 0: dense (fp32 scores), group 1: BM25 (fp64)
    const int k = a.k;

    WaveTopK<float> tf;
    WaveTopK<double> td;
    tf.init(k);
    td.init(k);
    if (group == 0) {
        if (a.n_dense > 0) merge_lists(tf, a.d_score, a.d_row, a.n_dense, k, gw, 4);
    } else {
        if (a.n_bm25 > 0) merge_lists(td, a.b_score, a.b_row, a.n_bm25, k, gw, 4);
    }
    // tree merge of each group's 4 waves; both groups walk the same barriers
    for (int half = 2; half >= 1; half >>= 1) {
        if (gw >= half && gw < 2 * half) {
            if (group == 0) {
                lf_s[gw * kListLen + lane] = tf.s;
                lf_r[gw * kListLen + lane] = tf.r;
            } else {
                ld_s[gw * kListLen + lane] = td.s;
                ld_r[gw * kListLen + lane] = td.r;
            }
        }
        __syncthreads();
        if (gw < half) {
            if (group == 0) tf.merge_sorted(lf_s + (gw + half) * kListLen, lf_r + (gw + half) * kListLen);
            else td.merge_sorted(ld_s + (gw + half) * kListLen, ld_r + (gw + half) * kListLen);
        }
        __syncthreads();
    }
    // candidates: entries [0,k) dense, [k,2k) BM25 (an absent leg = all padding)
    if (gw == 0 && lane < k) {
        const int slot = group * k + lane;
        if (group == 0) {
            const bool empty = tf.r == kNoRow || a.n_dense == 0;
            s_id[slot] = empty ? -1 : (a.dense_doc ? a.dense_doc[tf.r] : a.dense_base + (int64_t)tf.r);
            s_score[slot] = empty ? -__builtin_huge_val() : (double)tf.s;
        } else {
            const bool empty = td.r == kNoRow || a.n_bm25 == 0;
            s_id[slot] = empty ? -1 : (a.bm25_doc ? a.bm25_doc[td.r] : a.bm25_base + (int64_t)td.r);
            s_score[slot] = empty ? -__builtin_huge_val() : td.s;
        }
    }
    __syncthreads();
    if (a.mode != kTailFuse) {
        // one modality only: its k records at out[0..k); both (or kTailCandidates2k): dense then BM25
        const bool both = a.mode == kTailCandidates2k || (a.n_dense > 0 && a.n_bm25 > 0);
        const int first = (both || a.n_dense > 0) ? 0 : k;
        const int n_out = both ? 2 * k : k;
        if (tid < n_out) {
            anrag_candidate c;
            c.score = s_score[first + tid];
            c.doc = s_id[first + tid];
            a.out[tid] = c;
        }
        return;
    }
    // fuse: contribution of entry i = w_list * (1 / (wrrf_k + rank)), rank from 1 inside its list
    const int m = 2 * k;
    if (tid < m) {
        const int list = tid >= k;
        const int rank = tid - list * k + 1;
        const double w = list ? a.w_bm25 : a.w_dense;
        s_c[tid] = s_id[tid] < 0 ? 0.0 : w * (1.0 / (a.wrrf_k + (double)rank));
    }
    __syncthreads();
    wrrf_in_block(s_id, s_c, s_score, s_owner, &s_distinct, m, a.top_n, a.out, a.count);
}

int launch_tail_group(anrag_index *idx, hipStream_t st, const int *sets, bool use_dense, const bool *use_bm25, int32_t n,
                      int32_t k, TailMode mode, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                      anrag_candidate *const *d_out, int32_t *const *d_count) {
    ANRAG_REQUIRE(n >= 1 && n <= kScanGroupMax, "tail group of %d queries", n);
    TailGroup a;
    for (int i = 0; i < kScanGroupMax; ++i) {
        const int j = i < n ? i : 0;
        TailQuery &q = a.q[i];
        q.d_score = idx->d_blk_score_f32 + (int64_t)sets[j] * kMaxScanLists * kListLen;
        q.d_row = idx->d_blk_row_a + (int64_t)sets[j] * kMaxScanLists * kListLen;
        q.n_dense = use_dense ? dense_scan_lists(idx) : 0;
        q.b_score = idx->d_blk_score_f64 + (int64_t)sets[j] * idx->n_parts * kListLen;
        q.b_row = idx->d_blk_row_b + (int64_t)sets[j] * idx->n_parts * kListLen;
        q.n_bm25 = use_bm25[j] ? idx->n_parts : 0;
        q.out = d_out[j];
        q.count = d_count ? d_count[j] : nullptr;
    }
    a.dense_doc = idx->d_dense_doc;
    a.dense_base = idx->dense_doc_base;
    a.bm25_doc = idx->d_bm25_doc;
    a.bm25_base = idx->bm25_doc_base;
    a.k = k;
    a.mode = (int32_t)mode;
    a.top_n = top_n;
    a.w_dense = w_dense;
    a.w_bm25 = w_bm25;
    a.wrrf_k = wrrf_k;
    LaunchTimer t(idx, mode == kTailFuse ? ANRAG_KERNEL_WRRF : ANRAG_KERNEL_SELECT, st, n);
    if (n == 1) query_tail_kernel<false><<<1, kTailThreads, 0, st>>>(a);
    else query_tail_kernel<true><<<n, kTailThreads, 0, st>>>(a);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

int launch_tail(anrag_index *idx, hipStream_t st, int set, bool use_dense, bool use_bm25, int32_t k, TailMode mode,
                double w_dense, double w_bm25, double wrrf_k, int32_t top_n, anrag_candidate *d_out,
                int32_t *d_count) {
    return launch_tail_group(idx, st, &set, use_dense, &use_bm25, 1, k, mode, w_dense, w_bm25, wrrf_k, top_n, &d_out,
                             d_count ? &d_count : nullptr);
}

}  // namespace anrag
