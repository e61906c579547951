// K4 -- selection over candidate lists.
//
//  * merge_block_lists_kernel<S>: the per-workgroup sorted lists K1 / K3 leave in HBM
//    ([n_lists][64] score + local row) -> the shard's top-k as anrag_candidate {fp64 score, doc id}.
//    One workgroup, lane <-> list, round j offers every still-live list's j-th entry: a list drops
//    out at its first loser (it is sorted), so the number of dependent HBM round trips is the
//    largest number of winners any single list holds (+1), not k and not n_lists.
//  * merge_candidates_kernel: n_lists sorted lists of k anrag_candidate records (the RCCL
//    all-gather receive buffer of the sharded path) -> global top-k by
//    (score desc, doc asc).  Replicated on every rank, latency-bound, one wavefront.
#include "common.hpp"
#include "wave_topk.hpp"

namespace anrag {

template <typename S>
__global__ __launch_bounds__(256) void merge_block_lists_kernel(const S *__restrict__ blk_score,
                                                                const uint32_t *__restrict__ blk_row,
                                                                int32_t n_lists, int32_t k,
                                                                const int64_t *__restrict__ doc_of_row,
                                                                int64_t doc_base, anrag_candidate *__restrict__ out) {
    constexpr int W = 4;
    __shared__ S lds_s[W * kListLen];
    __shared__ uint32_t lds_r[W * kListLen];
    const int lane = lane_id(), wave = threadIdx.x / kWave;
    WaveTopK<S> top;
    top.init(k);
    for (int l0 = wave * kWave; l0 < n_lists; l0 += W * kWave) {
        const int list = l0 + lane;
        bool live = list < n_lists;
        const S *ps = blk_score + (int64_t)(live ? list : 0) * kListLen;
        const uint32_t *pr = blk_row + (int64_t)(live ? list : 0) * kListLen;
        S cs = live ? ps[0] : neg_inf<S>();
        uint32_t cr = live ? pr[0] : kNoRow;
        for (int j = 0; j < k; ++j) {
            // next round's entries are in flight while this round is merged
            const bool more = live && (j + 1 < k);
            const S ns = more ? ps[j + 1] : neg_inf<S>();
            const uint32_t nr = more ? pr[j + 1] : kNoRow;
            const bool cand = live && cr != kNoRow && top.admits(cs, cr);
            if (__ballot(cand) == 0) break;
            top.offer_lanes(cand, cs, cr);
            // still inside the top-k after everybody's insertions?  otherwise the list is exhausted
            live = cand && !beats(top.thr_s, top.thr_r, cs, cr);
            cs = ns;
            cr = nr;
        }
    }
    block_merge(top, lds_s, lds_r, W);
    if (wave == 0 && lane < k) {
        anrag_candidate c;
        const bool empty = top.r == kNoRow;
        c.score = empty ? -__builtin_huge_val() : (double)top.s;
        c.doc = empty ? -1 : (doc_of_row ? doc_of_row[top.r] : doc_base + (int64_t)top.r);
        out[lane] = c;
    }
}

int launch_merge_block_lists_f32(anrag_index *idx, hipStream_t st, const float *blk_score, const uint32_t *blk_row,
                                 int32_t n_lists, int32_t k, const int64_t *doc_of_row, int64_t doc_base,
                                 anrag_candidate *d_out) {
    LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
    merge_block_lists_kernel<float><<<1, 256, 0, st>>>(blk_score, blk_row, n_lists, k, doc_of_row, doc_base, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

int launch_merge_block_lists_f64(anrag_index *idx, hipStream_t st, const double *blk_score, const uint32_t *blk_row,
                                 int32_t n_lists, int32_t k, const int64_t *doc_of_row, int64_t doc_base,
                                 anrag_candidate *d_out) {
    LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
    merge_block_lists_kernel<double><<<1, 256, 0, st>>>(blk_score, blk_row, n_lists, k, doc_of_row, doc_base, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

// ------------------------------------------------------------------ cross-shard merge
// Same register top-k as wave_topk.hpp but keyed on (fp64 score, int64 doc).
struct DocTopK {
    double s;
    long long d;
    double thr_s;
    long long thr_d;
    int k;
    static __device__ __forceinline__ bool beats(double s, long long d, double ts, long long td) {
        // an empty slot is doc -1 / -inf and loses to everything real
        if (td < 0) return d >= 0;
        if (d < 0) return false;
        return s > ts || (s == ts && d < td);
    }
    __device__ __forceinline__ void init(int k_) {
        s = -__builtin_huge_val();
        d = -1;
        thr_s = s;
        thr_d = -1;
        k = k_;
    }
    __device__ __forceinline__ bool admits(double cs, long long cd) const { return beats(cs, cd, thr_s, thr_d); }
    __device__ __forceinline__ void insert(double cs, long long cd) {
        const int lane = threadIdx.x & 63;
        const bool ahead = beats(s, d, cs, cd);
        const double up_s = __shfl_up(s, 1);
        const long long up_d = __shfl_up(d, 1);
        const int up_ahead = __shfl_up((int)ahead, 1);
        if (!ahead) {
            const bool first = (lane == 0) || up_ahead;
            s = first ? cs : up_s;
            d = first ? cd : up_d;
        }
        thr_s = __shfl(s, k - 1);
        thr_d = __shfl(d, k - 1);
    }
};

__global__ __launch_bounds__(64) void merge_candidates_kernel(const anrag_candidate *__restrict__ lists,
                                                              int32_t n_lists, int32_t k, int64_t stride,
                                                              anrag_candidate *__restrict__ out) {
    const int lane = threadIdx.x;
    DocTopK top;
    top.init(k);
    for (int li = 0; li < n_lists; ++li) {
        // lane i <- record i of the list (k <= 64), then a uniform walk until the first loser
        anrag_candidate c;
        c.score = -__builtin_huge_val();
        c.doc = -1;
        if (lane < k) c = lists[(int64_t)li * stride + lane];
        for (int i = 0; i < k; ++i) {
            const double cs = __shfl(c.score, i);
            const long long cd = __shfl((long long)c.doc, i);
            if (!top.admits(cs, cd)) break;
            top.insert(cs, cd);
        }
    }
    if (lane < k) {
        anrag_candidate c;
        c.score = top.d < 0 ? -__builtin_huge_val() : top.s;
        c.doc = top.d;
        out[lane] = c;
    }
}

int launch_merge_candidates(anrag_index *idx, hipStream_t st, const anrag_candidate *d_lists, int32_t n_lists,
                            int32_t k, int64_t list_stride, anrag_candidate *d_out) {
    LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
    merge_candidates_kernel<<<1, 64, 0, st>>>(d_lists, n_lists, k, list_stride, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
