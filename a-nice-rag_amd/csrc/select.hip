// K4 -- selection over candidate lists.
//
//  * the per-workgroup lists K1 / K3 produce are merged by the tail kernel (tail.hip); what is left here is the
//    cross-shard merge:
//  * merge_candidates_kernel: n_lists sorted lists of k anrag_candidate records (the RCCL
//    all-gather receive buffer of the sharded path) -> global top-k by
//    (score desc, doc asc).  Replicated on every rank, latency-bound, one wavefront.
#include "common.hpp"
#include "wave_topk.hpp"
#include "wrrf_block.hpp"

namespace anrag {

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
        const double up_s = lane_up1(s);
        const long long up_d = __double_as_longlong(lane_up1(__longlong_as_double(d)));  // bits only
        const uint32_t up_ahead = lane_up1((uint32_t)ahead);
        if (!ahead) {
            const bool first = (lane == 0) || up_ahead;
            s = first ? cs : up_s;
            d = first ? cd : up_d;
        }
        thr_s = read_lane(s, k - 1);
        thr_d = __double_as_longlong(read_lane(__longlong_as_double(d), k - 1));
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
            const double cs = read_lane(c.score, i);
            const long long cd = __double_as_longlong(read_lane(__longlong_as_double((long long)c.doc), i));
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

// Global tail of the sharded path, ONE launch: per modality merge the n_lists per-shard candidate lists
// (record layout of one rank: [0,k) dense, [k,2k) BM25), then weighted RRF + top-n on the global ranks.
__global__ __launch_bounds__(128) void merge_fuse_kernel(const anrag_candidate *__restrict__ lists, int32_t n_lists,
                                                         int32_t k, int64_t stride, double w_dense, double w_bm25,
                                                         double wrrf_k, int32_t top_n,
                                                         anrag_candidate *__restrict__ out,
                                                         int32_t *__restrict__ count) {
    __shared__ int64_t s_id[2 * kListLen];
    __shared__ double s_c[2 * kListLen];
    __shared__ double s_score[2 * kListLen];
    __shared__ int32_t s_owner[2 * kListLen];
    __shared__ int32_t s_distinct;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;  // wave 0: dense lists, wave 1: BM25 lists
    // one workgroup per query of the exchanged group: its blocks sit 2k records apart inside a shard's slab
    lists += (int64_t)blockIdx.x * 2 * k;
    out += (int64_t)blockIdx.x * top_n;
    count += blockIdx.x;
    DocTopK top;
    top.init(k);
    for (int li = 0; li < n_lists; ++li) {
        anrag_candidate c;
        c.score = -__builtin_huge_val();
        c.doc = -1;
        if (lane < k) c = lists[(int64_t)li * stride + wave * k + lane];
        for (int i = 0; i < k; ++i) {
            const double cs = read_lane(c.score, i);
            const long long cd = __double_as_longlong(read_lane(__longlong_as_double((long long)c.doc), i));
            if (!top.admits(cs, cd)) break;
            top.insert(cs, cd);
        }
    }
    if (lane < k) {
        const int slot = wave * k + lane;
        const double w = wave ? w_bm25 : w_dense;
        s_id[slot] = top.d;
        s_c[slot] = top.d < 0 ? 0.0 : w * (1.0 / (wrrf_k + (double)(lane + 1)));
    }
    __syncthreads();
    wrrf_in_block(s_id, s_c, s_score, s_owner, &s_distinct, 2 * k, top_n, out, count);
}

int launch_merge_fuse(anrag_index *idx, hipStream_t st, const anrag_candidate *d_lists, int32_t n_lists, int32_t k,
                      int64_t list_stride, double w_dense, double w_bm25, double wrrf_k, int32_t top_n,
                      int32_t n_queries, anrag_candidate *d_out, int32_t *d_count) {
    LaunchTimer t(idx, ANRAG_KERNEL_WRRF, st);
    merge_fuse_kernel<<<n_queries, 128, 0, st>>>(d_lists, n_lists, k, list_stride, w_dense, w_bm25, wrrf_k, top_n, d_out,
                                         d_count);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

int launch_merge_candidates(anrag_index *idx, hipStream_t st, const anrag_candidate *d_lists, int32_t n_lists,
                            int32_t k, int64_t list_stride, anrag_candidate *d_out) {
    LaunchTimer t(idx, ANRAG_KERNEL_SELECT, st);
    merge_candidates_kernel<<<1, 64, 0, st>>>(d_lists, n_lists, k, list_stride, d_out);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag
