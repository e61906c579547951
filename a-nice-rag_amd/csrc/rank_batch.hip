// Full-ranking mode for LISTS of queries: similarity_k up to ~13,000 (retrieval_eval.py asks for 12,000, i.e. "rank
// everything", in 7 of its 9 configurations: src/retrieval_eval.py:142-143, :155-156, ...), every step on the device,
// one host sync per chunk of queries.
//
// Per chunk of C queries and per leg (a dense model or BM25; the reference's per-model blocks,
// src/query_rag_retrieval.py:197-335):
//   1. scores of every row into a tile [C][N]: K1T (dense_tile.hip: a batch of rows in registers, the launch's queries over
//      it, bit for bit a single query's scan) / K3 in its score-writing form, the chunk's queries in one launch;
//   2. ONE workgroup per (query, leg): the reference's argsort()[::-1] / argpartition + argsort
//      (src/search_engine.py:83-87, :233-243) under the build's order (score desc, row asc), filtered rows (-inf)
//      excluded.  A segment that fits the LDS: seg_radix_sort_kernel, a stable LSD radix sort over the score bits (the
//      entries arrive in row order).  Longer segments: seg_topk_sort_kernel -- an MSD radix SELECT over (order-preserving
//      key bits, ~row) composites (11-bit digits, LDS histogram, one pass over the segment per digit, stops as soon as
//      "everything not below the current prefix" fits the LDS), then a bitonic network over the survivors.
//   3. fusion (two or more legs), src/search_engine.py:21-34 on row lists instead of id strings: w * (1 / (k + rank)) per
//      leg in order (a leg names a document once, so the additions of a document happen in leg order = the reference's
//      dict update order; same fp64 bits), ties to the first insertion (leg, position) = Python's stable sort.  An id
//      space that fits the LDS: rank_fuse_sort_kernel, everything in one workgroup per query; larger ones: F arrays in
//      HBM, an accumulate launch per leg, the select + sort kernel, an emit kernel.
//   4. only the rank of an expected document wanted (expect_id, out_id == NULL: src/retrieval_eval.py:75-82): counts
//      instead of sorts wherever a list is not needed (rank_count_kernel; the count in rank_fuse_sort_kernel).
// HBM- / issue-bound score tiles, LDS-bound sorts.
#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "common.hpp"
#include "wave_topk.hpp"

namespace anrag {

constexpr int kRankThreads = 1024;
constexpr int kRankDigitBits = 11;
constexpr int kRankCtrlWords = 64;
constexpr int kRankLoads = 8;  // independent loads per thread and step of a pass over a segment
constexpr uint32_t kTiePosBits = 26;  // fusion insertion key = leg << 26 | position in the leg's list

// W 32-bit words per composite, w0 least significant.  fp32 key: {~tie, key bits}; fp64: {~tie, key lo, key hi}.
// (Named members, not an array: an array that any path indexes with a variable lives in scratch memory.)
template <int W>
struct Comp;
template <>
struct Comp<2> {
    uint32_t w0, w1;
};
template <>
struct Comp<3> {
    uint32_t w0;  // ~tie
    uint64_t k;   // key bits (words 1 and 2) -- kept as ONE 64-bit value: as three words the compiler assembled the
                  // 64-bit views it needs through a stack slot (store w0, store {w1,w2}, load {w0,w1}: 16 B of scratch)
};
template <int W>
constexpr int rank_cap() { return W == 2 ? 16384 : 13312; }
template <int W>
constexpr int rank_lds_bytes() { return (kRankCtrlWords + W * rank_cap<W>()) * 4; }
static_assert(rank_lds_bytes<3>() <= 160 * 1024 && rank_lds_bytes<2>() <= 160 * 1024, "one workgroup's LDS");

// order-preserving bits: unsigned ascending == numeric ascending; -0.0 == +0.0 as in `beats`
__device__ __forceinline__ uint32_t order_bits(float v) {
    uint32_t b = __float_as_uint(v);
    if ((b << 1) == 0) b = 0;
    return (b & 0x80000000u) ? ~b : (b | 0x80000000u);
}
__device__ __forceinline__ uint64_t order_bits(double v) {
    uint64_t b = (uint64_t)__double_as_longlong(v);
    if ((b << 1) == 0) b = 0;
    return (b >> 63) ? ~b : (b | 0x8000000000000000ull);
}
__device__ __forceinline__ float unorder_bits(uint32_t u) {
    return __uint_as_float((u & 0x80000000u) ? (u ^ 0x80000000u) : ~u);
}
__device__ __forceinline__ double unorder_bits(uint64_t u) {
    return __longlong_as_double((long long)((u >> 63) ? (u ^ 0x8000000000000000ull) : ~u));
}

__device__ __forceinline__ Comp<2> make_comp(float key, uint32_t tie) { return Comp<2>{~tie, order_bits(key)}; }
__device__ __forceinline__ Comp<3> make_comp(double key, uint32_t tie) {
    return Comp<3>{~tie, order_bits(key)};
}
__device__ __forceinline__ Comp<2> comp_zero(Comp<2> *) { return Comp<2>{0, 0}; }
__device__ __forceinline__ Comp<3> comp_zero(Comp<3> *) { return Comp<3>{0, 0}; }
__device__ __forceinline__ uint64_t hi64(const Comp<2> &c) { return ((uint64_t)c.w1 << 32) | c.w0; }
__device__ __forceinline__ uint64_t hi64(const Comp<3> &c) { return c.k; }
__device__ __forceinline__ uint64_t lo64(const Comp<3> &c) { return (c.k << 32) | c.w0; }

__device__ __forceinline__ bool comp_gt(const Comp<2> &a, const Comp<2> &b) { return hi64(a) > hi64(b); }
__device__ __forceinline__ bool comp_gt(const Comp<3> &a, const Comp<3> &b) {
    const uint64_t x = hi64(a), y = hi64(b);
    return x > y || (x == y && a.w0 > b.w0);
}

// bits [lo, lo + len) of the composite, counted from its least significant bit; len <= 11, lo wave-uniform
__device__ __forceinline__ uint32_t comp_bits(const Comp<2> &c, int lo, int len) {
    return (uint32_t)(hi64(c) >> lo) & ((1u << len) - 1u);
}
__device__ __forceinline__ uint32_t comp_bits(const Comp<3> &c, int lo, int len) {
    const uint64_t v = lo >= 32 ? hi64(c) >> (lo - 32) : lo64(c) >> lo;
    return (uint32_t)v & ((1u << len) - 1u);
}
// do the top `pbits` bits of c equal those of pre?
__device__ __forceinline__ bool comp_match(const Comp<2> &c, const Comp<2> &pre, int pbits) {
    return pbits == 0 || ((hi64(c) ^ hi64(pre)) >> (64 - pbits)) == 0;
}
__device__ __forceinline__ bool comp_match(const Comp<3> &c, const Comp<3> &pre, int pbits) {
    if (pbits == 0) return true;
    const uint64_t d = hi64(c) ^ hi64(pre);
    if (pbits <= 64) return (d >> (64 - pbits)) == 0;
    return d == 0 && ((c.w0 ^ pre.w0) >> (96 - pbits)) == 0;
}
__device__ __forceinline__ Comp<2> comp_or_bits(Comp<2> c, uint32_t v, int lo) {
    const uint64_t x = hi64(c) | ((uint64_t)v << lo);
    return Comp<2>{(uint32_t)x, (uint32_t)(x >> 32)};
}
__device__ __forceinline__ Comp<3> comp_or_bits(Comp<3> c, uint32_t v, int lo) {
    if (lo >= 32) {
        return Comp<3>{c.w0, c.k | ((uint64_t)v << (lo - 32))};
    }
    const uint64_t x = lo64(c) | ((uint64_t)v << lo);  // bits 0..63 of the composite: w0 and the low key word
    return Comp<3>{(uint32_t)x, (c.k & 0xFFFFFFFF00000000ull) | (x >> 32)};
}
// survivor i of the LDS arrays (word w of survivor i at arr[w * CAP + i])
template <int CAP>
__device__ __forceinline__ void comp_store(uint32_t *arr, int i, const Comp<2> &c) {
    arr[i] = c.w0;
    arr[CAP + i] = c.w1;
}
template <int CAP>
__device__ __forceinline__ void comp_store(uint32_t *arr, int i, const Comp<3> &c) {
    arr[i] = c.w0;
    arr[CAP + i] = (uint32_t)c.k;
    arr[2 * CAP + i] = (uint32_t)(c.k >> 32);
}
template <int CAP>
__device__ __forceinline__ void comp_load(const uint32_t *arr, int i, Comp<2> &c) {
    c.w0 = arr[i];
    c.w1 = arr[CAP + i];
}
template <int CAP>
__device__ __forceinline__ void comp_load(const uint32_t *arr, int i, Comp<3> &c) {
    c.w0 = arr[i];
    c.k = ((uint64_t)arr[2 * CAP + i] << 32) | arr[CAP + i];
}

// Sort the m composites held in LDS (word w of entry i at arr[w * CAP + i]), best first.  All threads of the workgroup.
template <int W, int CAP>
__device__ __forceinline__ void lds_bitonic_sort(uint32_t *arr, int32_t m) {
    constexpr int T = kRankThreads;
    const int tid = threadIdx.x;
    // bitonic network, best first, every comparator pointing the same way (a merge step starts with a "flip"):
    // positions >= m stand for entries worse than all real ones, and since a comparator only ever moves the better
    // entry DOWN in index they never move -- comparators that touch them are skipped, no padding is stored.
    int32_t v2 = 2;
    while (v2 < m) v2 <<= 1;
    auto cmp_swap = [&](int32_t a, int32_t b) __attribute__((always_inline)) {
        Comp<W> x, y;
        comp_load<CAP>(arr, a, x);
        comp_load<CAP>(arr, b, y);
        if (comp_gt(y, x)) {
            comp_store<CAP>(arr, a, y);
            comp_store<CAP>(arr, b, x);
        }
    };
    const int32_t half_all = v2 >> 1;
    auto flip_stage = [&](int32_t size, int lg_half) __attribute__((always_inline)) {
        const int32_t half = size >> 1;
        for (int32_t c = tid; c < half_all; c += T) {
            const int32_t blk = c >> lg_half, t = c & (half - 1);
            const int32_t a = blk * size + t, b = blk * size + size - 1 - t;
            if (b < m) cmp_swap(a, b);
        }
        __syncthreads();
    };
    auto stride_stage = [&](int32_t stride) __attribute__((always_inline)) {
        for (int32_t c = tid; c < half_all; c += T) {
            const int32_t a = ((c & ~(stride - 1)) << 1) | (c & (stride - 1)), b = a + stride;
            if (b < m) cmp_swap(a, b);
        }
        __syncthreads();
    };
    // The stages whose partners are < 16 apart stay in REGISTERS: a thread takes a block of 16 consecutive entries (four
    // 16-byte LDS reads per word array), runs the stages on them, writes the block back -- one LDS round trip for the
    // four stages (strides 8, 4, 2, 1) that end every merge, and for the whole of the first four merges: 66 instead of
    // 105 rounds at 16,384 entries.  Entries past m are read as the worst composite (all zero) and written back as such:
    // they never move (a comparator only moves the better entry down in index) and nobody else reads them.
    constexpr int LB = 16;
    auto local_round = [&](int32_t from_size) __attribute__((always_inline)) {  // from_size: 2 = the merges of 2..16; 0 = strides 8..1
        const int32_t n_blocks = (m + LB - 1) / LB;
        static_assert(CAP <= LB * T, "one register block per thread covers the array");
        if (const int32_t blk = tid; blk < n_blocks) {  // (a loop here was unrolled by two in one kernel: 38 spills)
            Comp<W> x[LB];
            uint32_t word[W][LB];
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int c = 0; c < LB / 4; ++c) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(&arr[w * CAP + blk * LB + 4 * c]);
                    word[w][4 * c] = v.x;
                    word[w][4 * c + 1] = v.y;
                    word[w][4 * c + 2] = v.z;
                    word[w][4 * c + 3] = v.w;
                }
#pragma unroll
            for (int i = 0; i < LB; ++i) {
                const bool real = blk * LB + i < m;
                if constexpr (W == 2) x[i] = Comp<2>{real ? word[0][i] : 0u, real ? word[1][i] : 0u};
                else x[i] = Comp<3>{real ? word[0][i] : 0u, real ? (((uint64_t)word[2][i] << 32) | word[1][i]) : 0ull};
            }
            auto cx = [&](int a, int b) __attribute__((always_inline)) {
                if (comp_gt(x[b], x[a])) {
                    const Comp<W> t = x[a];
                    x[a] = x[b];
                    x[b] = t;
                }
            };
            if (from_size == 2) {
#pragma unroll
                for (int size = 2; size <= LB; size <<= 1) {
#pragma unroll
                    for (int i = 0; i < LB; ++i)
                        if ((i & (size - 1)) < size / 2) cx(i, (i & ~(size - 1)) + size - 1 - (i & (size - 1)));
#pragma unroll
                    for (int stride = size / 4; stride >= 1; stride >>= 1)
#pragma unroll
                        for (int i = 0; i < LB; ++i)
                            if (!(i & stride)) cx(i, i + stride);
                }
            } else {
#pragma unroll
                for (int stride = LB / 2; stride >= 1; stride >>= 1)
#pragma unroll
                    for (int i = 0; i < LB; ++i)
                        if (!(i & stride)) cx(i, i + stride);
            }
#pragma unroll
            for (int w = 0; w < W; ++w)
#pragma unroll
                for (int c = 0; c < LB / 4; ++c) {
                    uint4 v;
                    auto wd = [&](int i) __attribute__((always_inline)) -> uint32_t {
                        if constexpr (W == 2) return w == 0 ? x[i].w0 : x[i].w1;
                        else return w == 0 ? x[i].w0 : (w == 1 ? (uint32_t)x[i].k : (uint32_t)(x[i].k >> 32));
                    };
                    v.x = wd(4 * c);
                    v.y = wd(4 * c + 1);
                    v.z = wd(4 * c + 2);
                    v.w = wd(4 * c + 3);
                    *reinterpret_cast<uint4 *>(&arr[w * CAP + blk * LB + 4 * c]) = v;
                }
        }
        __syncthreads();
    };
    if (v2 >= 2 * LB) {
        local_round(2);  // every block of 16 sorted
        int lg_half = 4;
        for (int32_t size = 2 * LB; size <= v2; size <<= 1, ++lg_half) {
            flip_stage(size, lg_half);
            for (int32_t stride = size >> 2; stride >= LB; stride >>= 1) stride_stage(stride);
            local_round(0);
        }
    } else {  // a handful of entries: the plain network
        int lg_half = 0;
        for (int32_t size = 2; size <= v2; size <<= 1, ++lg_half) {
            flip_stage(size, lg_half);
            for (int32_t stride = size >> 2; stride >= 1; stride >>= 1) stride_stage(stride);
        }
    }
}

// ------------------------------------------------------------------ select + sort, one workgroup per segment
// keys + seg * key_stride: n scores of the segment (a key of -inf = "not a candidate": a filtered row, a document no
// list names).  TIE: ties + seg * tie_stride holds each element's tie-break word (ascending; unique inside a segment),
// else the tie-break is the element's position (the row).  k: entries wanted (seg_k, when given, replaces it per
// segment: 0 = this segment is skipped, e.g. the BM25 leg of a query without tokens, src/search_engine.py:216-217).
// Out: out_tie[seg * out_stride + i] = tie-break word (row) of rank i, out_key (nullable) its score as fp64,
// out_count[seg] = min(k, candidates).
template <typename KEY, bool TIE>
__global__ __launch_bounds__(kRankThreads) void seg_topk_sort_kernel(
    const KEY *__restrict__ keys, int64_t key_stride, const uint32_t *__restrict__ ties, int64_t tie_stride, int32_t n,
    int32_t k, const int32_t *__restrict__ seg_k, uint32_t *__restrict__ out_tie, double *__restrict__ out_key,
    int64_t out_stride, int32_t *__restrict__ out_count) {
    constexpr int W = sizeof(KEY) == 4 ? 2 : 3;
    constexpr int B = 32 * W;
    constexpr int CAP = rank_cap<W>();
    constexpr int T = kRankThreads;
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_lds[];
    uint32_t *ctrl = rank_lds;  // [0] survivors  [1] candidates  [2] chosen bin  [3] entries above it  [4] entries in it
    uint32_t *arr = rank_lds + kRankCtrlWords;  // word w of survivor i at arr[w * CAP + i]
    uint32_t *hist = arr;                       // during the select passes

    const int tid = threadIdx.x, lane = tid & (kWave - 1), wave = tid / kWave;
    const int64_t seg = blockIdx.x;
    int32_t kq = k;
    if (seg_k) kq = seg_k[seg] < k ? seg_k[seg] : k;
    if (kq <= 0) {
        if (tid == 0) out_count[seg] = 0;
        return;
    }
    const KEY *__restrict__ kp = keys + seg * key_stride;
    const uint32_t *__restrict__ tp = TIE ? ties + seg * tie_stride : nullptr;

    Comp<W> thr = comp_zero((Comp<W> *)nullptr);  // survivors = candidates whose composite is not below thr
    int32_t k_eff = kq;  // min(kq, candidates), known after the first pass (n > CAP) or after the compaction

    // one pass over the segment: f(key, tie word) for every element, kRankLoads independent loads per thread in flight
    auto for_each = [&](auto &&f) __attribute__((always_inline)) {
        for (int32_t base = 0; base < n; base += T * kRankLoads) {
            KEY kv[kRankLoads];
            uint32_t tv[kRankLoads];
#pragma unroll
            for (int u = 0; u < kRankLoads; ++u) {
                const int32_t i = base + u * T + tid;
                kv[u] = i < n ? kp[i] : neg_inf<KEY>();
                tv[u] = (uint32_t)i;
                if constexpr (TIE) tv[u] = i < n ? tp[i] : 0u;
            }
#pragma unroll
            for (int u = 0; u < kRankLoads; ++u) f(kv[u], tv[u]);
        }
    };

    if (n > CAP) {
        // MSD radix select over the composite: after each pass the prefix grows by one digit; `above` counts the
        // candidates above the prefix's range, the bin holds the rest of the k_eff best.
        int pbits = 0;
        uint32_t above = 0;
        int32_t k_rem = 0;
        for (;;) {
            const int len = B - pbits < kRankDigitBits ? B - pbits : kRankDigitBits;
            const int lo = B - pbits - len;
            const int nb = 1 << len;
            for (int i = tid; i < nb; i += T) hist[i] = 0;
            __syncthreads();
            for_each([&](KEY key, uint32_t tie) __attribute__((always_inline)) {
                const Comp<W> c = make_comp(key, tie);
                const bool cand = key != neg_inf<KEY>() && comp_match(c, thr, pbits);
                const uint32_t bin = comp_bits(c, lo, len);
                const unsigned long long m = __ballot(cand);
                if (m) {
                    // heavy ties (a million zero BM25 scores) put a whole wave into one bin: one add instead of 64
                    // serialised ones
                    const int leader = __builtin_ctzll(m);
                    const uint32_t b0 = read_lane(bin, leader);
                    if (__ballot(cand && bin == b0) == m) {
                        if (lane == leader) atomicAdd(&hist[b0], (uint32_t)__builtin_popcountll(m));
                    } else if (cand) {
                        atomicAdd(&hist[bin], 1u);
                    }
                }
            });
            __syncthreads();
            if (wave == 0) {  // the bin that holds the k_rem-th best: lane l looks after bins [l * per, (l+1) * per)
                const int per = nb / kWave;
                uint32_t s = 0;
                for (int b = 0; b < per; ++b) s += hist[lane * per + b];
                uint32_t x = s;  // inclusive suffix sum over the lanes
#pragma unroll
                for (int off = 1; off < kWave; off <<= 1) {
                    const uint32_t y = __shfl_down(x, off, kWave);
                    if (lane + off < kWave) x += y;
                }
                const uint32_t total = read_lane(x, 0);
                int32_t want = k_rem;
                if (pbits == 0) want = (uint32_t)kq < total ? kq : (int32_t)total;  // = k_eff
                if (lane == 0) ctrl[1] = total;
                const uint32_t upper = x - s;  // candidates in the bins of higher lanes
                if (want > 0 && upper < (uint32_t)want && (uint32_t)want <= upper + s) {
                    uint32_t acc = upper;
                    for (int b = per - 1; b >= 0; --b) {
                        const uint32_t h = hist[lane * per + b];
                        if (acc + h >= (uint32_t)want) {
                            ctrl[2] = (uint32_t)(lane * per + b);
                            ctrl[3] = acc;
                            ctrl[4] = h;
                            break;
                        }
                        acc += h;
                    }
                }
            }
            __syncthreads();
            if (pbits == 0) {
                const uint32_t total = ctrl[1];
                k_eff = (uint32_t)kq < total ? kq : (int32_t)total;
                k_rem = k_eff;
                if (k_eff == 0) {
                    if (tid == 0) out_count[seg] = 0;
                    return;
                }
            }
            const uint32_t bin = ctrl[2], acc = ctrl[3], inbin = ctrl[4];
            above += acc;
            k_rem -= (int32_t)acc;
            thr = comp_or_bits(thr, bin, lo);
            pbits += len;
            if (above + inbin <= (uint32_t)CAP || pbits == B) break;
            __syncthreads();  // ctrl / hist are rewritten by the next pass
        }
        __syncthreads();  // the histogram's LDS becomes the survivor arrays
    }

    // compaction: candidates not below thr -> LDS (any order: they are sorted next)
    if (tid == 0) ctrl[0] = 0;
    __syncthreads();
    for_each([&](KEY key, uint32_t tie) __attribute__((always_inline)) {
        const Comp<W> c = make_comp(key, tie);
        const bool keep = key != neg_inf<KEY>() && !comp_gt(thr, c);
        const unsigned long long m = __ballot(keep);
        if (m) {
            uint32_t base = 0;
            if (lane == 0) base = atomicAdd(&ctrl[0], (uint32_t)__builtin_popcountll(m));
            base = read_lane(base, 0);
            if (keep) {
                const uint32_t pos = base + (uint32_t)__builtin_popcountll(m & ((1ull << lane) - 1ull));
                if (pos < (uint32_t)CAP) comp_store<CAP>(arr, (int)pos, c);
            }
        }
    });
    __syncthreads();
    const int32_t m = (int32_t)(ctrl[0] < (uint32_t)CAP ? ctrl[0] : (uint32_t)CAP);
    if (k_eff > m) k_eff = m;  // n <= CAP: everything that is a candidate

    lds_bitonic_sort<W, CAP>(arr, m);

    for (int32_t i = tid; i < k_eff; i += T) {
        out_tie[seg * out_stride + i] = ~arr[i];
        if (out_key) {
            if constexpr (W == 2) out_key[seg * out_stride + i] = (double)unorder_bits(arr[CAP + i]);
            else out_key[seg * out_stride + i] = unorder_bits(((uint64_t)arr[2 * CAP + i] << 32) | arr[CAP + i]);
        }
    }
    if (tid == 0) out_count[seg] = k_eff;
}

// ------------------------------------------------------------------ segments that fit the LDS: stable LSD radix sort
// The bitonic network above costs log^2: 66 LDS round trips for 9,609 entries (the evaluation corpus), 117 us per fp32
// segment, 72 % of the full-ranking route's GPU time.  A leg's segment has a property the network does not use: its entries
// arrive in tie-break order (entry i = row i, ties go to the lower row), so a STABLE sort by descending score alone is the
// whole order -- an LSD radix sort over the key bits only, 8 bits a pass: 4 passes for fp32 scores, 8 for fp64.
//   * wave w owns entries [w * S, (w + 1) * S), S a multiple of 64; lane l of row j holds entry w * S + 64 j + l in
//     REGISTERS for the whole pass, so the scatter is in place (everybody has read before anybody writes);
//   * rank inside the wave, in entry order: per row, the lanes with the same digit find each other with 8 ballots
//     (peers = AND over the digit's bits of "lanes whose bit equals mine"), the lowest of them bumps the wave's counter
//     cnt[digit][wave] by the group's size, everyone takes (counter before + peers below me);
//   * one exclusive scan over cnt in (digit, wave) order turns the counters into the waves' start offsets per digit.
// Keys are the complemented order bits (ascending = best first); a non-candidate (-inf: filtered row) is all ones and
// sorts behind every score.  Same order as the network's by construction (total order on (score desc, row asc)).
constexpr int kRadixBits = 8, kRadixBins = 1 << kRadixBits, kRadixWaves = kRankThreads / kWave;
template <int W>
constexpr int radix_cap() { return W == 2 ? 16384 : 12288; }  // fp64: 3 words an entry + the counters must fit 160 KB
template <int W>
constexpr int radix_lds_bytes() {
    return (kRankCtrlWords + W * radix_cap<W>()) * 4 + kRadixBins * kRadixWaves * 2 + kRadixWaves * 4;
}
static_assert(radix_lds_bytes<2>() <= 160 * 1024 && radix_lds_bytes<3>() <= 160 * 1024, "one workgroup's LDS");

template <typename KEY>
__global__ __launch_bounds__(kRankThreads) void seg_radix_sort_kernel(
    const KEY *__restrict__ keys, int64_t key_stride, int32_t n, int32_t k, const int32_t *__restrict__ seg_k,
    uint32_t *__restrict__ out_tie, double *__restrict__ out_key, int64_t out_stride, int32_t *__restrict__ out_count) {
    constexpr int W = sizeof(KEY) == 4 ? 2 : 3, KW = W - 1;  // KW key words + the row
    constexpr int CAP = radix_cap<W>();
    constexpr int RMAX = CAP / kRankThreads;  // rows of 64 entries per wave
    constexpr int T = kRankThreads, NW = kRadixWaves;
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_lds[];
    uint32_t *ctrl = rank_lds;
    uint32_t *arr = rank_lds + kRankCtrlWords;  // word w of entry i at arr[w * CAP + i]; word 0 = row, 1.. = key
    uint16_t *cnt = reinterpret_cast<uint16_t *>(arr + W * CAP);  // [digit * NW + wave]
    uint32_t *wsum = reinterpret_cast<uint32_t *>(cnt + kRadixBins * NW);

    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int wave = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int64_t seg = blockIdx.x;
    int32_t kq = k;
    if (seg_k) kq = seg_k[seg] < k ? seg_k[seg] : k;
    if (kq <= 0) {
        if (tid == 0) out_count[seg] = 0;
        return;
    }
    const KEY *__restrict__ kp = keys + seg * key_stride;
    const int32_t S = ((n + NW - 1) / NW + kWave - 1) / kWave * kWave;  // entries per wave
    const int32_t R = S / kWave;                                       // rows per wave, <= RMAX
    const int32_t e0 = wave * S + lane;

    uint32_t k0[RMAX], k1[W == 3 ? RMAX : 1], row[RMAX];  // (separate arrays: a [row][word] array went to scratch)
    uint32_t n_cand = 0;
    if (tid == 0) ctrl[0] = 0;
#pragma unroll
    for (int j = 0; j < RMAX; ++j) {
        if (j < R) {
            const int32_t e = e0 + j * kWave;
            const KEY v = e < n ? kp[e] : neg_inf<KEY>();
            const bool cand = v != neg_inf<KEY>();
            n_cand += (uint32_t)__builtin_popcountll(__ballot(cand));
            row[j] = (uint32_t)e;
            if constexpr (W == 2) {
                k0[j] = cand ? ~order_bits(v) : 0xFFFFFFFFu;
            } else {
                const uint64_t b = cand ? ~order_bits(v) : ~0ull;
                k0[j] = (uint32_t)b;
                k1[j] = (uint32_t)(b >> 32);
            }
        }
    }
    __syncthreads();
    if (lane == 0 && n_cand) atomicAdd(&ctrl[0], n_cand);

    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    for (int pass = 0; pass < KW * 4; ++pass) {
        const int sh = (pass & 3) * kRadixBits;
        const bool hi = pass >= 4;  // fp64: the upper key word
        for (int i = tid; i < kRadixBins * NW / 2; i += T) reinterpret_cast<uint32_t *>(cnt)[i] = 0;
        __syncthreads();
        uint32_t rank[RMAX], dig[RMAX];
#pragma unroll
        for (int j = 0; j < RMAX; ++j) {
            if (j < R) {
                uint32_t word = k0[j];
                if constexpr (W == 3) word = hi ? k1[j] : k0[j];
                const uint32_t d = (word >> sh) & (kRadixBins - 1);
                const bool valid = e0 + j * kWave < S * NW;  // (always: every lane of a row holds an entry, real or padding)
                unsigned long long peers = __ballot(valid);
#pragma unroll
                for (int b = 0; b < kRadixBits; ++b) {
                    const bool bit = (d >> b) & 1u;
                    const unsigned long long bal = __ballot(bit);
                    peers &= bit ? bal : ~bal;
                }
                const uint32_t below = (uint32_t)__builtin_popcountll(peers & lt_mask);
                const uint32_t group = (uint32_t)__builtin_popcountll(peers);
                uint16_t *c = &cnt[d * NW + wave];
                const uint32_t before = *c;
                if (below == 0) *c = (uint16_t)(before + group);  // the group's lowest lane
                rank[j] = before + below;
                dig[j] = d;
            }
        }
        __syncthreads();
        {  // exclusive scan of cnt in index order: 4 counters a thread
            const uint2 v = reinterpret_cast<const uint2 *>(cnt)[tid];
            const uint32_t c0 = v.x & 0xFFFFu, c1 = v.x >> 16, c2 = v.y & 0xFFFFu, c3 = v.y >> 16;
            const uint32_t s4 = c0 + c1 + c2 + c3;
            uint32_t x = s4;  // inclusive scan over the wave's lanes
#pragma unroll
            for (int off = 1; off < kWave; off <<= 1) {
                const uint32_t y = __shfl_up(x, off, kWave);
                if (lane >= off) x += y;
            }
            if (lane == kWave - 1) wsum[wave] = x;
            __syncthreads();
            uint32_t base = 0;
#pragma unroll
            for (int w2 = 0; w2 < NW; ++w2) base += w2 < wave ? wsum[w2] : 0u;
            const uint32_t ex = base + x - s4;
            uint2 o;
            o.x = (ex & 0xFFFFu) | ((ex + c0) << 16);
            o.y = ((ex + c0 + c1) & 0xFFFFu) | ((ex + c0 + c1 + c2) << 16);
            reinterpret_cast<uint2 *>(cnt)[tid] = o;
        }
        __syncthreads();
#pragma unroll
        for (int j = 0; j < RMAX; ++j) {
            if (j < R) {
                const uint32_t dest = (uint32_t)cnt[dig[j] * NW + wave] + rank[j];
                arr[dest] = row[j];
                arr[CAP + dest] = k0[j];
                if constexpr (W == 3) arr[2 * CAP + dest] = k1[j];
            }
        }
        __syncthreads();
        if (pass + 1 < KW * 4) {
#pragma unroll
            for (int j = 0; j < RMAX; ++j) {
                if (j < R) {
                    const int32_t e = e0 + j * kWave;
                    row[j] = arr[e];
                    k0[j] = arr[CAP + e];
                    if constexpr (W == 3) k1[j] = arr[2 * CAP + e];
                }
            }
        }
    }
    const int32_t cand_all = (int32_t)ctrl[0];
    const int32_t k_eff = kq < cand_all ? kq : cand_all;
    for (int32_t i = tid; i < k_eff; i += T) {
        out_tie[seg * out_stride + i] = arr[i];
        if (out_key) {
            if constexpr (W == 2) out_key[seg * out_stride + i] = (double)unorder_bits(~arr[CAP + i]);
            else out_key[seg * out_stride + i] = unorder_bits(~(((uint64_t)arr[2 * CAP + i] << 32) | arr[CAP + i]));
        }
    }
    if (tid == 0) out_count[seg] = k_eff;
}

// ------------------------------------------------------------------ fusion over row lists
__global__ void rank_fill_kernel(double *__restrict__ f, uint32_t *__restrict__ t, int64_t total) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    f[i] = -__builtin_huge_val();
    t[i] = 0xFFFFFFFFu;
}

// One leg's contributions for every query of the chunk: grid (ceil(list_len / 256), queries).  A leg names a document
// at most once per query (checked on the host: the row -> document map is injective), so no two threads of a launch
// touch the same F entry, and the launches of the legs follow each other on the stream in leg order.
__global__ __launch_bounds__(256) void rank_accumulate_kernel(const uint32_t *__restrict__ rows, int64_t rows_stride,
                                                              const int32_t *__restrict__ cnt,
                                                              const int32_t *__restrict__ doc_of_row, uint32_t leg,
                                                              double weight, double wrrf_k, int64_t id_space,
                                                              double *__restrict__ f, uint32_t *__restrict__ t) {
    const int64_t q = blockIdx.y;
    const int32_t p = blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= cnt[q]) return;
    const uint32_t row = rows[q * rows_stride + p];
    const int64_t d = doc_of_row ? (int64_t)doc_of_row[row] : (int64_t)row;
    const double c = weight * (1.0 / (wrrf_k + (double)(p + 1)));  // src/search_engine.py:30, rank from 1
    const int64_t at = q * id_space + d;
    if (t[at] == 0xFFFFFFFFu) {
        f[at] = 0.0 + c;  // `rrf_scores[doc_id] = 0`, then `+=`
        t[at] = (leg << kTiePosBits) | (uint32_t)p;
    } else {
        f[at] = f[at] + c;
    }
}

struct RankLegsDev {
    const uint32_t *rows[ANRAG_WRRF_MAX_LISTS];
    int64_t rows_stride[ANRAG_WRRF_MAX_LISTS];
    const int32_t *doc_of_row[ANRAG_WRRF_MAX_LISTS];
};

// fused order -> document ids: the insertion key names (leg, position); that leg's list names the row
// (expect, when given: out_rank[q] = 1-based position of document expect[q] in the answer -- retrieval_eval.py:75-82's
// search of the returned list -- pre-set to -1 by the host)
__global__ __launch_bounds__(256) void rank_emit_fused_kernel(RankLegsDev L, const uint32_t *__restrict__ s_tie,
                                                              const double *__restrict__ s_key, int64_t s_stride,
                                                              const int32_t *__restrict__ s_cnt, int32_t top_n,
                                                              int64_t *__restrict__ out_id,
                                                              double *__restrict__ out_score,
                                                              const int64_t *__restrict__ expect,
                                                              int32_t *__restrict__ out_rank) {
    const int64_t q = blockIdx.y;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= top_n) return;
    int64_t id = -1;
    double sc = -__builtin_huge_val();
    if (i < s_cnt[q]) {
        const uint32_t tie = s_tie[q * s_stride + i];
        const uint32_t leg = tie >> kTiePosBits, p = tie & ((1u << kTiePosBits) - 1u);
        const uint32_t row = L.rows[leg][q * L.rows_stride[leg] + p];
        id = L.doc_of_row[leg] ? (int64_t)L.doc_of_row[leg][row] : (int64_t)row;
        sc = s_key[q * s_stride + i];
        if (expect && id == expect[q]) out_rank[q] = i + 1;
    }
    out_id[q * top_n + i] = id;
    out_score[q * top_n + i] = sc;
}

// one leg only: its list IS the answer (src/query_rag_retrieval.py:363-366)
__global__ __launch_bounds__(256) void rank_emit_single_kernel(const uint32_t *__restrict__ rows,
                                                               const double *__restrict__ keys, int64_t stride,
                                                               const int32_t *__restrict__ cnt,
                                                               const int32_t *__restrict__ doc_of_row, int32_t top_n,
                                                               int64_t *__restrict__ out_id,
                                                               double *__restrict__ out_score,
                                                               int32_t *__restrict__ out_cnt,
                                                               const int64_t *__restrict__ expect,
                                                               int32_t *__restrict__ out_rank) {
    const int64_t q = blockIdx.y;
    const int32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= top_n) return;
    int64_t id = -1;
    double sc = -__builtin_huge_val();
    const int32_t c = cnt[q] < top_n ? cnt[q] : top_n;
    if (i < c) {
        const uint32_t row = rows[q * stride + i];
        id = doc_of_row ? (int64_t)doc_of_row[row] : (int64_t)row;
        sc = keys[q * stride + i];
        if (expect && id == expect[q]) out_rank[q] = i + 1;
    }
    out_id[q * top_n + i] = id;
    out_score[q * top_n + i] = sc;
    if (i == 0) out_cnt[q] = c;
}

// One leg, and only the rank of the expected document is wanted (src/retrieval_eval.py:75-82 with a single list): no
// select, no sort -- the rank is one more than the number of rows that beat the expected row under the list's order (score
// desc, row asc), and the list's length is the number of candidates cut to k.  One workgroup per query, two passes over
// the score segment (the first finds the expected document's row when the leg maps rows to documents).
template <typename KEY>
__global__ __launch_bounds__(kRankThreads) void rank_count_kernel(const KEY *__restrict__ keys, int64_t key_stride, int32_t n,
                                                                  int32_t k, const int32_t *__restrict__ seg_k,
                                                                  const int32_t *__restrict__ doc_of_row, int32_t top_n,
                                                                  const int64_t *__restrict__ expect,
                                                                  int32_t *__restrict__ out_rank,
                                                                  int32_t *__restrict__ out_cnt) {
    constexpr int T = kRankThreads;
    __shared__ uint32_t ctrl[4];  // [0] candidates  [1] rows that beat the expected one  [2] its row + 1
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int64_t q = blockIdx.x;
    int32_t kq = k;
    if (seg_k) kq = seg_k[q] < k ? seg_k[q] : k;
    if (kq <= 0) {
        if (tid == 0) {
            out_cnt[q] = 0;
            out_rank[q] = -1;
        }
        return;
    }
    const KEY *__restrict__ kp = keys + q * key_stride;
    const int64_t e = expect[q];
    if (tid < 4) ctrl[tid] = 0;
    __syncthreads();
    if (doc_of_row) {  // (a leg names a document once: at most one thread finds it)
        for (int32_t i = tid; i < n; i += T)
            if ((int64_t)doc_of_row[i] == e) ctrl[2] = (uint32_t)i + 1u;
    } else if (tid == 0 && e >= 0 && e < n) {
        ctrl[2] = (uint32_t)e + 1u;
    }
    __syncthreads();
    const int32_t r = (int32_t)ctrl[2] - 1;
    const KEY ke = r >= 0 ? kp[r] : neg_inf<KEY>();
    const bool listed = r >= 0 && ke != neg_inf<KEY>();
    const auto be = order_bits(ke);
    uint32_t cand = 0, better = 0;
    for (int32_t base = 0; base < n; base += T * kRankLoads) {
        KEY kv[kRankLoads];
#pragma unroll
        for (int u = 0; u < kRankLoads; ++u) {
            const int32_t i = base + u * T + tid;
            kv[u] = i < n ? kp[i] : neg_inf<KEY>();
        }
#pragma unroll
        for (int u = 0; u < kRankLoads; ++u) {
            const int32_t i = base + u * T + tid;
            const bool c = kv[u] != neg_inf<KEY>();
            const auto b = order_bits(kv[u]);
            cand += c ? 1u : 0u;
            better += c && (b > be || (b == be && i < r)) ? 1u : 0u;
        }
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) {
        cand += __shfl_xor(cand, off, kWave);
        better += __shfl_xor(better, off, kWave);
    }
    if (lane == 0) {
        if (cand) atomicAdd(&ctrl[0], cand);
        if (better) atomicAdd(&ctrl[1], better);
    }
    __syncthreads();
    if (tid == 0) {
        const int32_t total = (int32_t)ctrl[0];
        int32_t c = kq < total ? kq : total;
        if (c > top_n) c = top_n;
        const int32_t rank = (int32_t)ctrl[1] + 1;
        out_cnt[q] = c;
        out_rank[q] = listed && rank <= c ? rank : -1;
    }
}

// Fusion of an id space that fits the LDS, ONE workgroup per query, one launch for everything behind the legs' lists:
// the fp64 sums and the insertion keys live in LDS while the legs are added in order (a barrier between legs: a leg names
// a document once, so its adds do not collide, and legs follow each other = the reference's dict update order), are
// turned into composites in place, sorted by the network above, and the winners are mapped back to document ids -- no
// F / T arrays in HBM, no fill, accumulate or emit launches (they were 0.7 of the 6.3 us per query at 9,609 documents).

// (the leg table is read from HBM as a flat array of 64-bit words, field f of leg l at tab[f * 16 + l]: it is indexed per
// lane in the emit, and a struct -- by value or behind a pointer -- was copied to scratch for that: 1.2 KB, 2,087 spills)
__global__ __launch_bounds__(kRankThreads) void rank_fuse_sort_kernel(const uint64_t *tab, double wrrf_k, int32_t id_space,
                                                                      int32_t out_n, int64_t *__restrict__ out_id,
                                                                      double *__restrict__ out_score,
                                                                      int32_t *__restrict__ out_cnt,
                                                                      const int64_t *__restrict__ expect,
                                                                      int32_t *__restrict__ out_rank) {
    constexpr int W = 3, CAP = rank_cap<3>(), T = kRankThreads;
    extern __shared__ __attribute__((aligned(16))) uint32_t rank_lds[];
    uint32_t *ctrl = rank_lds;
    uint32_t *arr = rank_lds + kRankCtrlWords;
    uint32_t *tie = arr;  // word 0 of the composites: ~(insertion key), 0 = no list names it
    // while the legs are added, words 1 and 2 hold the raw bits of the fp64 sums (already in the composites' layout, so
    // the turn to order-preserving key bits is element-wise and in place)
    auto get_sum = [&](int32_t d) __attribute__((always_inline)) {
        return __longlong_as_double((long long)(((uint64_t)arr[2 * CAP + d] << 32) | arr[CAP + d]));
    };
    auto put_bits = [&](int32_t d, uint64_t b) __attribute__((always_inline)) {
        arr[CAP + d] = (uint32_t)b;
        arr[2 * CAP + d] = (uint32_t)(b >> 32);
    };
    const int tid = threadIdx.x, lane = tid & (kWave - 1);
    const int64_t q = blockIdx.x;
    constexpr int M = ANRAG_WRRF_MAX_LISTS;
    auto leg_rows = [&](uint32_t l) __attribute__((always_inline)) { return reinterpret_cast<const uint32_t *>(tab[l]) + q * (int64_t)tab[3 * M + l]; };
    auto leg_map = [&](uint32_t l) __attribute__((always_inline)) { return reinterpret_cast<const int32_t *>(tab[2 * M + l]); };
    for (int32_t i = tid; i < id_space; i += T) tie[i] = 0u;
    if (tid == 0) ctrl[0] = 0;
    __syncthreads();
    const int n_legs = (int)tab[5 * M];
#pragma unroll 1
    for (int l = 0; l < n_legs; ++l) {
        const int32_t n = reinterpret_cast<const int32_t *>(tab[M + l])[q];
        const uint32_t *rows = leg_rows(l);
        const int32_t *map = leg_map(l);
        const double w = __longlong_as_double((long long)tab[4 * M + l]);
        for (int32_t p = tid; p < n; p += T) {
            const uint32_t row = rows[p];
            const int32_t d = map ? map[row] : (int32_t)row;
            const double c = w * (1.0 / (wrrf_k + (double)(p + 1)));  // src/search_engine.py:30, rank from 1
            if (tie[d] == 0u) {
                put_bits(d, (uint64_t)__double_as_longlong(0.0 + c));  // `rrf_scores[doc_id] = 0`, then `+=`
                tie[d] = ~(((uint32_t)l << kTiePosBits) | (uint32_t)p);
            } else {
                put_bits(d, (uint64_t)__double_as_longlong(get_sum(d) + c));
            }
        }
        __syncthreads();
    }
    // sums -> order-preserving key bits; a document no list names becomes the all-zero composite, the worst there is
    int32_t named = 0;
    for (int32_t i = tid; i < id_space; i += T) {
        uint64_t b = 0;
        if (tie[i] != 0u) {
            b = order_bits(get_sum(i));
            ++named;
        }
        put_bits(i, b);
    }
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) named += __shfl_xor(named, off, kWave);
    if (lane == 0 && named) atomicAdd(&ctrl[0], (uint32_t)named);
    __syncthreads();
    const int32_t qe = (int32_t)blockIdx.x, out_ne = out_n;
    const int32_t n_named = (int32_t)ctrl[0];
    const int32_t k_out = n_named < out_ne ? n_named : out_ne;
    if (out_id == nullptr) {
        // Only the rank of the expected document is wanted (src/retrieval_eval.py:75-82 looks it up in the list): its
        // rank is one more than the number of documents whose composite beats its own -- a count, no sort.
        const int64_t e = expect[qe];
        const bool known = e >= 0 && e < id_space && tie[e < id_space && e >= 0 ? e : 0] != 0u;
        Comp<W> ce = comp_zero((Comp<W> *)nullptr);
        if (known) comp_load<CAP>(arr, (int)e, ce);
        uint32_t better = 0;
        for (int32_t i = tid; i < id_space; i += T) {
            Comp<W> c;
            comp_load<CAP>(arr, i, c);
            better += comp_gt(c, ce) ? 1u : 0u;
        }
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) better += __shfl_xor(better, off, kWave);
        if (tid == 0) ctrl[1] = 0;
        __syncthreads();
        if (lane == 0 && better) atomicAdd(&ctrl[1], better);
        __syncthreads();
        if (tid == 0) {
            const int32_t r = (int32_t)ctrl[1] + 1;
            out_rank[qe] = known && r <= k_out ? r : -1;
            out_cnt[qe] = k_out;
        }
        return;
    }
    lds_bitonic_sort<W, CAP>(arr, id_space);
    for (int32_t i = tid; i < out_ne; i += T) {
        int64_t id = -1;
        double sc = -__builtin_huge_val();
        if (i < k_out) {
            const uint32_t t = ~arr[i];
            const uint32_t l = t >> kTiePosBits, p = t & ((1u << kTiePosBits) - 1u);
            const uint32_t row = (reinterpret_cast<const uint32_t *>(tab[l]) + (int64_t)qe * (int64_t)tab[3 * M + l])[p];
            const int32_t *map = leg_map(l);
            id = map ? (int64_t)map[row] : (int64_t)row;
            sc = unorder_bits(((uint64_t)arr[2 * CAP + i] << 32) | arr[CAP + i]);
            if (expect && id == expect[qe]) out_rank[qe] = i + 1;
        }
        out_id[(int64_t)qe * out_ne + i] = id;
        out_score[(int64_t)qe * out_ne + i] = sc;
    }
    if (tid == 0) out_cnt[qe] = k_out;
}

template <typename KEY, bool TIE>
static int launch_seg_sort(int device, hipStream_t st, int32_t n_seg, const KEY *keys, int64_t key_stride,
                           const uint32_t *ties, int64_t tie_stride, int32_t n, int32_t k, const int32_t *seg_k,
                           uint32_t *out_tie, double *out_key, int64_t out_stride, int32_t *out_count) {
    constexpr int W = sizeof(KEY) == 4 ? 2 : 3;
    int rc;
    if constexpr (!TIE) {
        static const bool bitonic = getenv("ANRAG_RANK_BITONIC") != nullptr;  // measurements: the network for every segment
        if (n <= radix_cap<W>() && !bitonic) {  // the segment fits the LDS: stable radix sort, no select
            if ((rc = ensure_dynamic_lds(device, reinterpret_cast<const void *>(&seg_radix_sort_kernel<KEY>),
                                         radix_lds_bytes<W>())))
                return rc;
            seg_radix_sort_kernel<KEY><<<n_seg, kRankThreads, radix_lds_bytes<W>(), st>>>(keys, key_stride, n, k, seg_k, out_tie,
                                                                                         out_key, out_stride, out_count);
            ANRAG_HIP(hipGetLastError());
            return ANRAG_OK;
        }
    }
    rc = ensure_dynamic_lds(device, reinterpret_cast<const void *>(&seg_topk_sort_kernel<KEY, TIE>), rank_lds_bytes<W>());
    if (rc) return rc;
    seg_topk_sort_kernel<KEY, TIE><<<n_seg, kRankThreads, rank_lds_bytes<W>(), st>>>(
        keys, key_stride, ties, tie_stride, n, k, seg_k, out_tie, out_key, out_stride, out_count);
    ANRAG_HIP(hipGetLastError());
    return ANRAG_OK;
}

}  // namespace anrag

using namespace anrag;

extern "C" int anrag_rank_caps(int32_t *out_k_max_fp32, int32_t *out_k_max_fp64) {
    if (out_k_max_fp32) *out_k_max_fp32 = rank_cap<2>();
    if (out_k_max_fp64) *out_k_max_fp64 = rank_cap<3>();
    return ANRAG_OK;
}

extern "C" int anrag_rank_batch(const anrag_rank_leg *legs, int32_t n_legs, int32_t n_queries, int32_t similarity_k,
                                double wrrf_k, int32_t top_n, int64_t id_space, int64_t *out_id, double *out_score,
                                int32_t *out_count, const int64_t *expect_id, int32_t *out_rank) {
    ANRAG_REQUIRE(legs != nullptr && n_legs >= 1 && n_legs <= ANRAG_WRRF_MAX_LISTS, "n_legs %d out of range [1, %d]", n_legs,
                  ANRAG_WRRF_MAX_LISTS);
    ANRAG_REQUIRE(n_queries >= 0 && n_queries <= (1 << 22), "n_queries %d out of range", n_queries);
    ANRAG_REQUIRE(similarity_k > 0 && top_n > 0, "similarity_k and top_n must be positive");
    ANRAG_REQUIRE(out_count != nullptr, "out_count is NULL");
    ANRAG_REQUIRE(!out_score || out_id, "out_score without out_id");
    ANRAG_REQUIRE((expect_id != nullptr) == (out_rank != nullptr), "expect_id and out_rank come together");
    ANRAG_REQUIRE(out_id || out_rank, "nothing to return: out_id and out_rank are both NULL");
    if (n_queries == 0) return ANRAG_OK;
    const bool fuse = n_legs > 1;
    int device = -1;
    int64_t leg_rows[ANRAG_WRRF_MAX_LISTS];
    int32_t leg_k[ANRAG_WRRF_MAX_LISTS];
    int64_t sum_k = 0;
    for (int l = 0; l < n_legs; ++l) {
        const anrag_rank_leg &g = legs[l];
        ANRAG_REQUIRE(g.idx != nullptr, "leg %d: index handle is NULL", l);
        ANRAG_REQUIRE(g.kind == ANRAG_LEG_DENSE || g.kind == ANRAG_LEG_BM25, "leg %d: kind %d", l, g.kind);
        ANRAG_REQUIRE(g.weight > 0.0 && g.weight < __builtin_huge_val(), "leg %d: weight must be positive and finite", l);
        if (device < 0) device = g.idx->device;
        ANRAG_REQUIRE(g.idx->device == device, "leg %d lives on device %d, leg 0 on device %d", l, g.idx->device, device);
    }
    // every distinct index, locked in address order (two callers with the same legs in another order cannot deadlock)
    std::vector<anrag_index *> owners;
    for (int l = 0; l < n_legs; ++l)
        if (std::find(owners.begin(), owners.end(), legs[l].idx) == owners.end()) owners.push_back(legs[l].idx);
    std::sort(owners.begin(), owners.end());
    std::vector<std::unique_lock<std::mutex>> locks;
    for (anrag_index *o : owners) locks.emplace_back(o->mu);
    DeviceGuard guard(device);
    if (!guard.ok) {
        set_error("hipSetDevice(%d) failed", device);
        return ANRAG_ERR_HIP;
    }
    for (int l = 0; l < n_legs; ++l) {
        const anrag_rank_leg &g = legs[l];
        if (g.kind == ANRAG_LEG_DENSE) {
            ANRAG_REQUIRE(g.idx->d_emb != nullptr, "leg %d: dense ranking before anrag_dense_load", l);
            ANRAG_REQUIRE(g.queries != nullptr, "leg %d: queries is NULL", l);
            ANRAG_REQUIRE(!(g.allow_source && !g.idx->d_dense_src), "leg %d: a source filter needs source ids", l);
            leg_rows[l] = g.idx->n_rows;
        } else {
            ANRAG_REQUIRE(g.idx->d_post_doc != nullptr, "leg %d: BM25 ranking before anrag_bm25_load", l);
            ANRAG_REQUIRE(g.term_offsets != nullptr && g.term_offsets[0] == 0, "leg %d: bad term offsets", l);
            ANRAG_REQUIRE(g.term_offsets[n_queries] == 0 || g.term_ids != nullptr, "leg %d: term_ids is NULL", l);
            for (int32_t q = 0; q < n_queries; ++q)
                ANRAG_REQUIRE(g.term_offsets[q + 1] >= g.term_offsets[q] && g.term_offsets[q + 1] - g.term_offsets[q] <= 4096,
                              "leg %d, query %d: term count out of range [0, 4096]", l, q);
            ANRAG_REQUIRE(!(g.allow_source && !g.idx->d_bm25_src), "leg %d: a source filter needs source ids", l);
            leg_rows[l] = g.idx->n_docs;
        }
        ANRAG_REQUIRE(leg_rows[l] < 0x7FFFFFFFll, "leg %d: %lld rows (the batched ranking takes < 2^31)", l,
                      (long long)leg_rows[l]);
        ANRAG_REQUIRE(!g.allow_source || (g.n_sources >= 0 && g.n_sources <= 65536), "leg %d: n_sources out of range", l);
        leg_k[l] = (int32_t)std::min<int64_t>(similarity_k, leg_rows[l]);
        const int32_t cap = g.kind == ANRAG_LEG_DENSE ? rank_cap<2>() : rank_cap<3>();
        ANRAG_REQUIRE(leg_k[l] <= cap, "leg %d: min(similarity_k, rows) = %d is outside the batched ranking's envelope (%d)",
                      l, leg_k[l], cap);
        ANRAG_REQUIRE(leg_k[l] < (1 << kTiePosBits), "leg %d: list too long", l);
        sum_k += leg_k[l];
    }
    bool fuse_in_lds = false;  // the id space fits one workgroup's LDS: rank_fuse_sort_kernel does the whole fusion
    int32_t fuse_n = 0;  // entries the fused order is cut to
    if (fuse) {
        ANRAG_REQUIRE(id_space > 0 && id_space < 0x7FFFFFFFll, "id_space %lld out of range", (long long)id_space);
        fuse_n = (int32_t)std::min<int64_t>(std::min<int64_t>(top_n, sum_k), id_space);
        ANRAG_REQUIRE(fuse_n <= rank_cap<3>(), "min(top_n, entries) = %d is outside the batched ranking's envelope (%d)",
                      fuse_n, rank_cap<3>());
        fuse_in_lds = id_space <= rank_cap<3>() && !getenv("ANRAG_RANK_FUSE_IN_HBM");
        // a leg must name a document once and inside the id space: the additions of a document then happen in leg
        // order whatever the thread schedule (rank_accumulate_kernel)
        std::vector<uint64_t> seen((size_t)((id_space + 63) / 64));
        for (int l = 0; l < n_legs; ++l) {
            const int64_t *map = legs[l].doc_of_row;
            if (!map) {  // row r is document r: injective as it stands
                ANRAG_REQUIRE(leg_rows[l] <= id_space, "leg %d: %lld rows, documents [0, %lld)", l, (long long)leg_rows[l],
                              (long long)id_space);
                continue;
            }
            std::fill(seen.begin(), seen.end(), 0ull);
            for (int64_t r = 0; r < leg_rows[l]; ++r) {
                const int64_t d = map ? map[r] : r;
                ANRAG_REQUIRE(d >= 0 && d < id_space, "leg %d: row %lld maps to document %lld outside [0, %lld)", l,
                              (long long)r, (long long)d, (long long)id_space);
                ANRAG_REQUIRE(!((seen[(size_t)(d >> 6)] >> (d & 63)) & 1ull),
                              "leg %d: document %lld is named by two rows (the batched fusion needs one row per document)",
                              l, (long long)d);
                seen[(size_t)(d >> 6)] |= 1ull << (d & 63);
            }
        }
    }
    for (anrag_index *o : owners)
        if (int rc = settle_pipeline(o)) return rc;

    anrag_index *own = legs[0].idx;  // scratch and stream of the call
    hipStream_t st = own->primary;
    // ---- sizes: what one query of a chunk needs, and the per-call pieces
    const int32_t out_n = fuse ? fuse_n : (int32_t)std::min<int64_t>(top_n, leg_k[0]);
    int64_t per_query = 0, fixed = 0;
    for (int l = 0; l < n_legs; ++l) {
        const bool dense = legs[l].kind == ANRAG_LEG_DENSE;
        const int64_t stride = (leg_rows[l] + 63) / 64 * 64;
        per_query += stride * (dense ? 4 : 8) + (int64_t)leg_k[l] * 4 + 256 + 4 + 4;
        if (!fuse) per_query += (int64_t)leg_k[l] * 8 + 256;
        per_query += dense ? (int64_t)legs[l].idx->dim * 4 : 8;  // BM25: the query's offset into the chunk's terms
        fixed += 2048 * 4 + 256 + 8 + 256 + (legs[l].doc_of_row ? leg_rows[l] * 4 + 256 : 0);
    }
    if (fuse && !fuse_in_lds) per_query += id_space * 12 + 512 + (int64_t)fuse_n * 12 + 512 + 4;
    if (fuse) per_query += 4 + 256;
    fixed += 1024;
    per_query += (int64_t)out_n * 16 + 4 + 512 + 16;
    int64_t max_terms_chunk = 0;  // sized below, once the chunk is known
    const int64_t budget = 6ll << 30;
    int64_t chunk = std::max<int64_t>(1, std::min<int64_t>(n_queries, (budget - fixed) / std::max<int64_t>(per_query, 1)));
    if (chunk > 8) chunk = chunk / 8 * 8;
    for (int l = 0; l < n_legs; ++l)
        if (legs[l].kind == ANRAG_LEG_BM25)
            for (int64_t q0 = 0; q0 < n_queries; q0 += chunk) {
                const int64_t q1 = std::min<int64_t>(n_queries, q0 + chunk);
                max_terms_chunk = std::max(max_terms_chunk, legs[l].term_offsets[q1] - legs[l].term_offsets[q0]);
            }
    const int64_t need = fixed + chunk * per_query + n_legs * (max_terms_chunk * 4 + 256) + (1 << 16);
    int rc;
    if ((rc = ensure_pool(own, own->rank_pool, need))) return rc;

    Carver cv(own->rank_pool.p);
    struct LegDev {
        uint32_t *allow = nullptr;
        int32_t *doc_of_row = nullptr;
        float *q = nullptr;
        int32_t *terms = nullptr;
        int32_t *seg_k = nullptr;
        int64_t *term_off = nullptr;
        void *scores = nullptr;
        int64_t stride = 0;
        uint32_t *rows = nullptr;
        double *row_keys = nullptr;
        int32_t *cnt = nullptr;
    } ld[ANRAG_WRRF_MAX_LISTS];
    std::vector<uint32_t> h_bits(2048);
    std::vector<int32_t> h_map;
    for (int l = 0; l < n_legs; ++l) {
        const anrag_rank_leg &g = legs[l];
        const bool dense = g.kind == ANRAG_LEG_DENSE;
        ld[l].stride = (leg_rows[l] + 63) / 64 * 64;
        if (g.allow_source) {
            ld[l].allow = cv.take<uint32_t>(2048);
            std::fill(h_bits.begin(), h_bits.end(), 0u);
            for (int32_t s = 0; s < g.n_sources; ++s)
                if (g.allow_source[s]) h_bits[s >> 5] |= 1u << (s & 31);
            ANRAG_HIP(hipMemcpyAsync(ld[l].allow, h_bits.data(), 2048 * 4, hipMemcpyHostToDevice, st));
            ANRAG_HIP(hipStreamSynchronize(st));  // h_bits is reused by the next leg
        }
        if (g.doc_of_row) {
            ld[l].doc_of_row = cv.take<int32_t>(leg_rows[l]);
            h_map.resize((size_t)leg_rows[l]);
            for (int64_t r = 0; r < leg_rows[l]; ++r) h_map[(size_t)r] = (int32_t)g.doc_of_row[r];
            ANRAG_HIP(hipMemcpyAsync(ld[l].doc_of_row, h_map.data(), (size_t)leg_rows[l] * 4, hipMemcpyHostToDevice, st));
            ANRAG_HIP(hipStreamSynchronize(st));
        }
        if (dense) {
            ld[l].q = cv.take<float>(chunk * g.idx->dim);
            ld[l].scores = cv.take<float>(chunk * ld[l].stride);
        } else {
            ld[l].terms = cv.take<int32_t>(std::max<int64_t>(max_terms_chunk, 1));
            ld[l].seg_k = cv.take<int32_t>(chunk);
            ld[l].term_off = cv.take<int64_t>(chunk + 1);
            ld[l].scores = cv.take<double>(chunk * ld[l].stride);
        }
        ld[l].rows = cv.take<uint32_t>(chunk * leg_k[l]);
        ld[l].cnt = cv.take<int32_t>(chunk);
        if (!fuse) ld[l].row_keys = cv.take<double>(chunk * leg_k[l]);
    }
    double *d_f = nullptr, *d_skey = nullptr;
    uint32_t *d_t = nullptr, *d_stie = nullptr;
    int32_t *d_scnt = nullptr;
    if (fuse && !fuse_in_lds) {
        d_f = cv.take<double>(chunk * id_space);
        d_t = cv.take<uint32_t>(chunk * id_space);
        d_stie = cv.take<uint32_t>(chunk * fuse_n);
        d_skey = cv.take<double>(chunk * fuse_n);
    }
    if (fuse) d_scnt = cv.take<int32_t>(chunk);
    uint64_t *d_fuse_legs = fuse_in_lds ? cv.take<uint64_t>(6 * ANRAG_WRRF_MAX_LISTS) : nullptr;
    int64_t *d_out_id = cv.take<int64_t>(chunk * out_n);
    double *d_out_score = cv.take<double>(chunk * out_n);
    int32_t *d_out_cnt = fuse ? d_scnt : cv.take<int32_t>(chunk);
    int64_t *d_expect = expect_id ? cv.take<int64_t>(chunk) : nullptr;
    int32_t *d_rank = expect_id ? cv.take<int32_t>(chunk) : nullptr;
    if (cv.at > own->rank_pool.bytes) {
        set_error("internal: rank scratch carve %lld > pool %lld", (long long)cv.at, (long long)own->rank_pool.bytes);
        return ANRAG_ERR_STATE;
    }

    if (fuse_in_lds) {  // the legs' lists, counts, maps, strides and weights: the same for every chunk
        constexpr int M = ANRAG_WRRF_MAX_LISTS;
        uint64_t tab[6 * M] = {};
        for (int l = 0; l < n_legs; ++l) {
            tab[l] = reinterpret_cast<uint64_t>(ld[l].rows);
            tab[M + l] = reinterpret_cast<uint64_t>(ld[l].cnt);
            tab[2 * M + l] = reinterpret_cast<uint64_t>(ld[l].doc_of_row);
            tab[3 * M + l] = (uint64_t)leg_k[l];
            memcpy(&tab[4 * M + l], &legs[l].weight, 8);
        }
        tab[5 * M] = (uint64_t)n_legs;
        ANRAG_HIP(hipMemcpyAsync(d_fuse_legs, tab, sizeof(tab), hipMemcpyHostToDevice, st));
        ANRAG_HIP(hipStreamSynchronize(st));
    }
    const bool count_only = !fuse && out_id == nullptr;  // (then out_rank is wanted: checked above)
    std::vector<int32_t> h_segk;
    std::vector<int64_t> h_off;
    for (int64_t q0 = 0; q0 < n_queries; q0 += chunk) {
        const int32_t c = (int32_t)std::min<int64_t>(chunk, n_queries - q0);
        if (expect_id) {
            ANRAG_HIP(hipMemcpyAsync(d_expect, expect_id + q0, (size_t)c * 8, hipMemcpyHostToDevice, st));
            ANRAG_HIP(hipMemsetAsync(d_rank, 0xFF, (size_t)c * 4, st));  // -1: not in the answer
        }
        for (int l = 0; l < n_legs; ++l) {
            const anrag_rank_leg &g = legs[l];
            anrag_index *ix = g.idx;
            if (g.kind == ANRAG_LEG_DENSE) {
                ANRAG_HIP(hipMemcpyAsync(ld[l].q, g.queries + q0 * ix->dim, (size_t)c * ix->dim * sizeof(float),
                                         hipMemcpyHostToDevice, st));
                float *tile = static_cast<float *>(ld[l].scores);
                const int tile_n = getenv("ANRAG_RANK_SCAN_TILES") ? 0 : dense_tile_group_max(ix);
                for (int32_t g0 = 0; tile_n > 0 && g0 < c; g0 += kTileLaunchMax)  // K1T: the rows once per launch
                    if ((rc = launch_dense_tile(ix, st, ld[l].q + (int64_t)g0 * ix->dim, ix->dim,
                                                std::min<int32_t>(kTileLaunchMax, c - g0), ld[l].allow,
                                                tile + (int64_t)g0 * ld[l].stride, ld[l].stride)))
                        return rc;
                for (int32_t g0 = tile_n > 0 ? c : 0; g0 < c; g0 += kScanGroupMax) {  // other dimensions: a pass per query
                    const int n = std::min<int32_t>(kScanGroupMax, c - g0);
                    const float *qs[kScanGroupMax];
                    int sets[kScanGroupMax];
                    for (int i = 0; i < n; ++i) {
                        qs[i] = ld[l].q + (int64_t)(g0 + i) * ix->dim;
                        sets[i] = 0;  // unused: scores are written, no lists
                    }
                    if ((rc = launch_dense_scan_group(ix, st, qs, n, 0, ld[l].allow, tile + (int64_t)g0 * ld[l].stride, sets,
                                                      ld[l].stride)))
                        return rc;
                }
                if (count_only)  // one list, rank of the expected document only: a count instead of the list
                    rank_count_kernel<float><<<c, kRankThreads, 0, st>>>(tile, ld[l].stride, (int32_t)leg_rows[l], leg_k[l], nullptr,
                                                                        ld[l].doc_of_row, out_n, d_expect, d_rank, d_out_cnt);
                else if ((rc = launch_seg_sort<float, false>(device, st, c, tile, ld[l].stride, nullptr, 0, (int32_t)leg_rows[l],
                                                             leg_k[l], nullptr, ld[l].rows, ld[l].row_keys, leg_k[l], ld[l].cnt)))
                    return rc;
            } else {
                const int64_t t0 = g.term_offsets[q0];
                const int64_t nt = g.term_offsets[q0 + c] - t0;
                if (nt > 0)
                    ANRAG_HIP(hipMemcpyAsync(ld[l].terms, g.term_ids + t0, (size_t)nt * 4, hipMemcpyHostToDevice, st));
                h_segk.resize(c);
                double *tile = static_cast<double *>(ld[l].scores);
                for (int32_t q = 0; q < c; ++q) {
                    const int32_t n_terms = (int32_t)(g.term_offsets[q0 + q + 1] - g.term_offsets[q0 + q]);
                    h_segk[q] = n_terms > 0 ? leg_k[l] : 0;  // no tokens: the leg is skipped (search_engine.py:216-217)
                }
                ANRAG_HIP(hipMemcpyAsync(ld[l].seg_k, h_segk.data(), (size_t)c * 4, hipMemcpyHostToDevice, st));
                h_off.resize((size_t)c + 1);
                for (int32_t q = 0; q <= c; ++q) h_off[q] = g.term_offsets[q0 + q] - t0;
                ANRAG_HIP(hipMemcpyAsync(ld[l].term_off, h_off.data(), ((size_t)c + 1) * 8, hipMemcpyHostToDevice, st));
                ANRAG_HIP(hipStreamSynchronize(st));  // h_segk and h_off are reused
                // K3 in its score-writing form, the chunk's queries in launches of up to 32,768 (a query without terms
                // writes nothing: its segment is sorted to length 0)
                for (int32_t g0 = 0; g0 < c; g0 += 32768)
                    if ((rc = launch_bm25_scores_table(ix, st, ld[l].terms, ld[l].term_off + g0, std::min<int32_t>(32768, c - g0),
                                                       ld[l].allow, tile + (int64_t)g0 * ld[l].stride, ld[l].stride)))
                        return rc;
                if (count_only)
                    rank_count_kernel<double><<<c, kRankThreads, 0, st>>>(tile, ld[l].stride, (int32_t)leg_rows[l], leg_k[l],
                                                                         ld[l].seg_k, ld[l].doc_of_row, out_n, d_expect, d_rank,
                                                                         d_out_cnt);
                else if ((rc = launch_seg_sort<double, false>(device, st, c, tile, ld[l].stride, nullptr, 0,
                                                              (int32_t)leg_rows[l], leg_k[l], ld[l].seg_k, ld[l].rows,
                                                              ld[l].row_keys, leg_k[l], ld[l].cnt)))
                    return rc;
            }
        }
        if (fuse && fuse_in_lds) {
            if ((rc = ensure_dynamic_lds(device, reinterpret_cast<const void *>(&rank_fuse_sort_kernel), rank_lds_bytes<3>())))
                return rc;
            rank_fuse_sort_kernel<<<c, kRankThreads, rank_lds_bytes<3>(), st>>>(
                d_fuse_legs, wrrf_k, (int32_t)id_space, out_n, out_id ? d_out_id : nullptr, d_out_score, d_out_cnt, d_expect,
                d_rank);
        } else if (fuse) {
            const int64_t total = (int64_t)c * id_space;
            rank_fill_kernel<<<(unsigned)((total + 255) / 256), 256, 0, st>>>(d_f, d_t, total);
            RankLegsDev L;
            for (int l = 0; l < ANRAG_WRRF_MAX_LISTS; ++l) {
                L.rows[l] = l < n_legs ? ld[l].rows : nullptr;
                L.rows_stride[l] = l < n_legs ? leg_k[l] : 0;
                L.doc_of_row[l] = l < n_legs ? ld[l].doc_of_row : nullptr;
            }
            for (int l = 0; l < n_legs; ++l) {
                const dim3 grid((unsigned)((leg_k[l] + 255) / 256), (unsigned)c);
                rank_accumulate_kernel<<<grid, 256, 0, st>>>(ld[l].rows, leg_k[l], ld[l].cnt, ld[l].doc_of_row, (uint32_t)l,
                                                             legs[l].weight, wrrf_k, id_space, d_f, d_t);
            }
            ANRAG_HIP(hipGetLastError());
            if ((rc = launch_seg_sort<double, true>(device, st, c, d_f, id_space, d_t, id_space, (int32_t)id_space, fuse_n,
                                                    nullptr, d_stie, d_skey, fuse_n, d_scnt)))
                return rc;
            const dim3 grid((unsigned)((out_n + 255) / 256), (unsigned)c);
            rank_emit_fused_kernel<<<grid, 256, 0, st>>>(L, d_stie, d_skey, fuse_n, d_scnt, out_n, d_out_id, d_out_score,
                                                         d_expect, d_rank);
        } else if (!count_only) {
            const dim3 grid((unsigned)((out_n + 255) / 256), (unsigned)c);
            rank_emit_single_kernel<<<grid, 256, 0, st>>>(ld[0].rows, ld[0].row_keys, leg_k[0], ld[0].cnt,
                                                          ld[0].doc_of_row, out_n, d_out_id, d_out_score, d_out_cnt,
                                                          d_expect, d_rank);
        }
        ANRAG_HIP(hipGetLastError());
        // results straight into the caller's arrays: rows of out_n entries into rows of top_n (the device wrote the
        // -1 / -inf padding of the first out_n columns; columns past out_n, if the caller asked for more, are filled here)
        if (out_id)
            ANRAG_HIP(hipMemcpy2DAsync(out_id + q0 * (int64_t)top_n, (size_t)top_n * 8, d_out_id, (size_t)out_n * 8,
                                       (size_t)out_n * 8, (size_t)c, hipMemcpyDeviceToHost, st));
        if (out_score)
            ANRAG_HIP(hipMemcpy2DAsync(out_score + q0 * (int64_t)top_n, (size_t)top_n * 8, d_out_score, (size_t)out_n * 8,
                                       (size_t)out_n * 8, (size_t)c, hipMemcpyDeviceToHost, st));
        ANRAG_HIP(hipMemcpyAsync(out_count + q0, d_out_cnt, (size_t)c * 4, hipMemcpyDeviceToHost, st));
        if (out_rank) ANRAG_HIP(hipMemcpyAsync(out_rank + q0, d_rank, (size_t)c * 4, hipMemcpyDeviceToHost, st));
        ANRAG_HIP(hipStreamSynchronize(st));  // the chunk's one host sync for results
        for (int32_t q = 0; q < c; ++q) {
            if (out_count[q0 + q] > out_n) out_count[q0 + q] = out_n;
            if (out_n < top_n && out_id) {
                int64_t *oi = out_id + (q0 + q) * (int64_t)top_n;
                double *os = out_score ? out_score + (q0 + q) * (int64_t)top_n : nullptr;
                for (int32_t i = out_n; i < top_n; ++i) {
                    oi[i] = -1;
                    if (os) os[i] = -__builtin_huge_val();
                }
            }
        }
    }
    return ANRAG_OK;
}
