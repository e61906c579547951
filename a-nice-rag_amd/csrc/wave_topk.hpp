// Per-wavefront top-k kept in registers: lane i holds the i-th best candidate.
//
// Order: score descending, then row ascending (a total order: rows are unique),
// which is the build's deterministic reading of the reference's selection at
// src/search_engine.py:83-87 / :233 / :236-243.  An empty slot is
// (-inf, kNoRow): the minimum of that order, so real rows with score -inf or 0
// still rank (the reference does not drop zero-score BM25 documents).  All 64 lanes always hold the
// wave's 64 best so far (only the first k are consumed): merges are bitonic networks over whole lists.
// NaN never beats anything and is therefore never selected (documented gap:
// numpy would rank NaN first).
#pragma once
#include "common.hpp"

namespace anrag {

template <typename S>
__device__ __forceinline__ S neg_inf();
template <>
__device__ __forceinline__ float neg_inf<float>() { return -__builtin_huge_valf(); }
template <>
__device__ __forceinline__ double neg_inf<double>() { return -__builtin_huge_val(); }

template <typename S>
__device__ __forceinline__ bool beats(S s, uint32_t r, S ts, uint32_t tr) {
    return s > ts || (s == ts && r < tr);
}

// allow bitmap (staged in LDS): bit s of word s/32 = rows with source id s pass the filter
__device__ __forceinline__ bool source_ok(const uint32_t *allow_lds, uint32_t s) {
    return (allow_lds[s >> 5] >> (s & 31)) & 1u;
}

__device__ __forceinline__ int lane_id() { return threadIdx.x & (kWave - 1); }

// Wave-uniform read of one lane's value (v_readlane_b32: SGPR result).
__device__ __forceinline__ uint32_t read_lane(uint32_t v, int l) {
    return (uint32_t)__builtin_amdgcn_readlane((int)v, l);
}
__device__ __forceinline__ float read_lane(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}
__device__ __forceinline__ double read_lane(double v, int l) {
    long long b = __double_as_longlong(v);
    uint32_t lo = read_lane((uint32_t)b, l), hi = read_lane((uint32_t)((unsigned long long)b >> 32), l);
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- lane exchanges without LDS.  A ds_bpermute round trip is ~250 cycles and a sorting network is a chain of
// them (one 27-stage sort+merge of fp64 keys measured 2-3 us); these are VALU moves of a few cycles each:
//   lane ^ 1, ^ 2      DPP quad_perm                                   lane ^ 16   v_permlane16_swap (gfx950)
//   lane ^ 4           DPP row_half_mirror, then quad_perm [3,2,1,0]    lane ^ 32   v_permlane32_swap (gfx950)
//   lane ^ 8           DPP row_mirror, then row_half_mirror
template <int CTRL>
__device__ __forceinline__ uint32_t dpp_move(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, 0xF, 0xF, false);
}
template <int STRIDE>
__device__ __forceinline__ uint32_t lane_xor(uint32_t v) {
    static_assert(STRIDE == 1 || STRIDE == 2 || STRIDE == 4 || STRIDE == 8 || STRIDE == 16 || STRIDE == 32, "");
    if constexpr (STRIDE == 1) return dpp_move<0xB1>(v);
    else if constexpr (STRIDE == 2) return dpp_move<0x4E>(v);
    else if constexpr (STRIDE == 4) return dpp_move<0x1B>(dpp_move<0x141>(v));
    else if constexpr (STRIDE == 8) return dpp_move<0x141>(dpp_move<0x140>(v));
    else if constexpr (STRIDE == 16) {
        // odd rows of the first copy <-> even rows of the second: {r0,r0,r2,r2} and {r1,r1,r3,r3}
        const auto p = __builtin_amdgcn_permlane16_swap(v, v, false, false);
        return (threadIdx.x & 16) ? p[0] : p[1];
    } else {
        // upper half of the first copy <-> lower half of the second: {lo,lo} and {hi,hi}
        const auto p = __builtin_amdgcn_permlane32_swap(v, v, false, false);
        return (threadIdx.x & 32) ? p[0] : p[1];
    }
}
template <int STRIDE>
__device__ __forceinline__ float lane_xor(float v) {
    return __uint_as_float(lane_xor<STRIDE>(__float_as_uint(v)));
}
template <int STRIDE>
__device__ __forceinline__ double lane_xor(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_xor<STRIDE>((uint32_t)b), hi = lane_xor<STRIDE>((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// lane -> 63 - lane (= lane ^ 63): mirror inside the rows of 16, then swap rows and halves
__device__ __forceinline__ uint32_t lane_reverse(uint32_t v) { return lane_xor<32>(lane_xor<16>(dpp_move<0x140>(v))); }
__device__ __forceinline__ float lane_reverse(float v) { return __uint_as_float(lane_reverse(__float_as_uint(v))); }
__device__ __forceinline__ double lane_reverse(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_reverse((uint32_t)b), hi = lane_reverse((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// value of lane - 1 (lane 0 keeps its own): DPP wave_shr:1
__device__ __forceinline__ uint32_t lane_up1(uint32_t v) {
    return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, 0x138, 0xF, 0xF, false);
}
__device__ __forceinline__ float lane_up1(float v) { return __uint_as_float(lane_up1(__float_as_uint(v))); }
__device__ __forceinline__ double lane_up1(double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const uint32_t lo = lane_up1((uint32_t)b), hi = lane_up1((uint32_t)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}

// ---- sorting networks over the 64 lanes (one (score,row) pair per lane, best first)
// compare-exchange with lane ^ STRIDE: keep the better of the pair when `keep_better`, else the worse
template <int STRIDE, typename S>
__device__ __forceinline__ void cmp_exchange(S &s, uint32_t &r, bool keep_better) {
    const S os = lane_xor<STRIDE>(s);
    const uint32_t orow = lane_xor<STRIDE>(r);
    const bool other_better = beats(os, orow, s, r);
    if (other_better == keep_better) {
        s = os;
        r = orow;
    }
}

template <int SIZE, int STRIDE, typename S>
__device__ __forceinline__ void bitonic_sort_from(S &s, uint32_t &r, int lane) {
    if constexpr (SIZE <= kWave) {
        const bool lower = (lane & STRIDE) == 0;
        const bool desc = (lane & SIZE) == 0 || SIZE == kWave;
        cmp_exchange<STRIDE>(s, r, lower == desc);
        if constexpr (STRIDE > 1) bitonic_sort_from<SIZE, STRIDE / 2>(s, r, lane);
        else bitonic_sort_from<SIZE * 2, SIZE>(s, r, lane);
    }
}
// full bitonic sort, descending by (score desc, row asc): 21 compare-exchange stages
template <typename S>
__device__ __forceinline__ void bitonic_sort64(S &s, uint32_t &r) {
    bitonic_sort_from<2, 1>(s, r, threadIdx.x & (kWave - 1));
}

template <int STRIDE, typename S>
__device__ __forceinline__ void bitonic_merge_from(S &s, uint32_t &r, int lane) {
    cmp_exchange<STRIDE>(s, r, (lane & STRIDE) == 0);
    if constexpr (STRIDE > 1) bitonic_merge_from<STRIDE / 2>(s, r, lane);
}
// (s, r) holds a bitonic sequence (first descending, then ascending, or any rotation-free bitonic
// shape produced by the max-with-reversed trick below): 6 stages sort it descending
template <typename S>
__device__ __forceinline__ void bitonic_merge64(S &s, uint32_t &r) {
    bitonic_merge_from<kWave / 2>(s, r, threadIdx.x & (kWave - 1));
}

template <typename S>
struct WaveTopK {
    S s;          // this lane's slot
    uint32_t r;
    S thr_s;      // wave-uniform copy of slot k-1 (what a newcomer has to beat)
    uint32_t thr_r;
    int k;

    __device__ __forceinline__ void init(int k_) {
        s = neg_inf<S>();
        r = kNoRow;
        thr_s = neg_inf<S>();
        thr_r = kNoRow;
        k = k_;
    }
    __device__ __forceinline__ bool admits(S cs, uint32_t cr) const { return beats(cs, cr, thr_s, thr_r); }

    // Insert a wave-uniform candidate that admits() accepted: every slot it outranks moves down one lane.
    __device__ __forceinline__ void insert(S cs, uint32_t cr) {
        const int lane = lane_id();
        const bool ahead = beats(s, r, cs, cr);  // this slot outranks the newcomer
        const S up_s = lane_up1(s);
        const uint32_t up_r = lane_up1(r);
        const uint32_t up_ahead = lane_up1((uint32_t)ahead);
        if (!ahead) {
            const bool first = (lane == 0) || up_ahead;
            s = first ? cs : up_s;
            r = first ? cr : up_r;
        }
        thr_s = read_lane(s, k - 1);
        thr_r = read_lane(r, k - 1);
    }

    // Keep the best 64 of {this list} U {another list sorted best-first whose entry 63-lane is (rs, rr)}:
    // lane-wise max against the reversed list is a bitonic sequence holding exactly those 64.
    __device__ __forceinline__ void merge_reversed(S rs, uint32_t rr) {
        if (beats(rs, rr, s, r)) {
            s = rs;
            r = rr;
        }
        bitonic_merge64(s, r);
        thr_s = read_lane(s, k - 1);
        thr_r = read_lane(r, k - 1);
    }

    // Offer one candidate per flagged lane (flag already includes admits()).  Few candidates: wave-uniform
    // insertion loop.  Many (the first rounds of a merge, BM25's dense slices): sort them and merge networks,
    // whose cost does not depend on the count.
    __device__ __forceinline__ void offer_lanes(bool flag, S cs, uint32_t cr) {
        unsigned long long m = __ballot(flag);
        if (m == 0) return;
        if (__builtin_popcountll(m) <= 5) {
            while (m) {
                const int l = __builtin_ctzll(m);
                m &= m - 1;
                const S us = read_lane(cs, l);
                const uint32_t ur = read_lane(cr, l);
                if (admits(us, ur)) insert(us, ur);
            }
            return;
        }
        S ss = flag ? cs : neg_inf<S>();
        uint32_t sr = flag ? cr : kNoRow;
        bitonic_sort64(ss, sr);
        if (read_lane(r, 0) == kNoRow) {  // empty list (its best slot is): the sorted candidates ARE the new list
            s = ss;
            r = sr;
            thr_s = read_lane(s, k - 1);
            thr_r = read_lane(r, k - 1);
            return;
        }
        merge_reversed(lane_reverse(ss), lane_reverse(sr));
    }

    // Merge a sorted 64-entry list that lives in LDS (lane i reads entry 63-i: the reversal is free).
    __device__ __forceinline__ void merge_sorted(const S *ls, const uint32_t *lr) {
        const int rev = kWave - 1 - lane_id();
        merge_reversed(ls[rev], lr[rev]);
    }
};

// Tree-merge the per-wave lists of one workgroup through LDS; wave 0 ends up with the block's top-k.
// lds_s / lds_r: [n_waves][kListLen].  All waves must call this (it has barriers).
template <typename S>
__device__ __forceinline__ void block_merge(WaveTopK<S> &t, S *lds_s, uint32_t *lds_r, int n_waves) {
    const int wave = threadIdx.x / kWave, lane = lane_id();
    for (int half = n_waves >> 1; half >= 1; half >>= 1) {
        if (wave >= half && wave < 2 * half) {
            lds_s[wave * kListLen + lane] = t.s;
            lds_r[wave * kListLen + lane] = t.r;
        }
        __syncthreads();
        if (wave < half) t.merge_sorted(lds_s + (wave + half) * kListLen, lds_r + (wave + half) * kListLen);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// Merge the sorted per-workgroup lists K1 / K3 leave in HBM ([n_lists][64] score + local row) into `top`.
// Called by the `n_waves` waves of a group (wave_in_group = 0..n_waves-1), each taking every n_waves-th
// chunk of 64 lists: lane <-> list.  A lane first pulls kPre entries of its list in ONE batch of independent
// loads (the dependent HBM round trips are what this costs, not the arithmetic), then rounds: round j
// offers every still-live list's j-th entry; a list drops out at its first loser (it is sorted).
template <typename S>
__device__ __forceinline__ void merge_lists(WaveTopK<S> &top, const S *__restrict__ blk_score,
                                            const uint32_t *__restrict__ blk_row, int32_t n_lists, int32_t k,
                                            int wave_in_group, int n_waves) {
    constexpr int kPre = 4;
    const int lane = lane_id();
    for (int l0 = wave_in_group * kWave; l0 < n_lists; l0 += n_waves * kWave) {
        const int list = l0 + lane;
        bool live = list < n_lists;
        const S *ps = blk_score + (int64_t)(live ? list : 0) * kListLen;
        const uint32_t *pr = blk_row + (int64_t)(live ? list : 0) * kListLen;
        S es[kPre];
        uint32_t er[kPre];
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            es[j] = (live && j < k) ? ps[j] : neg_inf<S>();
            er[j] = (live && j < k) ? pr[j] : kNoRow;
        }
        bool done = false;
#pragma unroll
        for (int j = 0; j < kPre; ++j) {
            if (!done) {
                const bool cand = live && er[j] != kNoRow && top.admits(es[j], er[j]);
                if (__ballot(cand) == 0) {
                    done = true;
                } else {
                    top.offer_lanes(cand, es[j], er[j]);
                    live = cand && !beats(top.thr_s, top.thr_r, es[j], er[j]);  // pushed out again: exhausted
                }
            }
        }
        if (done) continue;
        for (int j = kPre; j < k; ++j) {  // rare: one list holds more than kPre winners
            const S cs = live ? ps[j] : neg_inf<S>();
            const uint32_t cr = live ? pr[j] : kNoRow;
            const bool cand = live && cr != kNoRow && top.admits(cs, cr);
            if (__ballot(cand) == 0) break;
            top.offer_lanes(cand, cs, cr);
            live = cand && !beats(top.thr_s, top.thr_r, cs, cr);
        }
    }
}

}  // namespace anrag
