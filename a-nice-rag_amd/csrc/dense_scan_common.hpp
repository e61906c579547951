// What K1 (dense_scan.hip) and K1T (dense_tile.hip) share: the per-row arithmetic -- the SAME FMA chain and reduction tree in
// both, which is what makes a tile score equal the scan's score bit for bit -- and the dimension -> kernel shape table.
#pragma once
#include "common.hpp"

namespace anrag {

typedef float f32x4 __attribute__((ext_vector_type(4)));  // native vector: global_load_dwordx4, nt-loadable
typedef float f32x2 __attribute__((ext_vector_type(2)));  // a 64-bit register pair: v_pk_fma_f32's operands

__device__ __forceinline__ float dot4(f32x4 a, f32x4 b, float acc) {
    acc = __builtin_fmaf(a.x, b.x, acc);
    acc = __builtin_fmaf(a.y, b.y, acc);
    acc = __builtin_fmaf(a.z, b.z, acc);
    acc = __builtin_fmaf(a.w, b.w, acc);
    return acc;
}

// ---- cross-lane sum without touching LDS: DPP row ops (GFX9 encodings)
//   0xB1 quad_perm[1,0,3,2]  0x4E quad_perm[2,3,0,1]  0x141 row_half_mirror  0x140 row_mirror
//   0x142 row_bcast:15 (row_mask 0xA)   0x143 row_bcast:31 (row_mask 0xC)
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_add(float v) {
    const int t = __builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xF, false);
    return v + __int_as_float(t);
}

// Sum over the G lanes that share a row.  The LAST lane of each group (lane % G == G-1) ends with the
// group's sum (for G <= 16 every lane does).
template <int G>
__device__ __forceinline__ float group_sum(float v) {
    if constexpr (G >= 2) v = dpp_add<0xB1, 0xF>(v);
    if constexpr (G >= 4) v = dpp_add<0x4E, 0xF>(v);
    if constexpr (G >= 8) v = dpp_add<0x141, 0xF>(v);
    if constexpr (G >= 16) v = dpp_add<0x140, 0xF>(v);
    if constexpr (G >= 32) v = dpp_add<0x142, 0xA>(v);
    if constexpr (G >= 64) v = dpp_add<0x143, 0xC>(v);
    return v;
}

// A NaN dot product (a corrupt row, a NaN in the query) ranks FIRST, as it does in the reference: numpy's
// argpartition / argsort order NaN above every number (src/search_engine.py:83-87).  It is carried as +inf from here
// on (and reported as +inf): every comparison downstream stays an ordinary float comparison.
__device__ __forceinline__ float nan_first(float v) { return v != v ? __builtin_huge_valf() : v; }
__device__ __forceinline__ double nan_first(double v) { return v != v ? __builtin_huge_val() : v; }

// row-groups per batch: 6-8 dwordx4 per lane per batch, two batches in flight = 12-16 loads per lane.  With counted
// waits (profiles/r02_scan_sweep.txt, 1M rows): 768-d R=1 5.27, R=2 7.23, R=3 6.97, R=4 6.97 TB/s; 1024-d R=1 6.34,
// R=2 7.15, R=3 7.00 TB/s; 384-d R=1 5.22, R=2 7.08, R=3 6.86 TB/s.  Two workgroups per CU: 768-d 6.96 TB/s.
template <int CH>
constexpr int scan_r() { return CH >= 6 ? 1 : (CH >= 3 ? 2 : (CH == 2 ? 3 : 6)); }

template <int G, int CH>
struct ScanShape {
    static constexpr int kG = G, kCH = CH, kR = scan_r<CH>();
};

// dim -> kernel shape; calls f(ScanShape<G, CH>{}) and returns true, or false for a dimension only the generic kernel takes
template <class F>
inline bool scan_dispatch(int d, F &&f) {
    if (d % 256 == 0 && d / 256 <= 16) {
        switch (d / 256) {
            case 1: f(ScanShape<32, 2>{}); return true;  // 256-d: two rows per load (<64,1>: 65 % of HBM peak, this: 86 %)
            case 2: f(ScanShape<64, 2>{}); return true;
            case 3: f(ScanShape<64, 3>{}); return true;
            case 4: f(ScanShape<64, 4>{}); return true;
            case 6: f(ScanShape<64, 6>{}); return true;
            case 8: f(ScanShape<64, 8>{}); return true;
            case 12: f(ScanShape<64, 12>{}); return true;  // 3072
            case 16: f(ScanShape<64, 16>{}); return true;  // 4096
            default: return false;
        }
    }
    if (d % 128 == 0 && d / 128 <= 8) {
        switch (d / 128) {
            case 1: f(ScanShape<32, 1>{}); return true;
            case 3: f(ScanShape<32, 3>{}); return true;
            case 5: f(ScanShape<32, 5>{}); return true;
            case 7: f(ScanShape<32, 7>{}); return true;
            default: return false;
        }
    }
    if (d % 64 == 0 && d / 64 <= 8) {
        switch (d / 64) {
            case 1: f(ScanShape<16, 1>{}); return true;
            case 3: f(ScanShape<16, 3>{}); return true;
            case 5: f(ScanShape<16, 5>{}); return true;
            case 7: f(ScanShape<16, 7>{}); return true;
            default: return false;
        }
    }
    return false;
}

}  // namespace anrag
