// Standalone experiment (not part of the library): where does the time of the split-precision K2 full pass go?
// The LDS-DMA ring of dense_batched_split_dma_kernel (four 32 KB stages, k-steps of 16, three in flight) with the copies,
// the fragment reads and the MFMAs switched on and off, with and without ping-pong waves, and with in-kernel clocks
// (s_memtime = shader clock, s_memrealtime = 100 MHz) so that every variant reports the clock it actually ran at.
// 200 warm-up + 200 timed back-to-back launches per variant (the clock needs tens of launches to settle);
// K2_REALISTIC=1 fills the images with N(0, 1/768) values split into bf16 hi / lo (the power the matrix pipes draw
// depends on the operand bits).
// RESULTS (MI355X, 1M x 768 as images, realistic operands; DESIGN.md section 3 has the table):
//   copies only 0.52 ms at 2.39 GHz (5.9 TB/s); MFMAs only 0.66 ms at 1.81 GHz with the pipe 100 % busy -- the matrix
//   pipes alone pull the board to its power limit; ping-pong without copies 0.70 ms; every FULL variant (no ping-pong,
//   ping-pong with the copies at phase start / in the read phase / spread over 4 or 8 waves, 2 or 3 stages in flight)
//   0.96-0.99 ms at 1.55-1.63 GHz: a schedule that removes stalls is answered by a lower clock.
//   The issue of an HBM-bound global_load_lds blocks its wave for 400-800 cycles (the stamps of the ping-pong variants).
//   The compiler moves MFMAs (register-only) across an `s_barrier` asm: sched_barrier(0) on both sides is needed.
//   hipcc -O3 --offload-arch=gfx950 scripts/exp/k2dma_bench.hip -o scripts/exp/k2dma_bench && K2_REALISTIC=1 scripts/exp/k2dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cstring>
#include <cmath>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kSM = 256, kDK = 16, kDImg = kSM * kDK * 2, kDStage = 4 * kDImg, kThreads = 512;
__device__ __forceinline__ int dswz(int row, int h) { return row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }

template <bool COPY_E, bool COPY_Q, bool COMPUTE, int WAVES_E, int NMFMA = 3, bool PP = false, int MODE = 0, int DEPTH = 3, bool STAMPS = false>
__global__ __launch_bounds__(kThreads, 2) void k(const unsigned char* __restrict__ e_img, const unsigned char* __restrict__ q_img,
                                                 int ksteps, long n_tiles, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, rw = wave & 1, qw = wave >> 1;
    const long first_tile = blockIdx.x, tile_step = gridDim.x;
    const long my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + tile_step - 1) / tile_step : 0;
    const long total = my_tiles * ksteps;
    if (total == 0) return;
    const long t0 = clock64(), w0 = wall_clock64();
    long ld_tile = first_tile; int ld_ks = 0;
    auto issue_stage = [&](int slot) {
        // WAVES_E waves copy the 16 KB of corpus images, the other 8 - WAVES_E the 16 KB of query images
        if constexpr (WAVES_E == 0) {  // every wave: 2 KB of corpus images + 2 KB of query images
            if constexpr (COPY_E) {
                const unsigned char* g = e_img + (ld_tile * ksteps + ld_ks) * (long)(2 * kDImg) + wave * 2048;
                unsigned char* d = lds + slot * kDStage + wave * 2048;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16, (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
            }
            if constexpr (COPY_Q) {
                const unsigned char* g = q_img + (long)ld_ks * (2 * kDImg) + wave * 2048;
                unsigned char* d = lds + slot * kDStage + 16384 + wave * 2048;
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16, (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
            }
        } else {
        constexpr int WQ = 8 - WAVES_E;
        if (wave < WAVES_E) {
            if constexpr (COPY_E) {
                const unsigned char* g = e_img + (ld_tile * ksteps + ld_ks) * (long)(2 * kDImg) + wave * (16384 / WAVES_E);
                unsigned char* d = lds + slot * kDStage + wave * (16384 / WAVES_E);
#pragma unroll
                for (int i = 0; i < 16 / WAVES_E; ++i)
                    __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16, (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
            }
        } else {
            if constexpr (COPY_Q) {
                const unsigned char* g = q_img + (long)ld_ks * (2 * kDImg) + (wave - WAVES_E) * (16384 / WQ);
                unsigned char* d = lds + slot * kDStage + 16384 + (wave - WAVES_E) * (16384 / WQ);
#pragma unroll
                for (int i = 0; i < 16 / WQ; ++i)
                    __builtin_amdgcn_global_load_lds(g + i * 1024 + lane * 16, (__attribute__((address_space(3))) void*)(d + i * 1024), 16, 0, 0);
            }
        }
        }
        if (++ld_ks == ksteps) { ld_ks = 0; if (ld_tile + tile_step < n_tiles) ld_tile += tile_step; }
    };
    f32x16 acc[4][2];
#pragma unroll
    for (int ti = 0; ti < 4; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    struct Frags { bf16x8 ah[4], al[4], bh[2], bl[2]; };
    auto read_frags = [&](int slot, Frags& f) {
        const unsigned char* base = lds + slot * kDStage;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int off = dswz(rw * 128 + t * 32 + l31, lh);
            f.ah[t] = *reinterpret_cast<const bf16x8*>(base + off);
            f.al[t] = *reinterpret_cast<const bf16x8*>(base + kDImg + off);
        }
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const int off = dswz(qw * 64 + t * 32 + l31, lh);
            f.bh[t] = *reinterpret_cast<const bf16x8*>(base + 2 * kDImg + off);
            f.bl[t] = *reinterpret_cast<const bf16x8*>(base + 3 * kDImg + off);
        }
    };
    issue_stage(0); issue_stage(1); if constexpr (DEPTH == 3) issue_stage(2);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(COPY_E || COPY_Q ? 8 : 0) : "memory");
    auto mfmas = [&](const Frags& f) {
        if constexpr (MODE >= 2) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bl[tj], acc[ti][tj], 0, 0, 0);
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
            return;
        }
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj) {
                acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
                if constexpr (NMFMA >= 2) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bl[tj], acc[ti][tj], 0, 0, 0);
                if constexpr (NMFMA >= 3) acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
            }
    };
    Frags f0;
    if constexpr (PP) {
        // ping-pong: waves 0-3 (one per SIMD) read step c's fragments while waves 4-7 multiply step c-1, then swap.
        // sched_barrier(0): the compiler otherwise moves MFMAs (register-only) across the s_barrier asm.
#define FENCE() __builtin_amdgcn_sched_barrier(0)
#define BAR() do { FENCE(); __builtin_amdgcn_s_barrier(); FENCE(); } while (0)
        Frags f;
        if (wave < 4) {
            long tr = 0, tb1 = 0, tm = 0, tb2 = 0, ti_ = 0;
            for (long c = 0; c < total; ++c) {
                FENCE(); const long s0 = STAMPS ? clock64() : 0; FENCE();
                if constexpr (MODE >= 3) read_frags((int)(c & 3), f);
                issue_stage((int)((c + DEPTH) & 3));
                FENCE(); const long si = STAMPS ? clock64() : 0; FENCE();
                if constexpr (MODE < 3) read_frags((int)(c & 3), f);
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                FENCE(); const long s1 = STAMPS ? clock64() : 0;
                ti_ += si - s0;
                BAR();
                const long s2 = STAMPS ? clock64() : 0; FENCE();
                mfmas(f);
                if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
                FENCE(); const long s3 = STAMPS ? clock64() : 0;
                BAR();
                const long s4 = STAMPS ? clock64() : 0; FENCE();
                tr += s1 - s0; tb1 += s2 - s1; tm += s3 - s2; tb2 += s4 - s3;
            }
            if (blockIdx.x == 17 && tid == 0) { sink[2] = (float)tr / total; sink[3] = (float)tb1 / total; sink[4] = (float)tm / total; sink[5] = (float)tb2 / total; sink[6] = (float)ti_ / total; }
        } else {
            for (long c = 0; c < total; ++c) {
                FENCE();
                if constexpr (MODE != 3) issue_stage((int)((c + DEPTH) & 3));
                if (c) mfmas(f);
                BAR();
                read_frags((int)(c & 3), f);
                if constexpr (MODE == 3) issue_stage((int)((c + DEPTH) & 3));
                if constexpr (DEPTH == 3) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
                BAR();
            }
            mfmas(f);
        }
    } else
    for (long c = 0; c < total; ++c) {
        const int slot = (int)(c & 3);
        issue_stage((slot + 3) & 3);
        if constexpr (COMPUTE) {
            if constexpr (MODE == 1) {
                if (c == 0) read_frags(slot, f0);
                asm volatile("" : "+v"(f0.ah[0]), "+v"(f0.bh[0]));
                mfmas(f0);
            } else {
                Frags f;
                read_frags(slot, f);
                mfmas(f);
            }
        }
        if constexpr (MODE != 1) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (blockIdx.x == 17 && tid == 0) { sink[0] = (float)(clock64() - t0); sink[1] = (float)(wall_clock64() - w0); }
    if (ksteps < 0) {  // never: keeps the products alive without an epilogue in the measured path
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink[8 + (ti * 2 + tj) * 16 + r + tid * 128] = acc[ti][tj][r];
    }
}

template <bool CE, bool CQ, bool CO, int WE, int NM = 3, bool PP = false, int MODE = 0, int DEPTH = 3, bool STAMPS = false>
static void run(const char* name, const unsigned char* e, const unsigned char* q, int ksteps, long tiles, float* sink, int kThreads = 512) {
    auto kern = k<CE, CQ, CO, WE, NM, PP, MODE, DEPTH, STAMPS>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kDStage);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 200; ++i) kern<<<256, kThreads, 4 * kDStage>>>(e, q, ksteps, tiles, sink);
    hipEventRecord(a);
    for (int i = 0; i < 200; ++i) kern<<<256, kThreads, 4 * kDStage>>>(e, q, ksteps, tiles, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 200;
    const double steps = (double)((tiles + 255) / 256) * ksteps;
    float clk[7]; hipMemcpy(clk, sink, 28, hipMemcpyDeviceToHost);
    if (PP && STAMPS) printf("   waves 0-3 per step: read %.0f (of which DMA issue %.0f)  barrier %.0f  mfma issue %.0f  barrier %.0f cycles\n", clk[2], clk[6], clk[3], clk[4], clk[5]);
    printf("%-44s %8.3f ms  %6.3f us/step  clock %.2f GHz  corpus %.2f TB/s\n", name, ms, ms * 1e3 / steps, clk[0] / clk[1] * 0.1, CE ? tiles * ksteps * 16384.0 / (ms * 1e-3) / 1e12 : 0.0);
    if (hipGetLastError() != hipSuccess) printf("  HIP error\n");
}

int main() {
    const long rows = 1000000; const int dim = 768, ksteps = dim / kDK; const long tiles = (rows + kSM - 1) / kSM;
    const size_t eb = (size_t)tiles * ksteps * 2 * kDImg, qb = (size_t)ksteps * 2 * kDImg;
    unsigned char *e, *q; float* sink;
    hipMalloc(&e, eb); hipMalloc(&q, qb); hipMalloc(&sink, 4096);
    // realistic operand bits (MFMA power depends on them): N(0, 1/768) values, hi = bf16(x), lo = bf16(x - hi), laid out
    // hi image / lo image alternately like the library's images (row order inside an image does not matter here)
    const bool realistic = getenv("K2_REALISTIC") != nullptr;
    auto bf16 = [](float x) { unsigned u; memcpy(&u, &x, 4); u += 0x7fff + ((u >> 16) & 1); return (unsigned short)(u >> 16); };
    auto back = [](unsigned short b) { unsigned u = (unsigned)b << 16; float x; memcpy(&x, &u, 4); return x; };
    auto fill = [&](std::vector<unsigned short>& v) {
        for (size_t blk = 0; blk + kDImg <= v.size(); blk += kDImg)  // kDImg shorts = one hi + one lo image
            for (int i = 0; i < kDImg / 2; ++i) {
                if (!realistic) { v[blk + i] = 0x3c00 + (rand() & 0x3ff); v[blk + kDImg / 2 + i] = 0x3c00 + (rand() & 0x3ff); continue; }
                float u1 = (rand() + 1.0f) / (RAND_MAX + 2.0f), u2 = (rand() + 1.0f) / (RAND_MAX + 2.0f);
                float x = sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2) * 0.0360844f;
                unsigned short hh = bf16(x);
                v[blk + i] = hh;
                v[blk + kDImg / 2 + i] = bf16(x - back(hh));
            }
    };
    std::vector<unsigned short> h(qb / 2);
    fill(h);
    hipMemcpy(q, h.data(), qb, hipMemcpyHostToDevice);
    std::vector<unsigned short> he(1 << 22);
    fill(he);
    for (size_t off = 0; off < eb; off += he.size() * 2) hipMemcpy(e + off, he.data(), std::min(he.size() * 2, eb - off), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    run<true, true, true, 4>("full (4 + 4 copy waves), no ping-pong", e, q, ksteps, tiles, sink);
    run<true, true, false, 4>("copies only", e, q, ksteps, tiles, sink);
    run<false, false, true, 4, 3, false, 1>("MFMAs only (fragments read once, no barriers)", e, q, ksteps, tiles, sink);
    run<false, false, true, 4, 3, true, 2>("ping-pong, no copies", e, q, ksteps, tiles, sink);
    run<true, true, true, 4, 3, true, 2>("ping-pong full, 4+4 copy waves at phase start", e, q, ksteps, tiles, sink);
    run<true, true, true, 0, 3, true, 2>("ping-pong full, 8 copy waves at phase start", e, q, ksteps, tiles, sink);
    run<true, true, true, 0, 3, true, 3>("ping-pong full, 8 copy waves in their read phase", e, q, ksteps, tiles, sink);
    run<true, true, true, 4, 3, true, 4>("ping-pong full, 4+4, E issue after fragment reads", e, q, ksteps, tiles, sink);
    run<true, true, true, 4, 3, true, 4, 3, true>("  the same with stamps", e, q, ksteps, tiles, sink);
    run<true, true, true, 0, 3, true, 3, 3, true>("  8-in-read-phase with stamps", e, q, ksteps, tiles, sink);
    run<true, true, true, 0, 3, true, 3, 2>("ping-pong full, 8 copy waves in read phase, 2 in flight", e, q, ksteps, tiles, sink);
    run<true, false, true, 0, 3, true, 3>("ping-pong, corpus copies only, in read phase", e, q, ksteps, tiles, sink);
    run<true, true, true, 4, 1>("full with 1 MFMA per product", e, q, ksteps, tiles, sink);
    return 0;
}
