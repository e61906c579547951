// Standalone experiment (not part of the library): where does the time of the split-precision K2 full pass go?
// The 4-stage LDS-DMA kernel of round 2 with the copies and / or the MFMAs switched off.
// RESULT (MI355X, 1M x 768 as images): the copies alone -- 16 KB of corpus images from HBM + 16 KB of query images from
// L2 per step and CU -- take 0.68 us per step = 5.85 TB/s of corpus (4 or 8 copying waves alike; query copies alone
// 0.16 us): LDS-DMA staging is NOT what holds the kernel at 1.8 us per step.  The variants with MFMAs spill in this
// stand-alone build (132-156 VGPRs, unlike the library build of the same code): their times are not usable.
//   hipcc -O3 --offload-arch=gfx950 scripts/exp/k2dma_bench.hip -o scripts/exp/k2dma_bench && scripts/exp/k2dma_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int kSM = 256, kDK = 16, kDImg = kSM * kDK * 2, kDStage = 4 * kDImg, kThreads = 512;
__device__ __forceinline__ int dswz(int row, int h) { return row * 32 + ((h ^ ((row >> 3) & 1)) << 4); }

template <bool COPY_E, bool COPY_Q, bool COMPUTE, int WAVES_E>
__global__ __launch_bounds__(kThreads, 2) void k(const unsigned char* __restrict__ e_img, const unsigned char* __restrict__ q_img,
                                                 int ksteps, long n_tiles, float* __restrict__ sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lh = lane >> 5, rw = wave & 1, qw = wave >> 1;
    const long first_tile = blockIdx.x, tile_step = gridDim.x;
    const long my_tiles = first_tile < n_tiles ? (n_tiles - first_tile + tile_step - 1) / tile_step : 0;
    const long total = my_tiles * ksteps;
    if (total == 0) return;
    long ld_tile = first_tile; int ld_ks = 0;
    auto issue_stage = [&](int slot) {
        // WAVES_E waves copy the 16 KB of corpus images, the other 8 - WAVES_E the 16 KB of query images
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
    auto step = [&](const Frags& f, int next_slot, Frags& fn) {
        if constexpr (COMPUTE) {
#pragma unroll
            for (int ti = 0; ti < 4; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) {
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.al[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bl[tj], acc[ti][tj], 0, 0, 0);
                    acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(f.ah[ti], f.bh[tj], acc[ti][tj], 0, 0, 0);
                    if (ti == 0 && tj == 0) {
                        __builtin_amdgcn_sched_barrier(0);
                        read_frags(next_slot, fn);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
        }
    };
    issue_stage(0); issue_stage(1); issue_stage(2); issue_stage(3);
    asm volatile("s_waitcnt vmcnt(%0)\n\ts_barrier" ::"n"(COPY_E || COPY_Q ? 8 : 0) : "memory");
    Frags f0, f1;
    read_frags(0, f0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    for (long c = 0; c < total; c += 2) {
        issue_stage((int)(c & 3));
        step(f0, (int)((c + 1) & 3), f1);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (c + 1 >= total) break;
        issue_stage((int)((c + 1) & 3));
        step(f1, (int)((c + 2) & 3), f0);
        asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (ksteps < 0) {  // never: keeps the products alive without an epilogue in the measured path
#pragma unroll
        for (int ti = 0; ti < 4; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink[(ti * 2 + tj) * 16 + r + tid * 128] = acc[ti][tj][r];
    }
}

template <bool CE, bool CQ, bool CO, int WE>
static void run(const char* name, const unsigned char* e, const unsigned char* q, int ksteps, long tiles, float* sink) {
    auto kern = k<CE, CQ, CO, WE>;
    hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 4 * kDStage);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 2; ++i) kern<<<256, kThreads, 4 * kDStage>>>(e, q, ksteps, tiles, sink);
    hipEventRecord(a);
    for (int i = 0; i < 5; ++i) kern<<<256, kThreads, 4 * kDStage>>>(e, q, ksteps, tiles, sink);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 5;
    const double steps = (double)((tiles + 255) / 256) * ksteps;
    printf("%-44s %8.3f ms  %6.3f us/step  corpus %.2f TB/s\n", name, ms, ms * 1e3 / steps, CE ? tiles * ksteps * 16384.0 / (ms * 1e-3) / 1e12 : 0.0);
    if (hipGetLastError() != hipSuccess) printf("  HIP error\n");
}

int main() {
    const long rows = 1000000; const int dim = 768, ksteps = dim / kDK; const long tiles = (rows + kSM - 1) / kSM;
    const size_t eb = (size_t)tiles * ksteps * 2 * kDImg, qb = (size_t)ksteps * 2 * kDImg;
    unsigned char *e, *q; float* sink;
    hipMalloc(&e, eb); hipMalloc(&q, qb); hipMalloc(&sink, 4096);
    std::vector<unsigned short> h(qb / 2);
    for (auto& x : h) x = 0x3c00 + (rand() & 0xff);  // bf16 values near 0.01
    hipMemcpy(q, h.data(), qb, hipMemcpyHostToDevice);
    std::vector<unsigned short> he(1 << 22);
    for (auto& x : he) x = 0x3c00 + (rand() & 0x3ff);
    for (size_t off = 0; off < eb; off += he.size() * 2) hipMemcpy(e + off, he.data(), std::min(he.size() * 2, eb - off), hipMemcpyHostToDevice);
    hipDeviceSynchronize();
    run<true, true, true, 4>("full (4 + 4 copy waves)", e, q, ksteps, tiles, sink);
    run<true, true, false, 4>("copies only", e, q, ksteps, tiles, sink);
    run<true, false, false, 4>("corpus copies only (4 waves)", e, q, ksteps, tiles, sink);
    run<true, false, false, 8>("corpus copies only (8 waves)", e, q, ksteps, tiles, sink);
    run<false, true, false, 4>("query copies only (L2)", e, q, ksteps, tiles, sink);
    run<false, false, true, 4>("MFMAs + fragment reads only", e, q, ksteps, tiles, sink);
    run<true, false, true, 4>("corpus copies + MFMAs", e, q, ksteps, tiles, sink);
    run<false, true, true, 4>("query copies + MFMAs", e, q, ksteps, tiles, sink);
    return 0;
}
