// Stand-alone experiment: the issue rate of v_mfma_f32_32x32x2_f32 as a function of which VGPR banks its A and B
// operands sit in (bank = register number mod 4).  64 MFMAs per step in the order of dense_batched.hip's loop, operands
// held in registers (no LDS, no barriers), 2 waves per SIMD, all 256 CUs.
//   hipcc -O3 --offload-arch=gfx950 scripts/exp/mfma_f32_banks.hip -o scripts/exp/mfma_f32_banks && scripts/exp/mfma_f32_banks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int ROT, int WAVES>
__global__ __launch_bounds__(WAVES * 64, 2) void k(float* __restrict__ sink, int steps, const float* __restrict__ src) {
    f32x16 acc[2][2];
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[ti][tj][r] = 0.f;
    f32x4 fa[2], fb[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        fa[t] = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4 + t * 4096);
        fb[t] = *reinterpret_cast<const f32x4*>(src + threadIdx.x * 4 + t * 4096 + 8192);
    }
    const long t0 = clock64(), w0 = wall_clock64();
    for (int it = 0; it < steps; ++it) {
        asm volatile("" : "+v"(fa[0]), "+v"(fa[1]), "+v"(fb[0]), "+v"(fb[1]));
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj)
                        acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[ti][s], fb[tj][(s + ROT) & 3], acc[ti][tj], 0, 0, 0);
    }
    if (blockIdx.x == 17 && threadIdx.x == 0) { sink[0] = (float)(clock64() - t0); sink[1] = (float)(wall_clock64() - w0); }
    if (steps < 0)
#pragma unroll
        for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
                for (int r = 0; r < 16; ++r) sink[2 + threadIdx.x * 64 + (ti * 2 + tj) * 16 + r] = acc[ti][tj][r];
}

template <int ROT, int WAVES>
static void run(const char* name, float* sink, const float* src) {
    const int steps = 720;  // ~ the k-steps one CU runs in a 1M x 768 pass (30.5 tiles x 24)
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 50; ++i) k<ROT, WAVES><<<256, WAVES * 64>>>(sink, steps, src);
    hipEventRecord(a);
    for (int i = 0; i < 100; ++i) k<ROT, WAVES><<<256, WAVES * 64>>>(sink, steps, src);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); ms /= 100;
    float clk[2]; hipMemcpy(clk, sink, 8, hipMemcpyDeviceToHost);
    const double cyc = clk[0] / steps, ideal = 64.0 * 64 * (WAVES / 4);
    printf("%-46s %7.3f ms  clock %.2f GHz  %6.0f cycles/step (ideal %5.0f: pipe %.1f %% busy)  %.1f TFLOP/s\n", name, ms,
           clk[0] / clk[1] * 0.1, cyc, ideal, 100 * ideal / cyc, 256.0 * WAVES * steps * 64 * 4096 / (ms * 1e-3) / 1e12);
}

int main() {
    float *sink, *src;
    (void)hipMalloc(&sink, 1 << 20); (void)hipMalloc(&src, 1 << 20);
    {   // realistic operand bits: the power the matrix pipes draw (and with it the clock) depends on them
        static float h[1 << 18];
        for (int i = 0; i < (1 << 18); ++i) h[i] = ((rand() & 0xffff) - 32768) * (0.0360844f / 18918.f);
        (void)hipMemcpy(src, h, sizeof h, hipMemcpyHostToDevice);
    }
    run<0, 8>("A[s] x B[s]      (same bank), 2 waves/SIMD", sink, src);
    run<2, 8>("A[s] x B[s+2]    (banks differ), 2 waves/SIMD", sink, src);
    run<1, 8>("A[s] x B[s+1]    (banks differ), 2 waves/SIMD", sink, src);
    run<0, 4>("A[s] x B[s]      (same bank), 1 wave/SIMD", sink, src);
    run<2, 4>("A[s] x B[s+2]    (banks differ), 1 wave/SIMD", sink, src);
    return 0;
}
