// Stand-alone experiment: is v_mfma_f32_16x16x4_f32 bit for bit D[i][j] = fma(a3,b3, fma(a2,b2, fma(a1,b1, fma(a0,b0, C))))
// -- a sequential fmaf chain over k starting from C -- also for operands with wildly different magnitudes, signs, exact
// cancellations and values near the denormal range?  (K1T on the matrix cores needs exactly that: one MFMA = the scan's
// per-lane chain over one float4.)  Also prints the operand / result lane layout it assumed.
//   hipcc -O3 --offload-arch=gfx950 -ffp-contract=off scripts/exp/mfma_chain_order.hip -o scripts/exp/mfma_chain_order
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// one wave: A is 16 x 4 (row i, k), B is 4 x 16 (k, col j), C / D 16 x 16.
// lane l supplies A[l % 16][l / 16] and B[l / 16][l % 16]; holds D[(l / 16) * 4 + v][l % 16] in register v.
__global__ void k(const float* A, const float* B, const float* C, float* D, int steps) {
    const int l = threadIdx.x;
    f32x4 acc;
    for (int v = 0; v < 4; ++v) acc[v] = C[((l / 16) * 4 + v) * 16 + l % 16];
    for (int s = 0; s < steps; ++s) {
        const float a = A[s * 64 + (l % 16) * 4 + l / 16];
        const float b = B[s * 64 + (l / 16) * 16 + l % 16];
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    }
    for (int v = 0; v < 4; ++v) D[((l / 16) * 4 + v) * 16 + l % 16] = acc[v];
}

int main() {
    const int steps = 6, trials = 4000;
    std::vector<float> A(steps * 64), B(steps * 64), C(256), D(256);
    float *dA, *dB, *dC, *dD;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 1024); hipMalloc(&dD, 1024);
    srand(7);
    auto rnd = [&](int mode) {
        float m = (float)rand() / RAND_MAX * 2.f - 1.f;
        int e = mode == 0 ? 0 : (mode == 1 ? rand() % 40 - 20 : (mode == 2 ? rand() % 120 - 60 : -(rand() % 30) - 45));
        return ldexpf(m, e);
    };
    long bad_fwd = 0, bad_rev = 0, bad_unfused = 0, total = 0;
    for (int t = 0; t < trials; ++t) {
        const int mode = t % 4;
        for (auto& x : A) x = rnd(mode);
        for (auto& x : B) x = rnd(mode);
        for (auto& x : C) x = t % 7 == 0 ? 0.f : rnd(mode);
        if (t % 5 == 0)  // exact cancellations inside a chain
            for (int s = 0; s < steps; ++s)
                for (int i = 0; i < 16; ++i) { A[s * 64 + i * 4 + 1] = -A[s * 64 + i * 4 + 0]; }
        if (t % 5 == 0)
            for (int s = 0; s < steps; ++s)
                for (int j = 0; j < 16; ++j) { B[s * 64 + 1 * 16 + j] = B[s * 64 + 0 * 16 + j]; }
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        k<<<1, 64>>>(dA, dB, dC, dD, steps);
        hipMemcpy(D.data(), dD, 1024, hipMemcpyDeviceToHost);
        for (int i = 0; i < 16; ++i)
            for (int j = 0; j < 16; ++j) {
                float f = C[i * 16 + j], r = C[i * 16 + j], u = C[i * 16 + j];
                for (int s = 0; s < steps; ++s) {
                    for (int kk = 0; kk < 4; ++kk) f = fmaf(A[s * 64 + i * 4 + kk], B[s * 64 + kk * 16 + j], f);
                    for (int kk = 3; kk >= 0; --kk) r = fmaf(A[s * 64 + i * 4 + kk], B[s * 64 + kk * 16 + j], r);
                    for (int kk = 0; kk < 4; ++kk) u = u + A[s * 64 + i * 4 + kk] * B[s * 64 + kk * 16 + j];
                }
                const float d = D[i * 16 + j];
                ++total;
                bad_fwd += memcmp(&d, &f, 4) != 0;
                bad_rev += memcmp(&d, &r, 4) != 0;
                bad_unfused += memcmp(&d, &u, 4) != 0;
            }
    }
    printf("v_mfma_f32_16x16x4_f32 vs host chains over %ld outputs (6 MFMAs each; magnitudes 1, 2^+-20, 2^+-60, 2^-45..-75; "
           "cancellations): fmaf k = 0..3 from C: %ld differ; fmaf k = 3..0: %ld differ; unfused mul+add: %ld differ\n",
           total, bad_fwd, bad_rev, bad_unfused);
    return bad_fwd != 0;
}
