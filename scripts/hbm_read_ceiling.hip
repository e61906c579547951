// Read-only HBM streaming ceiling on this box: what a kernel that ONLY loads (16 B/lane, nt) and adds can reach,
// in launch geometries around K1's (one 4-wave workgroup per CU, 12 dwordx4 per lane in flight).
// Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/hbm_read_ceiling.hip -o /tmp/ceil && /tmp/ceil
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void read_kernel(const f32x4 *__restrict__ p, size_t n_vec, float *out) {
    const size_t tid = (size_t)blockIdx.x * THREADS + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * THREADS;
    f32x4 acc = {0, 0, 0, 0};
    size_t i = tid;
    for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
        f32x4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += v[u];
    }
    for (; i < n_vec; i += stride) acc += p[i];
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

template <int THREADS, int UNROLL>
double run(const f32x4 *d, size_t n_vec, float *out, int grid) {
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; ++i) read_kernel<THREADS, UNROLL><<<grid, THREADS>>>(d, n_vec, out);
    hipEventRecord(a);
    const int iters = 20;
    for (int i = 0; i < iters; ++i) read_kernel<THREADS, UNROLL><<<grid, THREADS>>>(d, n_vec, out);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return n_vec * 16.0 / (ms / iters * 1e-3) / 1e9;
}

int main() {
    const size_t bytes = 3072000000ull, n_vec = bytes / 16;
    f32x4 *d; float *out;
    hipMalloc(&d, bytes); hipMalloc(&out, 4);
    hipMemset(d, 1, bytes);
    printf("read-only streaming, 3.072 GB, GB/s (back-to-back launches)\n");
    printf("T=256 U=12 grid=256 : %.0f\n", run<256, 12>(d, n_vec, out, 256));
    printf("T=256 U=6  grid=256 : %.0f\n", run<256, 6>(d, n_vec, out, 256));
    printf("T=256 U=12 grid=512 : %.0f\n", run<256, 12>(d, n_vec, out, 512));
    printf("T=512 U=6  grid=256 : %.0f\n", run<512, 6>(d, n_vec, out, 256));
    printf("T=256 U=24 grid=256 : %.0f\n", run<256, 24>(d, n_vec, out, 256));
    printf("T=1024 U=4 grid=256 : %.0f\n", run<1024, 4>(d, n_vec, out, 256));
    printf("T=256 U=8  grid=2048: %.0f\n", run<256, 8>(d, n_vec, out, 2048));
    return 0;
}
