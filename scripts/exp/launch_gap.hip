// Where do the ~10 us per launch of a 100k x 768 single-query scan go?  Stand-alone measurements on the box:
//   (a) an empty kernel back to back on one stream                      -> the launch boundary itself
//   (b) the same with a hipEventRecord behind every launch               -> what the pipeline's per-query marker adds
//   (c) a read-only streaming kernel of K1's geometry over 307 MB: N launches back to back, against ONE launch that
//       makes N passes (no boundary, no ramp, no drain between passes)   -> boundary + ramp + drain of a pure stream
//   (d) (c) with the passes of consecutive launches on 2 / 4 streams     -> do kernels of different streams overlap?
// Build + run:  hipcc --offload-arch=gfx950 -O3 scripts/exp/launch_gap.hip -o /tmp/launch_gap && /tmp/launch_gap
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void empty_kernel(int *p) {
    if (p && threadIdx.x == 12345) *p = 1;
}

template <int UNROLL>
__global__ __launch_bounds__(256) void read_kernel(const f32x4 *__restrict__ p, size_t n_vec, int passes, float *out) {
    const size_t tid = (size_t)blockIdx.x * 256 + threadIdx.x;
    const size_t stride = (size_t)gridDim.x * 256;
    f32x4 acc = {0, 0, 0, 0};
    for (int pass = 0; pass < passes; ++pass) {
        size_t i = tid;
        for (; i + (UNROLL - 1) * stride < n_vec; i += UNROLL * stride) {
            f32x4 v[UNROLL];
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(p + i + u * stride);
#pragma unroll
            for (int u = 0; u < UNROLL; ++u) acc += v[u];
        }
        for (; i < n_vec; i += stride) acc += p[i];
    }
    if (acc.x + acc.y + acc.z + acc.w == 123.456f) out[0] = acc.x;
}

static double now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int main() {
    const size_t bytes = 100000ull * 768 * 4, n_vec = bytes / 16;
    f32x4 *d;
    float *out;
    int *flag;
    hipMalloc(&d, bytes);
    hipMalloc(&out, 4);
    hipMalloc(&flag, 4);
    hipMemset(d, 1, bytes);
    hipStream_t s[4];
    hipEvent_t ev[64];
    for (auto &x : s) hipStreamCreateWithFlags(&x, hipStreamNonBlocking);
    for (auto &e : ev) hipEventCreateWithFlags(&e, hipEventDisableTiming);
    const int N = 2000;
    auto timed = [&](const char *what, auto &&body, int per) {
        body();
        hipDeviceSynchronize();
        const double t0 = now();
        body();
        hipDeviceSynchronize();
        printf("%-72s %8.2f us each\n", what, (now() - t0) / per * 1e6);
    };
    timed("(a) empty kernel, back to back, one stream", [&] { for (int i = 0; i < N; ++i) empty_kernel<<<256, 256, 0, s[0]>>>(flag); }, N);
    timed("(b) empty kernel + hipEventRecord behind each", [&] {
        for (int i = 0; i < N; ++i) { empty_kernel<<<256, 256, 0, s[0]>>>(flag); hipEventRecord(ev[i % 64], s[0]); } }, N);
    timed("(b') ... + another stream waiting for that event and running an empty kernel", [&] {
        for (int i = 0; i < N; ++i) { empty_kernel<<<256, 256, 0, s[0]>>>(flag); hipEventRecord(ev[i % 64], s[0]);
                                      hipStreamWaitEvent(s[1], ev[i % 64], 0); empty_kernel<<<1, 512, 0, s[1]>>>(flag); } }, N);
    timed("(c) 307 MB read-only stream, one pass per launch, back to back", [&] {
        for (int i = 0; i < N; ++i) read_kernel<6><<<256, 256, 0, s[0]>>>(d, n_vec, 1, out); }, N);
    timed("(c') the same passes inside ONE launch", [&] { read_kernel<6><<<256, 256, 0, s[0]>>>(d, n_vec, N, out); }, N);
    timed("(c'') 8 passes per launch", [&] { for (int i = 0; i < N / 8; ++i) read_kernel<6><<<256, 256, 0, s[0]>>>(d, n_vec, 8, out); }, N);
    timed("(d) one pass per launch, launches alternating over 2 streams", [&] {
        for (int i = 0; i < N; ++i) read_kernel<6><<<256, 256, 0, s[i % 2]>>>(d, n_vec, 1, out); }, N);
    timed("(d') ... over 4 streams", [&] {
        for (int i = 0; i < N; ++i) read_kernel<6><<<256, 256, 0, s[i % 4]>>>(d, n_vec, 1, out); }, N);
    timed("(e) one pass per launch + event record behind each (one stream)", [&] {
        for (int i = 0; i < N; ++i) { read_kernel<6><<<256, 256, 0, s[0]>>>(d, n_vec, 1, out); hipEventRecord(ev[i % 64], s[0]); } }, N);
    return 0;
}
