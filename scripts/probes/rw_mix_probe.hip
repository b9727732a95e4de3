// rw_mix_probe.hip -- what do WRITES cost against reads on MI355X for a working set of the backward pass's size?
// 20 propagator-sized arrays (176 MB, the backward step's working set), one element per thread of every array like a cell of the grid;
// W of the 20 are read-modify-written, the others only read.  W = 0 (pure read) ... 20 (every array read and written).
// The backward step reads 30 and writes 15 array passes per time step (ratio 2 : 1, i.e. W = 10 here).
//   hipcc --offload-arch=gfx950 -O3 -o rw_mix_probe rw_mix_probe.hip && ./rw_mix_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define OK(x)                                                                    \
    do {                                                                         \
        hipError_t e = (x);                                                      \
        if (e != hipSuccess) {                                                   \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                            \
        }                                                                        \
    } while (0)
constexpr int K = 20;
struct Ptrs {
    float *p[K];
};
template <int W>
__global__ __launch_bounds__(128) void k_sweep(Ptrs a, size_t n, float *sink) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = a.p[k][i];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) s += v[k];
#pragma unroll
    for (int k = 0; k < W; k++) a.p[k][i] = v[k] + 1e-9f * s;
    if (W == 0 && s == 123.456f) *sink = s;   // keep the loads alive
}
template <int W>
int run(const Ptrs &a, size_t n, float *sink, hipEvent_t e0, hipEvent_t e1) {
    const int reps = 60;
    const dim3 grid((unsigned)((n + 127) / 128));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_sweep<W>, grid, dim3(128), 0, 0, a, n, sink);
    OK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_sweep<W>, grid, dim3(128), 0, 0, a, n, sink);
    OK(hipEventRecord(e1, 0));
    OK(hipEventSynchronize(e1));
    float ms = 0;
    OK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / reps, rd = (double)n * 4.0 * K, wr = (double)n * 4.0 * W;
    printf("W = %2d of 20 arrays written: %6.2f us per sweep; read %5.1f MB + written %5.1f MB = %.2f TB/s in all\n", W, us, rd / 1e6, wr / 1e6, (rd + wr) / (us * 1e-6) / 1e12);
    return 0;
}
int main() {
    const size_t n = (size_t)1064 * 2064;
    Ptrs a;
    std::vector<float> h(n);
    unsigned s = 99u;
    for (size_t i = 0; i < n; i++) {
        s = s * 1664525u + 1013904223u;
        h[i] = (float)(s >> 8) / 16777216.0f;
    }
    for (int k = 0; k < K; k++) {
        OK(hipMalloc((void **)&a.p[k], n * sizeof(float)));
        OK(hipMemcpy(a.p[k], h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    float *sink;
    OK(hipMalloc((void **)&sink, 4));
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    if (run<0>(a, n, sink, e0, e1) || run<4>(a, n, sink, e0, e1) || run<7>(a, n, sink, e0, e1) || run<10>(a, n, sink, e0, e1) || run<14>(a, n, sink, e0, e1) ||
        run<20>(a, n, sink, e0, e1))
        return 1;
    return 0;
}
