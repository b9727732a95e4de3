// ic_capacity_probe.hip -- how large can the cyclically swept working set of the backward pass grow before the 256 MiB
// Infinity Cache stops holding it?  (could TWO backward passes share one thread, 28-30 arrays?)
//
// K arrays of one propagator field each (1064 x 2064 floats = 8.78 MB touched), swept once per launch the way the
// backward kernels do: every array read, one third of them written back (read-modify-write).  Reports the streaming rate
// for K = 12 .. 40.  The product's single backward pass is K = 20 (180 MB), a pair with shared accumulators and
// coefficients K = 28-30 (246-263 MB), two independent passes K = 35 (307 MB).
//   hipcc --offload-arch=gfx950 -O3 -o ic_capacity_probe ic_capacity_probe.hip && ./ic_capacity_probe
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

constexpr int MAXK = 40;
struct Ptrs {
    float *p[MAXK];
};

// one wave = 64 consecutive floats of one row; each thread touches its element of every array (like one cell of the grid)
template <int K>
__global__ __launch_bounds__(128) void k_sweep(Ptrs a, size_t n) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float v[K];
#pragma unroll
    for (int k = 0; k < K; k++) v[k] = a.p[k][i];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < K; k++) s += v[k];
#pragma unroll
    for (int k = 0; k < K; k += 3) a.p[k][i] = v[k] + 1e-9f * s;
}

template <int K>
int run(const Ptrs &a, size_t n, hipEvent_t e0, hipEvent_t e1) {
    const int reps = 60;
    const dim3 grid((unsigned)((n + 127) / 128));
    for (int r = 0; r < 10; r++) hipLaunchKernelGGL(k_sweep<K>, grid, dim3(128), 0, 0, a, n);
    OK(hipEventRecord(e0, 0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_sweep<K>, grid, dim3(128), 0, 0, a, n);
    OK(hipEventRecord(e1, 0));
    OK(hipEventSynchronize(e1));
    float ms = 0;
    OK(hipEventElapsedTime(&ms, e0, e1));
    const double bytes = (double)n * 4.0 * (K + (K + 2) / 3);
    printf("K = %2d arrays (%6.1f MB swept): %7.2f us per sweep, %.2f TB/s\n", K, n * 4.0 * K / 1e6, 1e3 * ms / reps, bytes / (ms / reps * 1e-3) / 1e12);
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
    for (int k = 0; k < MAXK; k++) {
        OK(hipMalloc((void **)&a.p[k], n * sizeof(float)));
        OK(hipMemcpy(a.p[k], h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    if (run<12>(a, n, e0, e1) || run<16>(a, n, e0, e1) || run<20>(a, n, e0, e1) || run<24>(a, n, e0, e1) || run<26>(a, n, e0, e1) ||
        run<28>(a, n, e0, e1) || run<30>(a, n, e0, e1) || run<32>(a, n, e0, e1) || run<35>(a, n, e0, e1) || run<40>(a, n, e0, e1))
        return 1;
    return 0;
}
