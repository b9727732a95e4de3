// bw_probe2.hip -- ceilings of the gfx950 memory hierarchy for read-only and read+write streams at
// several working-set sizes (L2-resident, Infinity-Cache-resident, HBM) and access widths.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, int UNROLL>
__global__ void k_read(const T *__restrict__ a, size_t n, float *out) {
    float acc = 0.f;
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            const float *f = reinterpret_cast<const float *>(&v[u]);
            for (int k = 0; k < (int)(sizeof(T) / 4); k++) acc += f[k];
        }
    }
    if (acc == 123.456f) out[0] = acc;
}

template <typename T, int UNROLL>
__global__ void k_rw(T *__restrict__ a, size_t n) {
    size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n; i += UNROLL * stride) {
        T v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) v[u] = a[i + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; u++) {
            float *f = reinterpret_cast<float *>(&v[u]);
            for (int k = 0; k < (int)(sizeof(T) / 4); k++) f[k] = f[k] * 1.0001f + 1.0f;
            a[i + u * stride] = v[u];
        }
    }
}

template <typename T, int UNROLL>
int run(const char *name, size_t bytes, int reps, int blocks, int threads, bool rw) {
    void *d; float *out;
    CK(hipMalloc(&d, bytes)); CK(hipMemset(d, 0, bytes)); CK(hipMalloc((void **)&out, 16));
    size_t n = bytes / sizeof(T);
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int w = 0; w < 3; w++) {
        if (rw) hipLaunchKernelGGL((k_rw<T, UNROLL>), dim3(blocks), dim3(threads), 0, 0, (T *)d, n);
        else hipLaunchKernelGGL((k_read<T, UNROLL>), dim3(blocks), dim3(threads), 0, 0, (const T *)d, n, out);
    }
    CK(hipEventRecord(a, 0));
    for (int r = 0; r < reps; r++) {
        if (rw) hipLaunchKernelGGL((k_rw<T, UNROLL>), dim3(blocks), dim3(threads), 0, 0, (T *)d, n);
        else hipLaunchKernelGGL((k_read<T, UNROLL>), dim3(blocks), dim3(threads), 0, 0, (const T *)d, n, out);
    }
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double moved = (double)bytes * (rw ? 2.0 : 1.0);
    printf("%-26s %9.1f MB  %s w%zu u%d blocks %6d x %4d : %8.2f us  %7.2f TB/s\n", name, bytes / 1e6, rw ? "r+w " : "read", sizeof(T),
           UNROLL, blocks, threads, ms * 1e3 / reps, moved / (ms * 1e-3 / reps) / 1e12);
    CK(hipFree(d)); CK(hipFree(out));
    return 0;
}

int main() {
    struct { const char *n; size_t b; int reps; } sets[] = {{"L2 (8 x 2 MB)", 16ull << 20, 400}, {"Infinity Cache", 128ull << 20, 100}, {"HBM", 2048ull << 20, 10}};
    for (auto &s : sets) {
        for (int blocks : {2048, 8192, 32768}) {
            run<float4, 4>(s.n, s.b, s.reps, blocks, 256, false);
        }
        run<float4, 1>(s.n, s.b, s.reps, 8192, 256, false);
        run<float, 4>(s.n, s.b, s.reps, 8192, 256, false);
        run<float, 1>(s.n, s.b, s.reps, 32768, 64, false);
        run<float4, 4>(s.n, s.b, s.reps, 8192, 256, true);
        run<float, 4>(s.n, s.b, s.reps, 8192, 256, true);
        run<float, 1>(s.n, s.b, s.reps, 32768, 64, true);
    }
    return 0;
}
