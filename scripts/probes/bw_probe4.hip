// bw_probe4.hip -- what does the *shape* of the propagator's stress kernel cost, bytes being equal?
// 8 arrays of one padded field (5 read, 3 read-modify-write), Infinity-Cache resident:
//   V0 float4 per lane, 256-thread blocks, grid-stride            (the plain streaming yardstick)
//   V1 float  per lane, 64-thread blocks, one cell per lane        (the propagator's shape, no stencil)
//   V2 V1 + the 4+4 stencil taps on two of the read arrays, XCD-banded block order
//   V3 float4 per lane, 64-thread blocks, 4 cells per lane, no stencil
//   V4 V3 + 4 z-taps (row-shifted float4 loads) on two of the read arrays, XCD-banded
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

struct Arr { float *p[8]; };
constexpr int P = 2112, ROWS = 1088;

__device__ __forceinline__ unsigned banded(unsigned bid, unsigned nb) { return (bid & 7u) * (nb >> 3) + (bid >> 3); }

__global__ void v0(Arr a, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 s = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < 5; r++) { float4 v = ((float4 *)a.p[r])[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
        for (int w = 5; w < 8; w++) { float4 v = ((float4 *)a.p[w])[i]; v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; ((float4 *)a.p[w])[i] = v; }
    }
}
template <bool STENCIL>
__global__ __launch_bounds__(64) void v12(Arr a, unsigned nb) {
    const unsigned b = STENCIL ? banded(blockIdx.x, nb) : blockIdx.x;
    const size_t i = (size_t)b * 64 + threadIdx.x;
    float s = 0.f;
    if (STENCIL) {
        const size_t lo = 2 * P + 2, hi = (size_t)ROWS * P - 2 * P - 2;
        if (i < lo || i >= hi) return;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float *f = a.p[r];
            s += 1.125f * (f[i] - f[i - P]) - 0.0417f * (f[i + P] - f[i - 2 * P]);
            s += 1.125f * (f[i + 1] - f[i]) - 0.0417f * (f[i + 2] - f[i - 1]);
        }
    } else {
        s = a.p[0][i] + a.p[1][i];
    }
#pragma unroll
    for (int r = 2; r < 5; r++) s += a.p[r][i];
#pragma unroll
    for (int w = 5; w < 8; w++) a.p[w][i] += s;
}
template <bool STENCIL>
__global__ __launch_bounds__(64) void v34(Arr a, unsigned nb) {
    const unsigned b = STENCIL ? banded(blockIdx.x, nb) : blockIdx.x;
    const size_t i = (size_t)b * 64 + threadIdx.x;  // float4 index
    float4 s = make_float4(0, 0, 0, 0);
    const size_t P4 = P / 4;
    if (STENCIL) {
        const size_t lo = 2 * P4, hi = (size_t)ROWS * P4 - 2 * P4;
        if (i < lo || i >= hi) return;
#pragma unroll
        for (int r = 0; r < 2; r++) {
            const float4 *f = (const float4 *)a.p[r];
            const float4 c = f[i], m1 = f[i - P4], p1 = f[i + P4], m2 = f[i - 2 * P4];
            s.x += 1.125f * (c.x - m1.x) - 0.0417f * (p1.x - m2.x);
            s.y += 1.125f * (c.y - m1.y) - 0.0417f * (p1.y - m2.y);
            s.z += 1.125f * (c.z - m1.z) - 0.0417f * (p1.z - m2.z);
            s.w += 1.125f * (c.w - m1.w) - 0.0417f * (p1.w - m2.w);
        }
    } else {
        float4 u = ((float4 *)a.p[0])[i], v = ((float4 *)a.p[1])[i];
        s.x = u.x + v.x; s.y = u.y + v.y; s.z = u.z + v.z; s.w = u.w + v.w;
    }
#pragma unroll
    for (int r = 2; r < 5; r++) { float4 v = ((float4 *)a.p[r])[i]; s.x += v.x; s.y += v.y; s.z += v.z; s.w += v.w; }
#pragma unroll
    for (int w = 5; w < 8; w++) { float4 v = ((float4 *)a.p[w])[i]; v.x += s.x; v.y += s.y; v.z += s.z; v.w += s.w; ((float4 *)a.p[w])[i] = v; }
}

#include <cstdlib>
int main(int argc, char **argv) {
    const size_t n = (size_t)ROWS * P;
    Arr a;
    const bool random_data = argc > 1;   // any argument: arrays start from random floats instead of zeros (is the fabric data-dependent?)
    std::vector<float> init(n);
    for (size_t i = 0; i < n; i++) init[i] = random_data ? (float)rand() / RAND_MAX - 0.5f : 0.0f;
    for (auto &p : a.p) { CK(hipMalloc((void **)&p, n * 4)); CK(hipMemcpy(p, init.data(), n * 4, hipMemcpyHostToDevice)); }
    printf("arrays start from %s\n", random_data ? "random floats" : "zeros");
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const int reps = 300;
    const unsigned nb1 = (unsigned)(n / 64), nb4 = (unsigned)(n / 256);
    for (int v = 0; v < 5; v++) {
        for (int r = -5; r < reps; r++) {
            if (r == 0) CK(hipEventRecord(e0, 0));
            switch (v) {
                case 0: hipLaunchKernelGGL(v0, dim3(2048), dim3(256), 0, 0, a, n / 4); break;
                case 1: hipLaunchKernelGGL(v12<false>, dim3(nb1), dim3(64), 0, 0, a, nb1); break;
                case 2: hipLaunchKernelGGL(v12<true>, dim3(nb1), dim3(64), 0, 0, a, nb1); break;
                case 3: hipLaunchKernelGGL(v34<false>, dim3(nb4), dim3(64), 0, 0, a, nb4); break;
                case 4: hipLaunchKernelGGL(v34<true>, dim3(nb4), dim3(64), 0, 0, a, nb4); break;
            }
        }
        CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / reps, bytes = (double)n * 4 * 11;
        printf("V%d : %7.2f us/launch  %5.2f TB/s of the 11 compulsory array passes (%.1f MB)\n", v, us, bytes / us / 1e6, bytes / 1e6);
    }
    return 0;
}
