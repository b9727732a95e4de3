// bw_probe.hip -- calibration: how fast can gfx950 stream arrays of the propagator's size?
// Working sets: N arrays of `elems` floats; kernel reads R of them and read-modify-writes W of them.
// Small sets (<= 256 MiB) live in the Infinity Cache between launches, large ones come from HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <int R, int W>
__global__ void k_stream(float4 *const *__restrict__ arrs, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; r++) { float4 v = arrs[r][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < W; w++) { float4 v = arrs[R + w][i]; v.x += acc.x; v.y += acc.y * 0.5f; v.z += acc.z; v.w += acc.w; arrs[R + w][i] = v; }
    }
}

template <int R, int O>
__global__ void k_stream_ro(float4 *const *__restrict__ arrs, size_t n4) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 acc = make_float4(0, 0, 0, 0);
#pragma unroll
        for (int r = 0; r < R; r++) { float4 v = arrs[r][i]; acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w; }
#pragma unroll
        for (int w = 0; w < O; w++) { arrs[R + w][i] = acc; }
    }
}

// fully fused backward step model: reads R arrays, writes O other arrays (double-buffered fields)
template <int R, int O>
int run_ro(const char *name, size_t elems, int reps, int blocks) {
    std::vector<float4 *> h(R + O);
    for (auto &p : h) { CK(hipMalloc((void **)&p, elems * sizeof(float))); CK(hipMemset(p, 0, elems * sizeof(float))); }
    float4 **d; CK(hipMalloc((void **)&d, sizeof(float4 *) * (R + O)));
    CK(hipMemcpy(d, h.data(), sizeof(float4 *) * (R + O), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_stream_ro<R, O>), dim3(blocks), dim3(256), 0, 0, d, elems / 4);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream_ro<R, O>), dim3(blocks), dim3(256), 0, 0, d, elems / 4);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)elems * 4.0 * (R + O);
    printf("%-34s arrays %2d x %7.1f MB  set %8.1f MB  blocks %5d : %7.2f us/launch  %6.2f TB/s (read %d + write %d)\n", name, R + O,
           elems * 4.0 / 1e6, (R + O) * elems * 4.0 / 1e6, blocks, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, R, O);
    for (auto p : h) CK(hipFree(p));
    CK(hipFree(d));
    return 0;
}

template <int R, int W>
int run(const char *name, size_t elems, int reps, int blocks) {
    std::vector<float4 *> h(R + W);
    for (auto &p : h) { CK(hipMalloc((void **)&p, elems * sizeof(float))); CK(hipMemset(p, 0, elems * sizeof(float))); }
    float4 **d; CK(hipMalloc((void **)&d, sizeof(float4 *) * (R + W)));
    CK(hipMemcpy(d, h.data(), sizeof(float4 *) * (R + W), hipMemcpyHostToDevice));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int i = 0; i < 5; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(blocks), dim3(256), 0, 0, d, elems / 4);
    CK(hipEventRecord(a, 0));
    for (int i = 0; i < reps; i++) hipLaunchKernelGGL((k_stream<R, W>), dim3(blocks), dim3(256), 0, 0, d, elems / 4);
    CK(hipEventRecord(b, 0)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    double bytes = (double)elems * 4.0 * (R + 2.0 * W);
    printf("%-34s arrays %2d x %7.1f MB  set %8.1f MB  blocks %5d : %7.2f us/launch  %6.2f TB/s (read %d + rmw %d)\n", name, R + W,
           elems * 4.0 / 1e6, (R + W) * elems * 4.0 / 1e6, blocks, ms * 1e3 / reps, bytes / (ms * 1e-3 / reps) / 1e12, R, W);
    for (auto p : h) CK(hipFree(p));
    CK(hipFree(d));
    return 0;
}

int main() {
    const size_t grid = 1088ull * 2112ull;  // one padded field of the 2000x1000 problem (pitch 2112)
    for (int blocks : {1024, 2048, 4096, 8192}) {
        run<5, 3>("stress-like  (5 read, 3 rmw)", grid, 300, blocks);
    }
    run<5, 2>("velocity-like (5 read, 2 rmw)", grid, 300, 2048);
    run<5, 5>("fused-fwd-like (5 read, 5 rmw)", grid, 300, 2048);
    run<9, 9>("fused-bwd-like (9 read, 9 rmw)", grid, 200, 2048);
    run_ro<10, 5>("fwd ping-pong (10 read, 5 write)", grid, 300, 2048);
    run_ro<19, 10>("bwd ping-pong 267MB (19 r, 10 w)", grid, 200, 2048);
    run_ro<15, 5>("184 MB set (15 r, 5 w)", grid, 200, 2048);
    run_ro<17, 8>("230 MB set (17 r, 8 w)", grid, 200, 2048);
    run<1, 1>("copy-ish small (1 read, 1 rmw)", grid, 300, 2048);
    run<1, 1>("copy-ish 1 GiB arrays (HBM)", 256ull << 20, 20, 4096);
    run<5, 3>("stress-like 256 MiB arrays (HBM)", 64ull << 20, 20, 4096);
    return 0;
}
