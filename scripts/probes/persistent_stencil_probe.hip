// persistent_stencil_probe.hip -- de-risking probe for the register-resident persistent time loop.
// A leapfrog of two fields (u, w) on a 1024-row x 2048-column grid, 4th-order taps in z and x:
//     phase A:  w += c * (Dz-(u) + Dx+(u))          phase B:  u += c * (Dz+(w) + Dx-(w))
// (a) reference: two plain kernels per step, fields in global memory;
// (b) persistent: ONE launch for all steps; 256 workgroups x 1024 threads, one per CU; each owns 4 rows and keeps them in
//     registers; x-neighbours through LDS, z-halos (2 rows up / down) through global halo buffers handed to the two
//     row-neighbours with release / acquire flags at agent scope.  Every spin is bounded (abort flag), so the grid
//     always drains.
// Prints microseconds per time step of both and the maximum deviation of the final fields.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

constexpr int R = 4, NCOL = 2, T = 1024, W = T * NCOL, NB = 256, NZ = NB * R;
constexpr float C1 = 9.0f / 8.0f, C2 = 1.0f / 24.0f, CC = 0.05f;

__device__ __forceinline__ float dminus(float m2, float m1, float c0, float p1) { return C1 * (c0 - m1) - C2 * (p1 - m2); }
__device__ __forceinline__ float dplus(float m1, float c0, float p1, float p2) { return C1 * (p1 - c0) - C2 * (p2 - m1); }
__device__ __forceinline__ float at(const float *f, int z, int x) { return (z < 0 || z >= NZ || x < 0 || x >= W) ? 0.0f : f[(size_t)z * W + x]; }

// ---------------- (a) reference: plain kernels ----------------
__global__ void ref_A(const float *__restrict__ u, float *__restrict__ w) {
    const int x = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
    const float dz = dminus(at(u, z - 2, x), at(u, z - 1, x), at(u, z, x), at(u, z + 1, x));
    const float dx = dplus(at(u, z, x - 1), at(u, z, x), at(u, z, x + 1), at(u, z, x + 2));
    w[(size_t)z * W + x] += CC * (dz + dx);
}
__global__ void ref_B(float *__restrict__ u, const float *__restrict__ w) {
    const int x = blockIdx.x * 256 + threadIdx.x, z = blockIdx.y;
    const float dz = dplus(at(w, z - 1, x), at(w, z, x), at(w, z + 1, x), at(w, z + 2, x));
    const float dx = dminus(at(w, z, x - 2), at(w, z, x - 1), at(w, z, x), at(w, z, x + 1));
    u[(size_t)z * W + x] += CC * (dz + dx);
}

// ---------------- (b) persistent ----------------
__device__ __forceinline__ int flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// wait until both neighbours' flags reached `want` (bounded); returns false if the run was aborted
__device__ __forceinline__ bool wait_neighbours(const int *flag, int b, int want, int *abort_flag) {
    __shared__ int ok;
    if (threadIdx.x == 0) {
        int good = 1;
        for (int nb = b - 1; nb <= b + 1; nb += 2) {
            if (nb < 0 || nb >= NB) continue;
            long spins = 0;
            while (flag_load(flag + nb * 32) < want) {   // flags on lines of their own
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1L << 22) || flag_load(abort_flag)) { good = 0; flag_store(abort_flag, 1); break; }
            }
            if (!good) break;
        }
        ok = good;
    }
    __syncthreads();
    return ok != 0;   // no acquire fence: every payload load below is an sc1 (agent-scope) load
}

// payload: sc1 write-through stores / sc1 loads instead of release / acquire fences (which write back / invalidate the L2)
__device__ __forceinline__ void pay_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float pay_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// halo buffers: [band][side 0 = top rows 0..1, 1 = bottom rows R-2..R-1][2][W]
__device__ __forceinline__ size_t halo_off(int band, int side, int row) { return (((size_t)band * 2 + side) * 2 + row) * W; }

__device__ __forceinline__ void publish(float *halo, int b, const float (&f)[R][NCOL], int *flag, int step) {
    for (int c = 0; c < NCOL; c++) {
        const int x = threadIdx.x + c * T;
        pay_store(halo + halo_off(b, 0, 0) + x, f[0][c]);
        pay_store(halo + halo_off(b, 0, 1) + x, f[1][c]);
        pay_store(halo + halo_off(b, 1, 0) + x, f[R - 2][c]);
        pay_store(halo + halo_off(b, 1, 1) + x, f[R - 1][c]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // write-through (sc1) stores drained before the flag (MI355X_MICROARCH.md hand-off rules)
    __syncthreads();
    if (threadIdx.x == 0) flag_store(flag + b * 32, step);
}

__global__ __launch_bounds__(T) void persistent(float *__restrict__ U, float *__restrict__ Wf, float *haloU, float *haloW, int *flagA,
                                                int *flagB, int steps, int *abort_flag) {
    // band order: consecutive bands share an XCD (blockIdx % 8 = XCD under round-robin dispatch)
    const int b = (blockIdx.x & 7) * (NB / 8) + (blockIdx.x >> 3);
    __shared__ float lds[R][W + 8];
    float u[R][NCOL], w[R][NCOL];
    for (int r = 0; r < R; r++)
        for (int c = 0; c < NCOL; c++) {
            u[r][c] = U[(size_t)(b * R + r) * W + threadIdx.x + c * T];
            w[r][c] = Wf[(size_t)(b * R + r) * W + threadIdx.x + c * T];
        }
    for (int s = 1; s <= steps; s++) {
        // ---------------- phase A: w += c (Dz-(u) + Dx+(u)); u of step s-1 from the neighbours
        if (s > 1 && !wait_neighbours(flagB, b, s - 1, abort_flag)) break;
        float top[2][NCOL], bot[2][NCOL];   // rows -2,-1 (bottom rows of band b-1) and rows R, R+1 (top rows of band b+1)
        for (int c = 0; c < NCOL; c++) {
            const int x = threadIdx.x + c * T;
            top[0][c] = b > 0 ? pay_load(haloU + halo_off(b - 1, 1, 0) + x) : 0.0f;
            top[1][c] = b > 0 ? pay_load(haloU + halo_off(b - 1, 1, 1) + x) : 0.0f;
            bot[0][c] = b < NB - 1 ? pay_load(haloU + halo_off(b + 1, 0, 0) + x) : 0.0f;
            bot[1][c] = b < NB - 1 ? pay_load(haloU + halo_off(b + 1, 0, 1) + x) : 0.0f;
        }
        for (int r = 0; r < R; r++)
            for (int c = 0; c < NCOL; c++) lds[r][4 + threadIdx.x + c * T] = u[r][c];
        if (threadIdx.x < 4) for (int r = 0; r < R; r++) { lds[r][threadIdx.x] = 0.0f; lds[r][4 + W + threadIdx.x] = 0.0f; }
        __syncthreads();
        for (int r = 0; r < R; r++)
            for (int c = 0; c < NCOL; c++) {
                const int x = 4 + threadIdx.x + c * T;
                const float m2 = r >= 2 ? u[r - 2][c] : top[r][c];
                const float m1 = r >= 1 ? u[r - 1][c] : top[1][c];
                const float p1 = r + 1 < R ? u[r + 1][c] : bot[0][c];
                const float dz = dminus(m2, m1, u[r][c], p1);
                const float dx = dplus(lds[r][x - 1], lds[r][x], lds[r][x + 1], lds[r][x + 2]);
                w[r][c] += CC * (dz + dx);
            }
        __syncthreads();
        publish(haloW, b, w, flagA, s);
        // ---------------- phase B: u += c (Dz+(w) + Dx-(w)); w of this step from the neighbours
        if (!wait_neighbours(flagA, b, s, abort_flag)) break;
        for (int c = 0; c < NCOL; c++) {
            const int x = threadIdx.x + c * T;
            top[0][c] = b > 0 ? pay_load(haloW + halo_off(b - 1, 1, 0) + x) : 0.0f;
            top[1][c] = b > 0 ? pay_load(haloW + halo_off(b - 1, 1, 1) + x) : 0.0f;
            bot[0][c] = b < NB - 1 ? pay_load(haloW + halo_off(b + 1, 0, 0) + x) : 0.0f;
            bot[1][c] = b < NB - 1 ? pay_load(haloW + halo_off(b + 1, 0, 1) + x) : 0.0f;
        }
        for (int r = 0; r < R; r++)
            for (int c = 0; c < NCOL; c++) lds[r][4 + threadIdx.x + c * T] = w[r][c];
        __syncthreads();
        for (int r = 0; r < R; r++)
            for (int c = 0; c < NCOL; c++) {
                const int x = 4 + threadIdx.x + c * T;
                const float m1 = r >= 1 ? w[r - 1][c] : top[1][c];
                const float p1 = r + 1 < R ? w[r + 1][c] : bot[0][c];
                const float p2 = r + 2 < R ? w[r + 2][c] : bot[r + 2 - R][c];
                const float dz = dplus(m1, w[r][c], p1, p2);
                const float dx = dminus(lds[r][x - 2], lds[r][x - 1], lds[r][x], lds[r][x + 1]);
                u[r][c] += CC * (dz + dx);
            }
        __syncthreads();
        publish(haloU, b, u, flagB, s);
    }
    for (int r = 0; r < R; r++)
        for (int c = 0; c < NCOL; c++) {
            U[(size_t)(b * R + r) * W + threadIdx.x + c * T] = u[r][c];
            Wf[(size_t)(b * R + r) * W + threadIdx.x + c * T] = w[r][c];
        }
}

int main(int argc, char **argv) {
    const int steps = argc > 1 ? atoi(argv[1]) : 500;
    const size_t n = (size_t)NZ * W;
    std::vector<float> hu(n), hw(n, 0.0f);
    for (size_t i = 0; i < n; i++) {
        const int z = (int)(i / W), x = (int)(i % W);
        hu[i] = expf(-((z - 500.f) * (z - 500.f) + (x - 900.f) * (x - 900.f)) / 2000.f);
    }
    float *U0, *W0, *U1, *W1, *haloU, *haloW;
    int *flags;
    CK(hipMalloc((void **)&U0, n * 4)); CK(hipMalloc((void **)&W0, n * 4)); CK(hipMalloc((void **)&U1, n * 4)); CK(hipMalloc((void **)&W1, n * 4));
    const size_t hn = (size_t)NB * 4 * W;
    CK(hipMalloc((void **)&haloU, hn * 4)); CK(hipMalloc((void **)&haloW, hn * 4)); CK(hipMalloc((void **)&flags, (2 * NB * 32 + 32) * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float ms;
    // (a)
    CK(hipMemcpy(U0, hu.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W0, hw.data(), n * 4, hipMemcpyHostToDevice));
    CK(hipEventRecord(e0, 0));
    for (int s = 0; s < steps; s++) {
        hipLaunchKernelGGL(ref_A, dim3(W / 256, NZ), dim3(256), 0, 0, U0, W0);
        hipLaunchKernelGGL(ref_B, dim3(W / 256, NZ), dim3(256), 0, 0, U0, W0);
    }
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    printf("plain kernels : %8.2f us per step (2 launches, 2 fields of %.1f MB)\n", ms * 1e3 / steps, n * 4 / 1e6);
    // (b)
    CK(hipMemcpy(U1, hu.data(), n * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(W1, hw.data(), n * 4, hipMemcpyHostToDevice));
    std::vector<float> hh(hn);
    for (int b = 0; b < NB; b++)
        for (int side = 0; side < 2; side++)
            for (int r = 0; r < 2; r++)
                for (int x = 0; x < W; x++) hh[(((size_t)b * 2 + side) * 2 + r) * W + x] = hu[(size_t)(b * R + (side ? R - 2 + r : r)) * W + x];
    CK(hipMemcpy(haloU, hh.data(), hn * 4, hipMemcpyHostToDevice)); CK(hipMemset(haloW, 0, hn * 4)); CK(hipMemset(flags, 0, (2 * NB * 32 + 32) * 4));
    int *flagA = flags, *flagB = flags + NB * 32, *abort_flag = flags + 2 * NB * 32;
    int occ = 0;
    CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, persistent, T, 0));
    hipDeviceProp_t prop; CK(hipGetDeviceProperties(&prop, 0));
    if (occ < 1 || prop.multiProcessorCount < NB) { printf("not co-resident: occupancy %d, CUs %d\n", occ, prop.multiProcessorCount); return 1; }
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(persistent, dim3(NB), dim3(T), 0, 0, U1, W1, haloU, haloW, flagA, flagB, steps, abort_flag);
    CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));
    int aborted = 0; CK(hipMemcpy(&aborted, abort_flag, 4, hipMemcpyDeviceToHost));
    printf("persistent    : %8.2f us per step (1 launch for %d steps, 2 hand-offs per step)%s\n", ms * 1e3 / steps, steps, aborted ? "  ABORTED (spin limit)" : "");
    std::vector<float> a(n), b2(n);
    CK(hipMemcpy(a.data(), U0, n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b2.data(), U1, n * 4, hipMemcpyDeviceToHost));
    double md = 0, mx = 0;
    for (size_t i = 0; i < n; i++) { md = fmax(md, fabs((double)a[i] - b2[i])); mx = fmax(mx, fabs((double)a[i])); }
    printf("max |u_persistent - u_plain| = %.3e (max |u| %.3e)\n", md, mx);
    return 0;
}
