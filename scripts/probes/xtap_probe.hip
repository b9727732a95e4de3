// xtap_probe.hip -- do wavefront shuffles pay for the x-neighbour taps of the 4th-order stencil on gfx950?
// (north_star: "LDS halo staging and wavefront shuffles"; DESIGN.md 3.1 records the answer.)
//
// A forward-velocity-shaped update on propagator-sized arrays (1068 rows x 2112 floats, Infinity-Cache resident like the
// real kernels): read 3 fields through 4 z-taps + 4 x-taps each, read-modify-write 2 fields, one wave = 64 consecutive x of
// one row, blocks of 2 rows, XCD-banded tile order -- the product kernels' shape.
//   variant 0: every tap is a global load (shifted loads hit the same lines in the vector L1)      [what the product does]
//   variant 1: the x-taps of a wave come from its own registers through ds_bpermute (__shfl); only the 2+2 edge lanes load
//   variant 2: x-taps through DPP wave shifts (wave_shr / wave_shl, 1 and 2 lanes), edge lanes load
// Prints microseconds per launch (mean of the back half of 400 launches) and the checksum of the result (equal for all).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -o xtap_probe xtap_probe.hip && ./xtap_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define OK(x)                                                                 \
    do {                                                                      \
        hipError_t e = (x);                                                   \
        if (e != hipSuccess) {                                                \
            printf("HIP error %s at line %d\n", hipGetErrorString(e), __LINE__); \
            return 1;                                                         \
        }                                                                     \
    } while (0)

constexpr int NZ = 1064, NX = 2064, P = 2112;
constexpr float C1 = 9.0f / 8.0f, C2 = 1.0f / 24.0f;

__device__ __forceinline__ float dminus(float m2, float m1, float c, float p1) { return C1 * (c - m1) - C2 * (p1 - m2); }
__device__ __forceinline__ float dplus(float m1, float c, float p1, float p2) { return C1 * (p1 - c) - C2 * (p2 - m1); }

// value of the wave's row at x + D for every lane: from lane + D inside the wave, from memory for the |D| edge lanes
template <int VAR, int D>
__device__ __forceinline__ float xtap(const float *__restrict__ row, int x, float own) {
    const int lane = threadIdx.x & 63;
    const bool legal = x + D >= 0 && x + D < P;
    if constexpr (VAR == 0) {
        return legal ? row[x + D] : 0.0f;
    } else if constexpr (VAR == 1) {
        float v = __shfl(own, lane + D, 64);
        if (lane + D < 0 || lane + D > 63) v = legal ? row[x + D] : 0.0f;
        return v;
    } else {
        float v;
        if constexpr (D == -1) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x138, 0xf, 0xf, false));  // wave_shr:1
        else if constexpr (D == 1) v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x130, 0xf, 0xf, false));  // wave_shl:1
        else if constexpr (D == -2) {
            int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x138, 0xf, 0xf, false);
            v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0x138, 0xf, 0xf, false));
        } else {
            int t = __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, own), 0x130, 0xf, 0xf, false);
            v = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, t, 0x130, 0xf, 0xf, false));
        }
        if (lane + D < 0 || lane + D > 63) v = legal ? row[x + D] : 0.0f;
        return v;
    }
}

template <int VAR>
__global__ __launch_bounds__(1024) void k_update(const float *__restrict__ szz, const float *__restrict__ sxz, const float *__restrict__ sxx,
                                                 const float *__restrict__ rho, float *__restrict__ vz, float *__restrict__ vx, int gx, int gy) {
    int t = blockIdx.x;
    const int per = (gx * gy + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    const int ty = t / gx, tx = t - ty * gx;
    const int x = tx * 64 + (threadIdx.x & 63);
    const int z = __builtin_amdgcn_readfirstlane(ty * 2 + (int)(threadIdx.x >> 6));
    if (ty >= gy || z < 2 || z > NZ - 3) return;
    const bool on = x >= 2 && x <= NX - 3;
    const size_t i = (size_t)z * P + x;
    const int xc = x;  // every lane loads its own column (x < P always): the shuffles need all 64 values
    const size_t ic = i;
    const float szz0 = szz[ic], sxz0 = sxz[ic], sxx0 = sxx[ic];
    const float dszz_dz = dplus(szz[ic - P], szz0, szz[ic + P], szz[ic + 2 * P]);
    const float dsxz_dz = dminus(sxz[ic - 2 * P], sxz[ic - P], sxz0, sxz[ic + P]);
    const float *rz = sxz + (size_t)z * P, *rx = sxx + (size_t)z * P;
    const float dsxz_dx = dminus(xtap<VAR, -2>(rz, xc, sxz0), xtap<VAR, -1>(rz, xc, sxz0), sxz0, xtap<VAR, 1>(rz, xc, sxz0));
    const float dsxx_dx = dplus(xtap<VAR, -1>(rx, xc, sxx0), sxx0, xtap<VAR, 1>(rx, xc, sxx0), xtap<VAR, 2>(rx, xc, sxx0));
    const float r0 = rho[ic];
    const float ba = 2.0f / (rho[ic + P] + r0), bb = 2.0f / (xtap<VAR, 1>(rho + (size_t)z * P, xc, r0) + r0);
    if (!on) return;
    vz[i] = vz[i] + (dszz_dz + dsxz_dx) * ba * 1e-3f;
    vx[i] = vx[i] + (dsxz_dz + dsxx_dx) * bb * 1e-3f;
}

int main() {
    const size_t n = (size_t)(NZ + 4) * P;
    float *d[6];
    std::vector<float> h(n);
    for (int k = 0; k < 6; k++) {
        OK(hipMalloc((void **)&d[k], n * sizeof(float)));
        unsigned s = 12345u + k;
        for (size_t i = 0; i < n; i++) {  // real-looking data: the memory system is data-dependent
            s = s * 1664525u + 1013904223u;
            h[i] = (k == 3 ? 2000.0f : 0.0f) + (float)(s >> 8) / 16777216.0f;
        }
        OK(hipMemcpy(d[k], h.data(), n * sizeof(float), hipMemcpyHostToDevice));
    }
    const int gx = (NX + 63) / 64, gy = (NZ + 1) / 2;
    const int nb = ((gx * gy + 7) / 8) * 8;
    hipEvent_t e0, e1;
    OK(hipEventCreate(&e0));
    OK(hipEventCreate(&e1));
    for (int var = 0; var < 3; var++) {
        for (int k = 4; k < 6; k++) OK(hipMemset(d[k], 0, n * sizeof(float)));
        float ms = 0;
        for (int rep = 0; rep < 400; rep++) {
            if (rep == 200) OK(hipEventRecord(e0, 0));
            if (var == 0) hipLaunchKernelGGL(k_update<0>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
            if (var == 1) hipLaunchKernelGGL(k_update<1>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
            if (var == 2) hipLaunchKernelGGL(k_update<2>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
        }
        OK(hipEventRecord(e1, 0));
        OK(hipEventSynchronize(e1));
        OK(hipEventElapsedTime(&ms, e0, e1));
        OK(hipMemcpy(h.data(), d[4], n * sizeof(float), hipMemcpyDeviceToHost));
        double cs = 0;
        for (size_t i = 0; i < n; i++) cs += h[i];
        OK(hipMemcpy(h.data(), d[5], n * sizeof(float), hipMemcpyDeviceToHost));
        for (size_t i = 0; i < n; i++) cs += 3.0 * h[i];
        printf("variant %d (%s): %.2f us per launch, checksum %.9e\n", var,
               var == 0 ? "all taps global loads" : var == 1 ? "x-taps by ds_bpermute" : "x-taps by DPP wave shifts", 1e3 * ms / 200.0, cs);
    }
    return 0;
}
