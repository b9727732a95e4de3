// xtap_probe.hip -- do wavefront shuffles pay for the x-neighbour taps of the 4th-order stencil on gfx950?
// (north_star: "LDS halo staging and wavefront shuffles"; DESIGN.md 3.1 records the answer.)
//
// A forward-velocity-shaped update on propagator-sized arrays (1068 rows x 2112 floats, Infinity-Cache resident like the
// real kernels): read 3 fields through 4 z-taps + 4 x-taps each, read-modify-write 2 fields, one wave = 64 consecutive x of
// one row, blocks of 2 rows, XCD-banded tile order -- the product kernels' shape.
//   variant 0: every tap is a global load (shifted loads hit the same lines in the vector L1)      [what the product does]
//   variant 1: the x-taps of a wave come from its own registers through ds_bpermute (__shfl); only the 2+2 edge lanes load
//   variant 2: x-taps through DPP wave shifts (wave_shr / wave_shl, 1 and 2 lanes), edge lanes load
//   variant 3: FOUR columns per lane (16-byte loads, a wave covers 256 columns): a quarter of the vector-memory instructions,
//              x-taps mostly from the lane's own registers, 1-2 halo values per lane and array by (cached) loads
//   variant 4: LDS halo staging of the z-direction: blocks of 8 rows stage their rows of szz / sxz (+ 5 halo rows) in LDS
//              once, one barrier, every z-tap is an LDS read (the x-taps stay shifted loads)
//   variants 5, 6 (round 4): REGISTER TILING IN Z -- a wave owns R = 2 / 4 consecutive rows of its 64 columns, so the z-taps of
//              neighbouring rows are loaded once (szz: R + 3 loads instead of 4 R; sxz: R + 3 + 3 R instead of 7 R): 17 % / 26 % fewer
//              loads per cell at R / 2 ... R / 4 of the waves
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

// variant 3: four consecutive columns per lane
__global__ __launch_bounds__(1024) void k_update4(const float *__restrict__ szz, const float *__restrict__ sxz, const float *__restrict__ sxx,
                                                  const float *__restrict__ rho, float *__restrict__ vz, float *__restrict__ vx, int gx4, int gy) {
    int t = blockIdx.x;
    const int per = (gx4 * gy + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    const int ty = t / gx4, tx = t - ty * gx4;
    const int x = tx * 256 + 4 * (threadIdx.x & 63);
    const int z = __builtin_amdgcn_readfirstlane(ty * 2 + (int)(threadIdx.x >> 6));
    if (ty >= gy || z < 2 || z > NZ - 3 || x >= P) return;
    const size_t i = (size_t)z * P + x;
    auto ld4 = [](const float *p) { return *reinterpret_cast<const float4 *>(p); };
    const float4 zz_m1 = ld4(szz + i - P), zz_0 = ld4(szz + i), zz_p1 = ld4(szz + i + P), zz_p2 = ld4(szz + i + 2 * P);
    const float4 xz_m2 = ld4(sxz + i - 2 * P), xz_m1 = ld4(sxz + i - P), xz_0 = ld4(sxz + i), xz_p1 = ld4(sxz + i + P);
    const float4 xx_0 = ld4(sxx + i), r_0 = ld4(rho + i), r_p = ld4(rho + i + P);
    const bool left = x >= 2, right = x + 5 < P;
    const float xz_l2 = left ? sxz[i - 2] : 0.f, xz_l1 = left ? sxz[i - 1] : 0.f, xz_r = right ? sxz[i + 4] : 0.f;
    const float xx_l1 = left ? sxx[i - 1] : 0.f, xx_r1 = right ? sxx[i + 4] : 0.f, xx_r2 = right ? sxx[i + 5] : 0.f;
    const float r_r = right ? rho[i + 4] : 1.f;
    float4 v_z = ld4(vz + i), v_x = ld4(vx + i);
    const float a_xz[7] = {xz_l2, xz_l1, xz_0.x, xz_0.y, xz_0.z, xz_0.w, xz_r};
    const float a_xx[7] = {xx_l1, xx_0.x, xx_0.y, xx_0.z, xx_0.w, xx_r1, xx_r2};
    const float a_r[5] = {r_0.x, r_0.y, r_0.z, r_0.w, r_r};
    const float zzm1[4] = {zz_m1.x, zz_m1.y, zz_m1.z, zz_m1.w}, zz0[4] = {zz_0.x, zz_0.y, zz_0.z, zz_0.w}, zzp1[4] = {zz_p1.x, zz_p1.y, zz_p1.z, zz_p1.w},
                zzp2[4] = {zz_p2.x, zz_p2.y, zz_p2.z, zz_p2.w};
    const float xzm2[4] = {xz_m2.x, xz_m2.y, xz_m2.z, xz_m2.w}, xzm1[4] = {xz_m1.x, xz_m1.y, xz_m1.z, xz_m1.w}, xzp1[4] = {xz_p1.x, xz_p1.y, xz_p1.z, xz_p1.w};
    const float rp[4] = {r_p.x, r_p.y, r_p.z, r_p.w};
    float oz[4] = {v_z.x, v_z.y, v_z.z, v_z.w}, ox[4] = {v_x.x, v_x.y, v_x.z, v_x.w};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const int xc = x + k;
        if (xc < 2 || xc > NX - 3) continue;
        const float dszz_dz = dplus(zzm1[k], zz0[k], zzp1[k], zzp2[k]);
        const float dsxz_dz = dminus(xzm2[k], xzm1[k], a_xz[k + 2], xzp1[k]);
        const float dsxz_dx = dminus(a_xz[k], a_xz[k + 1], a_xz[k + 2], a_xz[k + 3]);
        const float dsxx_dx = dplus(a_xx[k], a_xx[k + 1], a_xx[k + 2], a_xx[k + 3]);
        const float ba = 2.0f / (rp[k] + a_r[k]), bb = 2.0f / (a_r[k + 1] + a_r[k]);
        oz[k] = oz[k] + (dszz_dz + dsxz_dx) * ba * 1e-3f;
        ox[k] = ox[k] + (dsxz_dz + dsxx_dx) * bb * 1e-3f;
    }
    *reinterpret_cast<float4 *>(vz + i) = make_float4(oz[0], oz[1], oz[2], oz[3]);
    *reinterpret_cast<float4 *>(vx + i) = make_float4(ox[0], ox[1], ox[2], ox[3]);
}

// variant 4: z-taps from an LDS tile of 8 rows + halo rows
__global__ __launch_bounds__(512) void k_update_lds(const float *__restrict__ szz, const float *__restrict__ sxz, const float *__restrict__ sxx,
                                                    const float *__restrict__ rho, float *__restrict__ vz, float *__restrict__ vx, int gx, int gy8) {
    __shared__ float t_zz[11][64], t_xz[11][64];  // rows z0-1 .. z0+9 of szz, rows z0-2 .. z0+8 of sxz
    int t = blockIdx.x;
    const int per = (gx * gy8 + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    const int ty = t / gx, tx = t - ty * gx;
    if (ty >= gy8) return;  // whole block
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int x = tx * 64 + lane, z0 = ty * 8;
    const int z = __builtin_amdgcn_readfirstlane(z0 + w);
    auto row = [&](const float *a, int zz) { return (zz >= 0 && zz < NZ + 4) ? a[(size_t)zz * P + x] : 0.0f; };
    t_zz[w + 1][lane] = row(szz, z);
    t_xz[w + 2][lane] = row(sxz, z);
    if (w == 0) { t_zz[0][lane] = row(szz, z0 - 1); t_xz[0][lane] = row(sxz, z0 - 2); }
    if (w == 1) t_xz[1][lane] = row(sxz, z0 - 1);
    if (w == 2) t_zz[9][lane] = row(szz, z0 + 8);
    if (w == 3) t_zz[10][lane] = row(szz, z0 + 9);
    if (w == 4) t_xz[10][lane] = row(sxz, z0 + 8);
    const bool on = z >= 2 && z <= NZ - 3 && x >= 2 && x <= NX - 3;
    const size_t i = (size_t)z * P + x;
    float sxx_m1 = 0, sxx_0 = 0, sxx_p1 = 0, sxx_p2 = 0, xz_m2 = 0, xz_m1 = 0, xz_p1 = 0, r0 = 1, rz = 1, rx = 1, vz0 = 0, vx0 = 0;
    if (on) {  // everything that is not a z-tap: issued before the barrier so that it overlaps the staging
        sxx_m1 = sxx[i - 1]; sxx_0 = sxx[i]; sxx_p1 = sxx[i + 1]; sxx_p2 = sxx[i + 2];
        xz_m2 = sxz[i - 2]; xz_m1 = sxz[i - 1]; xz_p1 = sxz[i + 1];
        r0 = rho[i]; rz = rho[i + P]; rx = rho[i + 1];
        vz0 = vz[i]; vx0 = vx[i];
    }
    __syncthreads();
    if (!on) return;
    const float dszz_dz = dplus(t_zz[w][lane], t_zz[w + 1][lane], t_zz[w + 2][lane], t_zz[w + 3][lane]);
    const float dsxz_dz = dminus(t_xz[w][lane], t_xz[w + 1][lane], t_xz[w + 2][lane], t_xz[w + 3][lane]);
    const float dsxz_dx = dminus(xz_m2, xz_m1, t_xz[w + 2][lane], xz_p1);
    const float dsxx_dx = dplus(sxx_m1, sxx_0, sxx_p1, sxx_p2);
    const float ba = 2.0f / (rz + r0), bb = 2.0f / (rx + r0);
    vz[i] = vz0 + (dszz_dz + dsxz_dx) * ba * 1e-3f;
    vx[i] = vx0 + (dsxz_dz + dsxx_dx) * bb * 1e-3f;
}

// variants 5, 6: R rows per wave, z-taps shared in registers
template <int R>
__global__ __launch_bounds__(1024) void k_update_rows(const float *__restrict__ szz, const float *__restrict__ sxz, const float *__restrict__ sxx,
                                                      const float *__restrict__ rho, float *__restrict__ vz, float *__restrict__ vx, int gx, int gyR) {
    int t = blockIdx.x;
    const int per = (gx * gyR + 7) >> 3;
    t = (t & 7) * per + (t >> 3);
    const int ty = t / gx, tx = t - ty * gx;
    if (ty >= gyR) return;
    const int x = tx * 64 + (threadIdx.x & 63);
    const int z0 = __builtin_amdgcn_readfirstlane((ty * 2 + (int)(threadIdx.x >> 6)) * R);
    if (z0 > NZ - 3) return;
    auto row = [&](const float *a, int zz, int dx) { return (zz >= 0 && zz < NZ + 4 && x + dx >= 0 && x + dx < P) ? a[(size_t)zz * P + x + dx] : 0.0f; };
    float zz[R + 3], xz[R + 3], rr[R + 1];
#pragma unroll
    for (int k = 0; k < R + 3; k++) {
        zz[k] = row(szz, z0 - 1 + k, 0);   // rows z0-1 .. z0+R+1
        xz[k] = row(sxz, z0 - 2 + k, 0);   // rows z0-2 .. z0+R
    }
#pragma unroll
    for (int k = 0; k < R + 1; k++) rr[k] = row(rho, z0 + k, 0);
    float xzm2[R], xzm1[R], xzp1[R], xxm1[R], xx0[R], xxp1[R], xxp2[R], rx[R], ovz[R], ovx[R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int z = z0 + r;
        xzm2[r] = row(sxz, z, -2); xzm1[r] = row(sxz, z, -1); xzp1[r] = row(sxz, z, 1);
        xxm1[r] = row(sxx, z, -1); xx0[r] = row(sxx, z, 0); xxp1[r] = row(sxx, z, 1); xxp2[r] = row(sxx, z, 2);
        rx[r] = row(rho, z, 1);
        ovz[r] = row(vz, z, 0); ovx[r] = row(vx, z, 0);
    }
    if (x < 2 || x > NX - 3) return;
#pragma unroll
    for (int r = 0; r < R; r++) {
        const int z = z0 + r;
        if (z < 2 || z > NZ - 3) continue;
        const size_t i = (size_t)z * P + x;
        const float dszz_dz = dplus(zz[r], zz[r + 1], zz[r + 2], zz[r + 3]);
        const float dsxz_dz = dminus(xz[r], xz[r + 1], xz[r + 2], xz[r + 3]);
        const float dsxz_dx = dminus(xzm2[r], xzm1[r], xz[r + 2], xzp1[r]);
        const float dsxx_dx = dplus(xxm1[r], xx0[r], xxp1[r], xxp2[r]);
        const float ba = 2.0f / (rr[r + 1] + rr[r]), bb = 2.0f / (rx[r] + rr[r]);
        vz[i] = ovz[r] + (dszz_dz + dsxz_dx) * ba * 1e-3f;
        vx[i] = ovx[r] + (dsxz_dz + dsxx_dx) * bb * 1e-3f;
    }
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
    const int gx4 = (NX + 255) / 256, nb4 = ((gx4 * gy + 7) / 8) * 8;
    const int gy8 = (NZ + 7) / 8, nb8 = ((gx * gy8 + 7) / 8) * 8;
    const int gyR2 = (NZ + 3) / 4, nbR2 = ((gx * gyR2 + 7) / 8) * 8;      // 2 waves x 2 rows per block
    const int gyR4 = (NZ + 7) / 8, nbR4 = ((gx * gyR4 + 7) / 8) * 8;      // 2 waves x 4 rows per block
    for (int var = 0; var < 7; var++) {
        for (int k = 4; k < 6; k++) OK(hipMemset(d[k], 0, n * sizeof(float)));
        float ms = 0;
        for (int rep = 0; rep < 400; rep++) {
            if (rep == 200) OK(hipEventRecord(e0, 0));
            if (var == 0) hipLaunchKernelGGL(k_update<0>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
            if (var == 1) hipLaunchKernelGGL(k_update<1>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
            if (var == 2) hipLaunchKernelGGL(k_update<2>, dim3(nb), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy);
            if (var == 3) hipLaunchKernelGGL(k_update4, dim3(nb4), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx4, gy);
            if (var == 4) hipLaunchKernelGGL(k_update_lds, dim3(nb8), dim3(512), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gy8);
            if (var == 5) hipLaunchKernelGGL(k_update_rows<2>, dim3(nbR2), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gyR2);
            if (var == 6) hipLaunchKernelGGL(k_update_rows<4>, dim3(nbR4), dim3(128), 0, 0, d[0], d[1], d[2], d[3], d[4], d[5], gx, gyR4);
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
               var == 0 ? "all taps global loads" : var == 1 ? "x-taps by ds_bpermute" : var == 2 ? "x-taps by DPP wave shifts" : var == 3 ? "four columns per lane, 16-byte loads" : var == 4 ? "z-taps staged in LDS, 8-row blocks" : var == 5 ? "two rows per wave, z-taps shared in registers" : "four rows per wave, z-taps shared in registers", 1e3 * ms / 200.0, cs);
    }
    return 0;
}
