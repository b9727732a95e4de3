// fwd_fused.hip -- one launch per forward time step: stress update, source, velocity update, boundary
// saving and receiver sampling fused (gfx950, wave64, LDS-staged).
//
// Replaces, per time step of the reference's forward loop (Src/libCUFD.cu:268-332):
//     from_bnd x5  ->  el_stress  ->  add_source  ->  el_velocity  ->  recording x4     (12 launches)
// with ONE kernel that moves every array once: 60 algorithmic bytes per cell instead of the 80 of the
// two-kernel form (the stresses written by the stress update are consumed from LDS, not re-read).
//
// Tiling ("one cell per thread per phase": no per-thread loops, so memory-level parallelism comes from
// the 16 waves of the block and the 2 blocks a CU holds, not from compiler unrolling).
//   A 1024-thread block = 16 waves = 16 rows x 64 columns of STRESS cells: the output tile of
//   FT_Z = 12 rows x FT_X = 60 columns plus the 2 rows / 2 columns around it that the velocity stencil
//   reaches.
//   phase A  every thread: sigma_new = sigma_old + f(v_old) for its cell -> LDS; cells of the output tile
//            also -> HBM (with their C-PML memory variables and the boundary-saving frame).  The halo
//            cells belong to neighbouring tiles, which compute the identical value from identical inputs.
//   barrier
//   phase B  threads of the output tile: v_new = v_old + g(sigma_new from LDS) -> HBM.
// Because a neighbour reads this tile's OLD v and sigma for its halo while this tile writes NEW values,
// fields are double-buffered (old -> new, swapped every step); so are the four stress-side C-PML memory
// variables (their halo values are recomputed too).  The velocity-side memory variables stay in place.
//
// Receivers.  Seismogram column `it` is a function of the state at the START of step `it`
// (libCUFD.cu:309-330 records after the velocity update of step it-1), i.e. of the OLD arrays this kernel
// reads: the tile's receivers are sampled here; the last column is sampled by k_record after the loop.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace sepfwi {

using namespace dev;

namespace {

constexpr int FT_Z = 12;  // output rows per tile
constexpr int FT_X = 60;  // output columns per tile
constexpr int LR = FT_Z + 4, LC = FT_X + 4;  // stress cells per block: 16 x 64 = one cell per thread

// Lean addressing: every array is "uniform base pointer (SGPR pair) + 32-bit byte offset (one VGPR shared
// by all arrays) + small immediate", which is gfx950's global_load_dword v, v_off, s[base] offset:imm form.
// 64-bit per-array address arithmetic (2 VALU per load) was half of the instruction stream of the first
// version of this kernel (profiles/r01_pmc_fwdfused_v1_summary.txt).
__device__ __forceinline__ float ldg(const float *base, unsigned off, int imm = 0) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off + imm);
}
__device__ __forceinline__ void stg(float *base, unsigned off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) = v;
}

}  // namespace

template <bool SAVE>
__global__ __launch_bounds__(1024, 8) void k_fwd_fused(Grid g, FwdFusedArgs a) {
    static_assert(LR * LC == 1024 && LC == 64, "one stress cell per thread, one row per wave");
    __shared__ float s_zz[LR][LC];
    __shared__ float s_xx[LR][LC];
    __shared__ float s_xz[LR][LC];

    int t = blockIdx.x;
    if (g.xcd_remap) {
        const int per = (g.gx * g.gy + 7) >> 3;
        t = (t & 7) * per + (t >> 3);
    }
    if (t >= g.gx * g.gy) return;  // whole block leaves together: no barrier hazard
    const int tz = t / g.gx, tx = t - tz * g.gx;
    const int lane = threadIdx.x & 63;
    const int r = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));  // LDS row = wave index (uniform)
    const int z = tz * FT_Z - 2 + r;                                        // uniform
    const int x = tx * FT_X - 2 + lane;
    const unsigned n = a.n;
    const unsigned P4 = 4u * (unsigned)g.pitch;  // bytes per row

    // array bases (uniform)
    const float *o_vz = a.fo, *o_vx = a.fo + n, *o_szz = a.fo + 2 * n, *o_sxx = a.fo + 3 * n, *o_sxz = a.fo + 4 * n;

    const bool own_row = (r >= 2 && r < FT_Z + 2);                          // uniform
    const bool own = own_row && lane >= 2 && lane < FT_X + 2;
    const bool row_in = (z >= 0 && z < g.nzc);                              // uniform
    const bool inside = row_in && x >= 0 && x < g.nx;
    const bool row_comp = (z >= 2 && z <= g.nzc - 3);                       // uniform
    const bool comp = row_comp && x >= 2 && x <= g.nx - 3;                  // el_stress.cu:52 / el_velocity.cu:47
    const unsigned off = inside ? 4u * ((unsigned)z * (unsigned)g.pitch + (unsigned)x) : 0u;
    // which stress components the velocity stencil of the tile needs from this row (uniform)
    const bool need_n = (r >= 1);          // szz (rows 1..15); sxx only on own rows, computed with szz
    const bool need_s = (r <= FT_Z + 2);   // sxz (rows 0..14)

    float szz = 0.f, sxx = 0.f, sxz = 0.f;
    float vz_c = 0.f, vx_c = 0.f, ba = 0.f, bb = 0.f;
    if (inside) {
        if (need_n) {
            szz = ldg(o_szz, off);
            sxx = ldg(o_sxx, off);
        }
        if (need_s) sxz = ldg(o_sxz, off);
        if (SAVE || comp) {
            vz_c = ldg(o_vz, off);
            vx_c = ldg(o_vx, off);
        }
        if constexpr (SAVE) {
            if (own) {  // boundary saving of the state BEFORE the update (libCUFD.cu:271-273)
                const int s = frame_slot(g, z, x);
                if (s >= 0) {
                    const int L = g.frame_len;
                    a.frame_t[s] = szz;
                    a.frame_t[L + s] = sxz;
                    a.frame_t[2 * L + s] = sxx;
                    a.frame_t[3 * L + s] = vz_c;
                    a.frame_t[4 * L + s] = vx_c;
                }
            }
        }
        if (comp) {
            const bool pz = in_pml_z(g, z);                                  // uniform
            const bool px = (x < g.nPml || x > g.nx - g.nPml - 1);           // el_stress.cu:61,77
            const float *cz = a.cz, *cx = a.cx;
            const int nzc = g.nzc, nx = g.nx;
            const unsigned um1 = off - P4, up1 = off + P4;
            if (own) {
                ba = ldg(a.media + 3 * n, off);
                bb = ldg(a.media + 4 * n, off);
            }
            if (need_n) {
                float dvz_dz = dminus(ldg(o_vz, um1 - P4), ldg(o_vz, um1), vz_c, ldg(o_vz, up1), g.rdz);
                float dvx_dx = dminus(ldg(o_vx, off, -8), ldg(o_vx, off, -4), vx_c, ldg(o_vx, off, 4), g.rdx);
                if (pz) {
                    const float p = cz[nzc + z] * ldg(a.mo, off) + cz[z] * dvz_dz;          // b_z, a_z
                    dvz_dz = dvz_dz * cz[2 * nzc + z] + p;                                  // 1/K_z
                    if (own) stg(a.mn, off, p);
                }
                if (px) {
                    const float p = cx[nx + x] * ldg(a.mo + 3 * n, off) + cx[x] * dvx_dx;   // b_x, a_x
                    dvx_dx = dvx_dx * cx[2 * nx + x] + p;
                    if (own) stg(a.mn + 3 * n, off, p);
                }
                const float lam = ldg(a.media, off), mu = ldg(a.media + n, off);
                const float l2m = lam + 2.0f * mu;
                szz = szz + (l2m * dvz_dz + lam * dvx_dx) * g.dt;
                sxx = sxx + (lam * dvz_dz + l2m * dvx_dx) * g.dt;
                if (z == a.z_src && x == a.x_src) {  // add_source, utilities.cu:531-538
                    szz += a.src_amp;
                    sxx += a.src_amp;
                }
                if (own) {
                    stg(a.fn + 2 * n, off, szz);
                    stg(a.fn + 3 * n, off, sxx);
                }
            }
            if (need_s) {
                float dvx_dz = dplus(ldg(o_vx, um1), vx_c, ldg(o_vx, up1), ldg(o_vx, up1 + P4), g.rdz);
                float dvz_dx = dplus(ldg(o_vz, off, -4), vz_c, ldg(o_vz, off, 4), ldg(o_vz, off, 8), g.rdx);
                if (pz) {
                    const float q = cz[4 * nzc + z] * ldg(a.mo + 2 * n, off) + cz[3 * nzc + z] * dvx_dz;  // b_zh, a_zh
                    dvx_dz = dvx_dz * cz[5 * nzc + z] + q;
                    if (own) stg(a.mn + 2 * n, off, q);
                }
                if (px) {
                    const float q = cx[4 * nx + x] * ldg(a.mo + n, off) + cx[3 * nx + x] * dvz_dx;        // b_xh, a_xh
                    dvz_dx = dvz_dx * cx[5 * nx + x] + q;
                    if (own) stg(a.mn + n, off, q);
                }
                sxz = sxz + ldg(a.media + 2 * n, off) * (dvx_dz + dvz_dx) * g.dt;
                if (own) stg(a.fn + 4 * n, off, sxz);
            }
        }
    }
    s_zz[r][lane] = szz;
    s_xx[r][lane] = sxx;
    s_xz[r][lane] = sxz;
    __syncthreads();

    // ---- phase B: velocities of the output tile from the new stresses in LDS ----
    if (own && comp) {
        float dszz_dz = dplus(s_zz[r - 1][lane], s_zz[r][lane], s_zz[r + 1][lane], s_zz[r + 2][lane], g.rdz);
        float dsxz_dx = dminus(s_xz[r][lane - 2], s_xz[r][lane - 1], s_xz[r][lane], s_xz[r][lane + 1], g.rdx);
        float dsxz_dz = dminus(s_xz[r - 2][lane], s_xz[r - 1][lane], s_xz[r][lane], s_xz[r + 1][lane], g.rdz);
        float dsxx_dx = dplus(s_xx[r][lane - 1], s_xx[r][lane], s_xx[r][lane + 1], s_xx[r][lane + 2], g.rdx);
        const float *cz = a.cz, *cx = a.cx;
        const int nzc = g.nzc, nx = g.nx;
        if (in_pml_z(g, z)) {
            const float p = cz[4 * nzc + z] * ldg(a.mv, off) + cz[3 * nzc + z] * dszz_dz;          // b_zh, a_zh
            stg(a.mv, off, p);
            dszz_dz = dszz_dz * cz[5 * nzc + z] + p;
            const float q = cz[nzc + z] * ldg(a.mv + 2 * n, off) + cz[z] * dsxz_dz;                // b_z, a_z
            stg(a.mv + 2 * n, off, q);
            dsxz_dz = dsxz_dz * cz[2 * nzc + z] + q;
        }
        if (x < g.nPml || x > g.nx - g.nPml) {  // el_velocity.cu:56,71
            const float p = cx[nx + x] * ldg(a.mv + n, off) + cx[x] * dsxz_dx;                     // b_x, a_x
            stg(a.mv + n, off, p);
            dsxz_dx = dsxz_dx * cx[2 * nx + x] + p;
            const float q = cx[4 * nx + x] * ldg(a.mv + 3 * n, off) + cx[3 * nx + x] * dsxx_dx;    // b_xh, a_xh
            stg(a.mv + 3 * n, off, q);
            dsxx_dx = dsxx_dx * cx[5 * nx + x] + q;
        }
        stg(a.fn, off, vz_c + (dszz_dz + dsxz_dx) * ba * g.dt);
        stg(a.fn + n, off, vx_c + (dsxz_dz + dsxx_dx) * bb * g.dt);
    }

    // ---- receivers of this tile: sample the OLD state into seismogram column `it` ----
    if (a.comps) {
        const int beg = a.rt_off[t], cnt = a.rt_off[t + 1] - beg;
        for (int k = threadIdx.x; k < cnt; k += 1024) {
            const int ci = a.rt_cell[beg + k], rr = a.rt_rec[beg + k];
            if (a.comps & 1) a.d_pr[rr] = o_szz[ci] + o_sxx[ci];
            const float vx = o_vx[ci];
            if (a.comps & 2) a.d_vx[rr] = vx;
            if (a.comps & 4) a.d_vz[rr] = o_vz[ci];
            if (a.comps & 8) a.d_ett[rr] = vx - o_vx[ci - 1];
        }
    }
}

void fwd_fused_tile_shape(int *rows, int *cols) {
    *rows = FT_Z;
    *cols = FT_X;
}

void launch_fwd_fused(hipStream_t st, const Grid &g0, const FwdFusedArgs &a, int xcd_remap) {
    Grid g = g0;
    g.bz = FT_Z;
    g.gx = (g.nx + FT_X - 1) / FT_X;
    g.gy = (g.nzc + FT_Z - 1) / FT_Z;
    g.xcd_remap = xcd_remap;
    const int nb = g.gx * g.gy;
    const dim3 grid(xcd_remap ? ((nb + 7) / 8) * 8 : nb);
    if (a.frame_t)
        hipLaunchKernelGGL((k_fwd_fused<true>), grid, dim3(1024), 0, st, g, a);
    else
        hipLaunchKernelGGL((k_fwd_fused<false>), grid, dim3(1024), 0, st, g, a);
}

}  // namespace sepfwi
