// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 2-D elastic propagator.
//
// What each kernel replaces in the reference (paths relative to DAS_Waveform_Inversion/Ops/FWI/Src):
//   k_stress<FWD>    el_stress(isFor=true)   el_stress.cu:50-87    + from_bnd x5 (utilities.cu:362-392,
//                    Boundary.cu:57-80) + add_source (utilities.cu:524-552)
//   k_velocity<FWD>  el_velocity(isFor=true) el_velocity.cu:45-82
//   k_velocity<REV>  el_velocity(isFor=false) el_velocity.cu:87-113 + source_grad (utilities.cu:719-730)
//                    + to_bnd(vz,vx) (utilities.cu:395-425)
//   k_stress<REV>    add_source(-) + el_stress(isFor=false) el_stress.cu:92-125 + to_bnd(szz,sxz,sxx)
//   k_velocity_adj   el_velocity_adj.cu:57-102
//   k_stress_adj     el_stress_adj.cu:53-97
//   k_bwd_a          k_velocity<REV> + k_stress_adj of the previous step in one launch   } the default backward step:
//   k_bwd_b          source_grad + k_stress<REV> + k_velocity_adj + line injection       } two launches
//   k_bwd_velocity / k_bwd_stress   the other legal pairing (option bwd_fuse=1)
//   k_record         recording, recording_vx, recording_vz, recording_exx / _ezz (utilities.cu:593-602,620-629,645-703)
//   k_inject         res_injection_exx / _ezz (utilities.cu:605-615,632-641)
//   k_residual       gpuMinus + cuda_cal_objective (utilities.cu:154-205)
//   k_model_prep     host transpose x MEGA (libCUFD.cu:71-77) + velInit/aveMuInit/aveBycInit
//                    (utilities.cu:109-152, Model.cu:66-87)
//   k_finalize_gradients  the atomicAdd sprays of el_stress.cu:112-123 / el_velocity.cu:105-110 in
//                    gather form, and the D2H transpose of libCUFD.cu:718-724 (not needed here)
//
// Arrays are row-major, x fastest, pitch `g.pitch` (fwi_types.hpp).  A wave covers 64 consecutive
// x of one row, so every global access of a wave is one 256-B line-aligned segment (plus the +-1/+-2
// shifted re-reads that hit the same lines in the vector L1).  Every cell update reads all its operands before its
// first store (one memory round trip per wave, DESIGN.md 3.1 "loads first").
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_common.hpp"
#include "kernels.hpp"
#include "persist_plan.hpp"

namespace sepfwi {

using namespace dev;

namespace {

constexpr int BX = 64;              // threads along x  (one wave)
constexpr int MAXT = 1024;          // block = 64 x bz threads, bz in {1..16} (run-time option "bz")

struct Cell {
    int z, x;
    size_t i;  // z*pitch + x
};

// Tile (and, in batched launches, shot) of this block.  Blocks are dealt round-robin to the 8 XCDs (blockIdx % 8 shares an
// L2); with xcd_remap the logical order gives each XCD a contiguous run of logical indices, so z-halo rows are re-read
// from the SAME L2 instead of once per XCD.  Batched launches (g.nb shots in one grid) order the pairs either shot-major
// (all tiles of shot 0, then shot 1, ...) or, shot_fastest, tile-major: the nb shots of one tile are dispatched back to
// back on one XCD, so the media coefficients of the tile (the same for every shot) are fetched from the fabric once
// and hit that XCD's L2 for the other shots.
__device__ __forceinline__ Cell my_cell(const Grid &g, int *shot = nullptr, int block = -1 /* default: blockIdx.x */) {
    Cell c;
    int t = block < 0 ? (int)blockIdx.x : block;
    const int ntile = g.gx * g.gy;
    const int nb = shot ? g.nb : 1;
    if (g.xcd_remap) {
        const int per = (ntile * nb + 7) >> 3;
        t = (t & 7) * per + (t >> 3);
    }
    if (shot) {
        int sh;
        if (g.shot_fastest) {
            const int q = t / nb;
            sh = t - q * nb;
            t = q;
        } else {
            sh = t / ntile;
            t -= sh * ntile;
            if (sh >= nb) {  // surplus block of the remapped numbering
                sh = nb - 1;
                t = ntile;
            }
        }
        *shot = sh;
    }
    const int ty = t / g.gx, tx = t - ty * g.gx;
    c.x = tx * BX + (threadIdx.x & (BX - 1));
    // row is wave-uniform: keep it in an SGPR so the z-profile loads and PML tests are scalar
    c.z = __builtin_amdgcn_readfirstlane(ty * g.bz + (int)(threadIdx.x >> 6));
    if (ty >= g.gy) c.z = g.nz + 1;  // surplus block of the remapped numbering: out of range
    c.i = (size_t)c.z * (size_t)g.pitch + (size_t)c.x;
    return c;
}

// 4-point harmonic mean of mu at the staggered corner (z+1/2, x+1/2): aveMuInit, utilities.cu:124-137.  amu_fly: rebuilt
// from mu (three neighbour taps that hit the cache) instead of streaming a second array; single precision with the
// hardware reciprocal (<= 1 ulp each), i.e. within 4e-7 of the reference's double-precision value.  While the option is
// on, k_model_prep stores exactly THIS value in md.ave_mu as well, so kernels that read the array (the backward ones,
// by default) and kernels that rebuild it see the same bits and reverse-time reconstruction cancels as before.  A zero
// mu gives 1/0 = inf -> 4/inf = 0, the reference's fluid rule.  Valid on [2, n-3]^2 (every cell the kernels update).
__device__ __forceinline__ float ave_mu_at(const Grid &g, const Media &md, size_t i, float mu0) {
    if (g.amu_fly) {
        const float s = (__builtin_amdgcn_rcpf(mu0) + __builtin_amdgcn_rcpf(md.mu[i + g.pitch])) +
                        (__builtin_amdgcn_rcpf(md.mu[i + 1]) + __builtin_amdgcn_rcpf(md.mu[i + g.pitch + 1]));
        return 4.0f * __builtin_amdgcn_rcpf(s);
    }
    return md.ave_mu[i];
}


// Imaging accumulators behind an accessor, so that the same bodies serve the per-step launches (accumulators in HBM, AccG)
// and the persistent time loop (accumulators of the workgroup's own tile in LDS, AccT below).
enum { ACC_LAM = 0, ACC_MU = 1, ACC_XZ = 2, ACC_A = 3, ACC_B = 4 };
template <int K>
__device__ __forceinline__ float *acc_array(const ImgAcc &a) {
    return K == ACC_LAM ? a.lam : K == ACC_MU ? a.mu : K == ACC_XZ ? a.xz : K == ACC_A ? a.a : a.b;
}
struct AccG {
    ImgAcc p;
    template <int K> __device__ __forceinline__ float ld(size_t i) const { return acc_array<K>(p)[i]; }
    template <int K> __device__ __forceinline__ void st(size_t i, float v) const { acc_array<K>(p)[i] = v; }
};
typedef __attribute__((address_space(3))) float lds_float;
// MASK bit K set: accumulator K of this lane's cell lives in LDS at cell[rank of K among the set bits * stride]
template <int MASK>
struct AccT {
    ImgAcc p;
    lds_float *cell;  // this lane's slot of the current row segment
    int stride;       // floats between two LDS-resident accumulator arrays of the tile
    template <int K> __device__ __forceinline__ float ld(size_t i) const {
        if constexpr ((MASK >> K) & 1) return cell[__builtin_popcount(MASK & ((1 << K) - 1)) * stride];
        else return acc_array<K>(p)[i];
    }
    template <int K> __device__ __forceinline__ void st(size_t i, float v) const {
        if constexpr ((MASK >> K) & 1) cell[__builtin_popcount(MASK & ((1 << K) - 1)) * stride] = v;
        else acc_array<K>(p)[i] = v;
    }
};

// How the backward bodies touch the wavefields, the adjoint fields and the C-PML memories.  MemPlain: ordinary loads / stores
// (every per-step launch; inside the persistent loop every row segment whose stencils stay within one XCD's band of rows).
// MemAgent: agent-scope accesses (`sc1`: loads bypass the vector L1 and are served coherently, stores are written through) for
// the persistent loop's segments next to another XCD's band -- the L2s of different XCDs are not coherent with each other.
struct MemPlain {
    static __device__ __forceinline__ float ld(const float *p) { return *p; }
    static __device__ __forceinline__ void st(float *p, float v) { *p = v; }
};
struct MemAgent {
    static __device__ __forceinline__ float ld(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    static __device__ __forceinline__ void st(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
};

}  // namespace

// ---------------------------------------------------------------------------------------------
// stress update
// ---------------------------------------------------------------------------------------------
// Returns whether this lane stored a non-zero value.  quiet (wave-uniform, option quiet_skip): every value the update would read
// is +0 -- nothing to do but the boundary save and the receiver samples; no_img: the adjoint stresses of the segment are all +0, the
// imaging condition would add +-0.
template <bool FWD, bool SAVE, class ACC, class MEM = MemPlain>
__device__ __forceinline__ bool stress_body(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md,
                                            const PmlCoef &pc, float *__restrict__ frame_t,  // this step's 5*frame_len block
                                            int z_src, int x_src, float src_amp,              // scale*stf[it]*dt
                                            const Fields &adj, const ACC &acc, const LineRec &lr, bool quiet = false,
                                            bool no_img = false) {
    const int z = c.z, x = c.x, P = g.pitch;
    if (z >= g.nzc || x >= g.nx) return false;
    const size_t i = c.i;

    if constexpr (FWD) {
        if constexpr (SAVE) {
            // boundary saving BEFORE this step's update (libCUFD.cu:271-273)
            const int s = frame_slot(g, z, x);
            if (s >= 0) {
                const int L = g.frame_len;
                frame_t[s] = f.szz[i];
                frame_t[L + s] = f.sxz[i];
                frame_t[2 * L + s] = f.sxx[i];
                frame_t[3 * L + s] = f.vz[i];
                frame_t[4 * L + s] = f.vx[i];
            }
        }
        if (z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3) return false;  // el_stress.cu:52
        if (quiet) {
            if (lr.n && z == lr.z) {
                const int r = x - lr.x0;
                if (r >= 0 && r < lr.n) {
                    if (lr.d_vx) lr.d_vx[r] = 0.0f;
                    if (lr.d_vz) lr.d_vz[r] = 0.0f;
                    if (lr.d_ett) lr.d_ett[r] = 0.0f;
                }
            }
            return false;
        }

        // every unconditional load of the cell is issued here, before the first store: a store makes the compiler
        // keep all later (may-alias) loads behind it, i.e. one more dependent memory round trip per wave
        const float vz0 = f.vz[i], vx0 = f.vx[i], vxm1 = f.vx[i - 1];
        float dvz_dz = dminus(f.vz[i - 2 * P], f.vz[i - P], vz0, f.vz[i + P], g.rdz);
        float dvx_dx = dminus(f.vx[i - 2], vxm1, vx0, f.vx[i + 1], g.rdx);
        float dvx_dz = dplus(f.vx[i - P], vx0, f.vx[i + P], f.vx[i + 2 * P], g.rdz);
        float dvz_dx = dplus(f.vz[i - 1], vz0, f.vz[i + 1], f.vz[i + 2], g.rdx);
        const float lam = md.lam[i], mu = md.mu[i], amu = ave_mu_at(g, md, i, mu);
        const float szz0 = f.szz[i], sxx0 = f.sxx[i], sxz0 = f.sxz[i];
        if (lr.n && z == lr.z) {
            // line receivers: seismogram column `it` = velocities at the START of step `it`, which this kernel
            // only reads (recording_vx / _vz / _exx, utilities.cu:593-602,645-677)
            const int r = x - lr.x0;
            if (r >= 0 && r < lr.n) {
                if (lr.d_vx) lr.d_vx[r] = vx0;
                if (lr.d_vz) lr.d_vz[r] = vz0;
                if (lr.d_ett) lr.d_ett[r] = vx0 - vxm1;
            }
        }

        bool nz = false;
        if (in_pml_z(g, z)) {  // wave-uniform branch
            float p = pc.b_z[z] * m.dvz_dz[i] + pc.a_z[z] * dvz_dz;
            m.dvz_dz[i] = p;
            dvz_dz = dvz_dz * pc.rK_z[z] + p;
            float q = pc.b_zh[z] * m.dvx_dz[i] + pc.a_zh[z] * dvx_dz;
            m.dvx_dz[i] = q;
            dvx_dz = dvx_dz * pc.rK_zh[z] + q;
            nz = (p != 0.0f) | (q != 0.0f);
        }
        if (x < g.nPml || x > g.nx - g.nPml - 1) {  // el_stress.cu:61,77
            float p = pc.b_x[x] * m.dvx_dx[i] + pc.a_x[x] * dvx_dx;
            m.dvx_dx[i] = p;
            dvx_dx = dvx_dx * pc.rK_x[x] + p;
            float q = pc.b_xh[x] * m.dvz_dx[i] + pc.a_xh[x] * dvz_dx;
            m.dvz_dx[i] = q;
            dvz_dx = dvz_dx * pc.rK_xh[x] + q;
            nz |= (p != 0.0f) | (q != 0.0f);
        }
        const float l2m = lam + 2.0f * mu;
        float szz = szz0 + (l2m * dvz_dz + lam * dvx_dx) * g.dt;
        float sxx = sxx0 + (lam * dvz_dz + l2m * dvx_dx) * g.dt;
        if (z == z_src && x == x_src) {  // add_source, utilities.cu:531-538
            szz += src_amp;
            sxx += src_amp;
        }
        const float sxz = sxz0 + amu * (dvx_dz + dvz_dx) * g.dt;
        f.szz[i] = szz;
        f.sxx[i] = sxx;
        f.sxz[i] = sxz;
        return nz | (szz != 0.0f) | (sxx != 0.0f) | (sxz != 0.0f);
    } else {
        // ---- reverse-time reconstruction + lambda/mu imaging ----
        if (quiet) return false;  // (a segment that never held a value: its saved frames are zeros as well)
        const bool interior = (z >= g.nPml && z <= g.zmax && x >= g.nPml && x <= g.xmax);
        const int s = frame_slot(g, z, x);
        if (!interior && s < 0) return false;
        float szz = 0.f, sxx = 0.f, sxz = 0.f;
        if (interior) {
            szz = MEM::ld(&f.szz[i]);
            sxx = MEM::ld(&f.sxx[i]);
            sxz = MEM::ld(&f.sxz[i]);
            if (z == z_src && x == x_src) {  // add_source(isFor=false) comes first (libCUFD.cu:566-569)
                szz -= src_amp;
                sxx -= src_amp;
            }
            const float dvz_dz = dminus(MEM::ld(&f.vz[i - 2 * P]), MEM::ld(&f.vz[i - P]), MEM::ld(&f.vz[i]), MEM::ld(&f.vz[i + P]), g.rdz);
            const float dvx_dx = dminus(MEM::ld(&f.vx[i - 2]), MEM::ld(&f.vx[i - 1]), MEM::ld(&f.vx[i]), MEM::ld(&f.vx[i + 1]), g.rdx);
            const float dvx_dz = dplus(MEM::ld(&f.vx[i - P]), MEM::ld(&f.vx[i]), MEM::ld(&f.vx[i + P]), MEM::ld(&f.vx[i + 2 * P]), g.rdz);
            const float dvz_dx = dplus(MEM::ld(&f.vz[i - 1]), MEM::ld(&f.vz[i]), MEM::ld(&f.vz[i + 1]), MEM::ld(&f.vz[i + 2]), g.rdx);
            const float lam = md.lam[i], mu = md.mu[i], amu = ave_mu_at(g, md, i, mu);
            const bool img = g.dt_img != 0.0f && !no_img;  // launch-uniform: option img_every images every k-th step only
            float za = 0.f, xa = 0.f, sa = 0.f, g_lam = 0.f, g_mu = 0.f, g_xz = 0.f;
            if (img) {
                za = MEM::ld(&adj.szz[i]); xa = MEM::ld(&adj.sxx[i]); sa = MEM::ld(&adj.sxz[i]);
                g_lam = acc.template ld<ACC_LAM>(i); g_mu = acc.template ld<ACC_MU>(i); g_xz = acc.template ld<ACC_XZ>(i);
            }
            const float l2m = lam + 2.0f * mu;
            szz -= (l2m * dvz_dz + lam * dvx_dx) * g.dt;
            sxx -= (lam * dvz_dz + l2m * dvx_dx) * g.dt;
            sxz -= amu * (dvx_dz + dvz_dx) * g.dt;
            if (img) {
                // imaging condition, el_stress.cu:108-115 (constant factors deferred to finalize)
                acc.template st<ACC_LAM>(i, g_lam + -(za + xa) * (dvz_dz + dvx_dx) * g.dt_img);
                acc.template st<ACC_MU>(i, g_mu + -2.0f * (za * dvz_dz + xa * dvx_dx) * g.dt_img);
                acc.template st<ACC_XZ>(i, g_xz + -sa * (dvx_dz + dvz_dx) * g.dt_img);
            }
        }
        if (s >= 0) {  // to_bnd(szz, sxz, sxx) overrides the frame (libCUFD.cu:582)
            const int L = g.frame_len;
            szz = frame_t[s];
            sxz = frame_t[L + s];
            sxx = frame_t[2 * L + s];
        }
        MEM::st(&f.szz[i], szz);
        MEM::st(&f.sxx[i], sxx);
        MEM::st(&f.sxz[i], sxz);
        return (szz != 0.0f) | (sxx != 0.0f) | (sxz != 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// velocity update
// ---------------------------------------------------------------------------------------------
// Buoyancy averages of cell i: byc_a = 2/(rho(z+1,x)+rho(z,x)), byc_b = 2/(rho(z,x+1)+rho(z,x))  (aveBycInit,
// utilities.cu:139-152).  rho_fly: rebuilt from the density -- one array streamed (+ two neighbour taps that hit the
// cache) instead of two; the IEEE float quotient equals the reference's (float)(2.0 / (double)sum) bit for bit
// (a double quotient of two floats rounds to float exactly like the float division).  Valid on [2, n-3]^2 of the
// padded grid, which contains every cell the velocity-type kernels update.
__device__ __forceinline__ void buoyancies(const Grid &g, const Media &md, size_t i, float &ba, float &bb) {
    if (g.rho_fly) {
        const float r0 = md.rho[i];
        ba = 2.0f / (md.rho[i + g.pitch] + r0);
        bb = 2.0f / (md.rho[i + 1] + r0);
    } else {
        ba = md.byc_a[i];
        bb = md.byc_b[i];
    }
}

template <bool FWD, class ACC, class MEM = MemPlain>
__device__ __forceinline__ bool velocity_body(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md,
                                              const PmlCoef &pc, const float *__restrict__ frame_t, int z_src, int x_src,
                                              float src_rxz, float *__restrict__ stf_grad_it, const Fields &adj,
                                              const ACC &acc, bool quiet = false, bool no_img = false) {
    const int z = c.z, x = c.x, P = g.pitch;
    if (z >= g.nzc || x >= g.nx) return false;
    const size_t i = c.i;

    if constexpr (FWD) {
        if (z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3) return false;  // el_velocity.cu:47
        if (quiet) return false;
        float dszz_dz = dplus(f.szz[i - P], f.szz[i], f.szz[i + P], f.szz[i + 2 * P], g.rdz);
        float dsxz_dx = dminus(f.sxz[i - 2], f.sxz[i - 1], f.sxz[i], f.sxz[i + 1], g.rdx);
        float dsxz_dz = dminus(f.sxz[i - 2 * P], f.sxz[i - P], f.sxz[i], f.sxz[i + P], g.rdz);
        float dsxx_dx = dplus(f.sxx[i - 1], f.sxx[i], f.sxx[i + 1], f.sxx[i + 2], g.rdx);
        const float vz0 = f.vz[i], vx0 = f.vx[i];  // all loads before the first store
        float ba, bb;
        buoyancies(g, md, i, ba, bb);
        bool nz = false;
        if (in_pml_z(g, z)) {
            float p = pc.b_zh[z] * m.dszz_dz[i] + pc.a_zh[z] * dszz_dz;
            m.dszz_dz[i] = p;
            dszz_dz = dszz_dz * pc.rK_zh[z] + p;
            float q = pc.b_z[z] * m.dsxz_dz[i] + pc.a_z[z] * dsxz_dz;
            m.dsxz_dz[i] = q;
            dsxz_dz = dsxz_dz * pc.rK_z[z] + q;
            nz = (p != 0.0f) | (q != 0.0f);
        }
        if (x < g.nPml || x > g.nx - g.nPml) {  // el_velocity.cu:56,71 (one column narrower on the right)
            float p = pc.b_x[x] * m.dsxz_dx[i] + pc.a_x[x] * dsxz_dx;
            m.dsxz_dx[i] = p;
            dsxz_dx = dsxz_dx * pc.rK_x[x] + p;
            float q = pc.b_xh[x] * m.dsxx_dx[i] + pc.a_xh[x] * dsxx_dx;
            m.dsxx_dx[i] = q;
            dsxx_dx = dsxx_dx * pc.rK_xh[x] + q;
            nz |= (p != 0.0f) | (q != 0.0f);
        }
        const float vz = vz0 + (dszz_dz + dsxz_dx) * ba * g.dt;
        const float vx = vx0 + (dsxz_dz + dsxx_dx) * bb * g.dt;
        f.vz[i] = vz;
        f.vx[i] = vx;
        return nz | (vz != 0.0f) | (vx != 0.0f);
    } else {
        // source_grad uses the adjoint stresses as they stand at the start of the step (libCUFD.cu:547)
        if (z == z_src && x == x_src) *stf_grad_it = -(MEM::ld(&adj.szz[i]) + src_rxz * MEM::ld(&adj.sxx[i])) * g.dt;
        if (quiet) return false;
        const bool interior = (z >= g.nPml && z <= g.zmax && x >= g.nPml && x <= g.xmax);
        const int s = frame_slot(g, z, x);
        if (!interior && s < 0) return false;
        float vz = 0.f, vx = 0.f;
        if (interior) {
            const float dszz_dz = dplus(MEM::ld(&f.szz[i - P]), MEM::ld(&f.szz[i]), MEM::ld(&f.szz[i + P]), MEM::ld(&f.szz[i + 2 * P]), g.rdz);
            const float dsxz_dx = dminus(MEM::ld(&f.sxz[i - 2]), MEM::ld(&f.sxz[i - 1]), MEM::ld(&f.sxz[i]), MEM::ld(&f.sxz[i + 1]), g.rdx);
            const float dsxz_dz = dminus(MEM::ld(&f.sxz[i - 2 * P]), MEM::ld(&f.sxz[i - P]), MEM::ld(&f.sxz[i]), MEM::ld(&f.sxz[i + P]), g.rdz);
            const float dsxx_dx = dplus(MEM::ld(&f.sxx[i - 1]), MEM::ld(&f.sxx[i]), MEM::ld(&f.sxx[i + 1]), MEM::ld(&f.sxx[i + 2]), g.rdx);
            const bool img = g.dt_img != 0.0f && !no_img;  // launch-uniform
            float g_a = 0.f, g_b = 0.f, avz = 0.f, avx = 0.f;
            if (img) {
                g_a = acc.template ld<ACC_A>(i); g_b = acc.template ld<ACC_B>(i); avz = MEM::ld(&adj.vz[i]); avx = MEM::ld(&adj.vx[i]);
            }
            float ba, bb;
            buoyancies(g, md, i, ba, bb);
            vz = MEM::ld(&f.vz[i]) - (dszz_dz + dsxz_dx) * ba * g.dt;
            vx = MEM::ld(&f.vx[i]) - (dsxz_dz + dsxx_dx) * bb * g.dt;
            if (img) {
                // density imaging, el_velocity.cu:101-104 (the -byc^2/2 factor is applied in finalize)
                acc.template st<ACC_A>(i, g_a + -avz * (dszz_dz + dsxz_dx) * g.dt_img);
                acc.template st<ACC_B>(i, g_b + -avx * (dsxz_dz + dsxx_dx) * g.dt_img);
            }
        }
        if (s >= 0) {  // to_bnd(vz, vx) (libCUFD.cu:563)
            const int L = g.frame_len;
            vz = frame_t[3 * L + s];
            vx = frame_t[4 * L + s];
        }
        MEM::st(&f.vz[i], vz);
        MEM::st(&f.vx[i], vx);
        return (vz != 0.0f) | (vx != 0.0f);
    }
}

// ---------------------------------------------------------------------------------------------
// adjoint velocity update.  el_velocity_adj.cu:57-102.  `f` holds the ADJOINT fields.
// The a*dpsi terms are evaluated only where a != 0 (inside the PML strips; a is exactly 0 elsewhere,
// utilities.cu:272-275,347-353), which lets k_stress_adj keep psi only near the strips.
// Split into LOAD (every unconditional global load of the cell, issued back to back) and APPLY (arithmetic, the
// rare C-PML branches, stores): a wave waits once for all of them, and the fused backward kernels can issue the
// LOAD of their second update before the first update's stores (a store keeps later may-alias loads behind it).
// ---------------------------------------------------------------------------------------------
// 1/K of the four C-PML profiles at (z, x).  K is exactly 1 outside the layers (cpmlInit, utilities.cu:272-275,
// 344-353: the damping profile is zero there; tests/test_host_logic.py checks it on the profiles), and a product with
// 1.0f is exact, so the interior skips the four loads without changing a bit.
__device__ __forceinline__ void load_rK(const Grid &g, const PmlCoef &pc, int z, int x, float &rKx, float &rKxh, float &rKz,
                                        float &rKzh) {
    rKx = rKxh = rKz = rKzh = 1.0f;
    if (g.rk_lazy == 0 || x < g.nPml || x > g.nx - g.nPml - 1) {
        rKx = pc.rK_x[x];
        rKxh = pc.rK_xh[x];
    }
    if (g.rk_lazy == 0 || in_pml_z(g, z)) {
        rKz = pc.rK_z[z];
        rKzh = pc.rK_zh[z];
    }
}

struct VelAdjIn {
    bool on;
    float szz_xm1, szz_0, szz_xp1, szz_xp2, szz_zm1, szz_zp1, szz_zp2;
    float sxx_xm1, sxx_0, sxx_xp1, sxx_xp2, sxx_zm1, sxx_zp1, sxx_zp2;
    float sxz_zm2, sxz_zm1, sxz_0, sxz_zp1, sxz_xm2, sxz_xm1, sxz_xp1;
    float vx, vz, lam, mu, amu, rKx, rKxh, rKz, rKzh;
};
template <class MEM = MemPlain>
__device__ __forceinline__ VelAdjIn velocity_adj_load(const Grid &g, const Cell &c, const Fields &f, const Media &md,
                                                      const PmlCoef &pc) {
    VelAdjIn q;
    const int z = c.z, x = c.x, P = g.pitch;
    q.on = !(z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3);
    if (!q.on) return q;
    const size_t i = c.i;
    q.szz_xm1 = MEM::ld(&f.szz[i - 1]); q.szz_0 = MEM::ld(&f.szz[i]); q.szz_xp1 = MEM::ld(&f.szz[i + 1]); q.szz_xp2 = MEM::ld(&f.szz[i + 2]);
    q.szz_zm1 = MEM::ld(&f.szz[i - P]); q.szz_zp1 = MEM::ld(&f.szz[i + P]); q.szz_zp2 = MEM::ld(&f.szz[i + 2 * P]);
    q.sxx_xm1 = MEM::ld(&f.sxx[i - 1]); q.sxx_0 = MEM::ld(&f.sxx[i]); q.sxx_xp1 = MEM::ld(&f.sxx[i + 1]); q.sxx_xp2 = MEM::ld(&f.sxx[i + 2]);
    q.sxx_zm1 = MEM::ld(&f.sxx[i - P]); q.sxx_zp1 = MEM::ld(&f.sxx[i + P]); q.sxx_zp2 = MEM::ld(&f.sxx[i + 2 * P]);
    q.sxz_zm2 = MEM::ld(&f.sxz[i - 2 * P]); q.sxz_zm1 = MEM::ld(&f.sxz[i - P]); q.sxz_0 = MEM::ld(&f.sxz[i]); q.sxz_zp1 = MEM::ld(&f.sxz[i + P]);
    q.sxz_xm2 = MEM::ld(&f.sxz[i - 2]); q.sxz_xm1 = MEM::ld(&f.sxz[i - 1]); q.sxz_xp1 = MEM::ld(&f.sxz[i + 1]);
    q.vx = MEM::ld(&f.vx[i]); q.vz = MEM::ld(&f.vz[i]);
    q.lam = md.lam[i]; q.mu = md.mu[i]; q.amu = ave_mu_at(g, md, i, q.mu);
    load_rK(g, pc, z, x, q.rKx, q.rKxh, q.rKz, q.rKzh);
    return q;
}
template <class MEM = MemPlain>
__device__ __forceinline__ bool velocity_adj_apply(const VelAdjIn &q, const Grid &g, const Cell &c, const Fields &f,
                                                   const PmlMem &m, const Media &md, const PmlCoef &pc, const LineRec &lr) {
    if (!q.on) return false;
    const int z = c.z, x = c.x, P = g.pitch;
    const size_t i = c.i;
    const bool pz = in_pml_z(g, z);
    const bool px = (x < g.nPml || x > g.nx - g.nPml - 1);
    const float lam = q.lam, amu = q.amu;
    const float l2m = lam + 2.0f * q.mu;

    // vx
    const float dszz_dx = -dplus(q.szz_xm1, q.szz_0, q.szz_xp1, q.szz_xp2, g.rdx);
    const float dsxx_dx = -dplus(q.sxx_xm1, q.sxx_0, q.sxx_xp1, q.sxx_xp2, g.rdx);
    const float dsxz_dz = -dminus(q.sxz_zm2, q.sxz_zm1, q.sxz_0, q.sxz_zp1, g.rdz);
    float upd = lam * dszz_dx * q.rKx * g.dt + l2m * dsxx_dx * q.rKx * g.dt + amu * q.rKzh * dsxz_dz * g.dt;
    // vz
    const float dszz_dz = -dplus(q.szz_zm1, q.szz_0, q.szz_zp1, q.szz_zp2, g.rdz);
    const float dsxx_dz = -dplus(q.sxx_zm1, q.sxx_0, q.sxx_zp1, q.sxx_zp2, g.rdz);
    const float dsxz_dx = -dminus(q.sxz_xm2, q.sxz_xm1, q.sxz_0, q.sxz_xp1, g.rdx);
    float upz = l2m * dszz_dz * q.rKz * g.dt + lam * dsxx_dz * q.rKz * g.dt + amu * q.rKxh * dsxz_dx * g.dt;
    if (px) {
        upd += pc.a_x[x] * -dplus(MEM::ld(&m.dvx_dx[i - 1]), MEM::ld(&m.dvx_dx[i]), MEM::ld(&m.dvx_dx[i + 1]), MEM::ld(&m.dvx_dx[i + 2]), g.rdx);
        upz += pc.a_xh[x] * -dminus(MEM::ld(&m.dvz_dx[i - 2]), MEM::ld(&m.dvz_dx[i - 1]), MEM::ld(&m.dvz_dx[i]), MEM::ld(&m.dvz_dx[i + 1]), g.rdx);
    }
    if (pz) {
        upd += pc.a_zh[z] * -dminus(MEM::ld(&m.dvx_dz[i - 2 * P]), MEM::ld(&m.dvx_dz[i - P]), MEM::ld(&m.dvx_dz[i]), MEM::ld(&m.dvx_dz[i + P]), g.rdz);
        upz += pc.a_z[z] * -dplus(MEM::ld(&m.dvz_dz[i - P]), MEM::ld(&m.dvz_dz[i]), MEM::ld(&m.dvz_dz[i + P]), MEM::ld(&m.dvz_dz[i + 2 * P]), g.rdz);
    }
    const float vx = q.vx + upd;
    const float vz = q.vz + upz;
    bool nz = vz != 0.0f;
    {
        // res_injection_exx (utilities.cu:605-615) for line receivers, applied by the thread that owns the cell:
        // vx_adj(z,x) += r[x]; vx_adj(z,x) -= r[x+1]   (after this kernel's update, libCUFD.cu:585-610)
        float vs = vx;
        if (lr.n && z == lr.z) {
            const int r = x - lr.x0;
            if (r >= 0 && r < lr.n) vs += lr.res[r];
            if (r + 1 >= 0 && r + 1 < lr.n) vs -= lr.res[r + 1];
        }
        MEM::st(&f.vx[i], vs);
        nz |= (vs != 0.0f) | (vx != 0.0f);
    }
    MEM::st(&f.vz[i], vz);
    if (px || pz) {  // the buoyancies are only needed inside the layers: keep their loads out of the interior
        const float bb = md.byc_b[i], ba = md.byc_a[i];
        if (px) {
            const float p = pc.b_xh[x] * MEM::ld(&m.dsxx_dx[i]) + bb * vx * g.dt, q2 = pc.b_x[x] * MEM::ld(&m.dsxz_dx[i]) + ba * vz * g.dt;
            MEM::st(&m.dsxx_dx[i], p);
            MEM::st(&m.dsxz_dx[i], q2);
            nz |= (p != 0.0f) | (q2 != 0.0f);
        }
        if (pz) {
            const float p = pc.b_z[z] * MEM::ld(&m.dsxz_dz[i]) + bb * vx * g.dt, q2 = pc.b_zh[z] * MEM::ld(&m.dszz_dz[i]) + ba * vz * g.dt;
            MEM::st(&m.dsxz_dz[i], p);
            MEM::st(&m.dszz_dz[i], q2);
            nz |= (p != 0.0f) | (q2 != 0.0f);
        }
    }
    return nz;
}
template <class MEM = MemPlain>
__device__ __forceinline__ bool velocity_adj_body(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m,
                                                  const Media &md, const PmlCoef &pc, const LineRec &lr, bool quiet = false) {
    if (quiet) return false;  // (wave-uniform: every adjoint value within reach is +0 and no channel of the line lies in the segment)
    const VelAdjIn q = velocity_adj_load<MEM>(g, c, f, md, pc);
    return velocity_adj_apply<MEM>(q, g, c, f, m, md, pc, lr);
}

// ---------------------------------------------------------------------------------------------
// adjoint stress update.  el_stress_adj.cu:53-97.  The reference updates the four psi arrays over
// the whole domain (strip tests commented out, :67-72,:88-95); they are only ever READ through
// stencils multiplied by a (zero outside the strips), so updating them on the strips widened by
// the stencil radius (2) gives identical results.  LOAD / APPLY split as above.
// ---------------------------------------------------------------------------------------------
struct StressAdjIn {
    bool on;
    float vz_xm1, vz_0, vz_xp1, vz_xp2, vz_zm2, vz_zm1, vz_zp1;
    float vx_zm1, vx_0, vx_zp1, vx_zp2, vx_xm2, vx_xm1, vx_xp1;
    float sxz, sxx, szz, ba, bb, rKx, rKxh, rKz, rKzh;
};
template <class MEM = MemPlain>
__device__ __forceinline__ StressAdjIn stress_adj_load(const Grid &g, const Cell &c, const Fields &f, const Media &md,
                                                       const PmlCoef &pc) {
    StressAdjIn q;
    const int z = c.z, x = c.x, P = g.pitch;
    q.on = !(z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3);
    if (!q.on) return q;
    const size_t i = c.i;
    q.vz_xm1 = MEM::ld(&f.vz[i - 1]); q.vz_0 = MEM::ld(&f.vz[i]); q.vz_xp1 = MEM::ld(&f.vz[i + 1]); q.vz_xp2 = MEM::ld(&f.vz[i + 2]);
    q.vz_zm2 = MEM::ld(&f.vz[i - 2 * P]); q.vz_zm1 = MEM::ld(&f.vz[i - P]); q.vz_zp1 = MEM::ld(&f.vz[i + P]);
    q.vx_zm1 = MEM::ld(&f.vx[i - P]); q.vx_0 = MEM::ld(&f.vx[i]); q.vx_zp1 = MEM::ld(&f.vx[i + P]); q.vx_zp2 = MEM::ld(&f.vx[i + 2 * P]);
    q.vx_xm2 = MEM::ld(&f.vx[i - 2]); q.vx_xm1 = MEM::ld(&f.vx[i - 1]); q.vx_xp1 = MEM::ld(&f.vx[i + 1]);
    q.sxz = MEM::ld(&f.sxz[i]); q.sxx = MEM::ld(&f.sxx[i]); q.szz = MEM::ld(&f.szz[i]);
    buoyancies(g, md, i, q.ba, q.bb);
    load_rK(g, pc, z, x, q.rKx, q.rKxh, q.rKz, q.rKzh);
    return q;
}
template <class MEM = MemPlain>
__device__ __forceinline__ bool stress_adj_apply(const StressAdjIn &q, const Grid &g, const Cell &c, const Fields &f,
                                                 const PmlMem &m, const Media &md, const PmlCoef &pc) {
    if (!q.on) return false;
    const int z = c.z, x = c.x, P = g.pitch;
    const size_t i = c.i;
    const bool pz = in_pml_z(g, z);
    const bool px = (x < g.nPml || x > g.nx - g.nPml - 1);
    const bool wz = (z < g.nPml + 2 || z > g.nzc - g.nPml - 3);  // psi needed by stencils centred in the strip
    const bool wx = (x < g.nPml + 2 || x > g.nx - g.nPml - 3);
    const float ba = q.ba, bb = q.bb;

    // sxz
    const float dvz_dx = -dplus(q.vz_xm1, q.vz_0, q.vz_xp1, q.vz_xp2, g.rdx);
    const float dvx_dz = -dplus(q.vx_zm1, q.vx_0, q.vx_zp1, q.vx_zp2, g.rdz);
    float us = dvz_dx * q.rKx * ba * g.dt + dvx_dz * q.rKz * bb * g.dt;
    // sxx, szz
    const float dvx_dx = -dminus(q.vx_xm2, q.vx_xm1, q.vx_0, q.vx_xp1, g.rdx);
    const float dvz_dz = -dminus(q.vz_zm2, q.vz_zm1, q.vz_0, q.vz_zp1, g.rdz);
    float ux = bb * dvx_dx * q.rKxh * g.dt;
    float uz = ba * dvz_dz * q.rKzh * g.dt;
    if (px) {
        us += pc.a_x[x] * -dplus(MEM::ld(&m.dsxz_dx[i - 1]), MEM::ld(&m.dsxz_dx[i]), MEM::ld(&m.dsxz_dx[i + 1]), MEM::ld(&m.dsxz_dx[i + 2]), g.rdx);
        ux += pc.a_xh[x] * -dminus(MEM::ld(&m.dsxx_dx[i - 2]), MEM::ld(&m.dsxx_dx[i - 1]), MEM::ld(&m.dsxx_dx[i]), MEM::ld(&m.dsxx_dx[i + 1]), g.rdx);
    }
    if (pz) {
        us += pc.a_z[z] * -dplus(MEM::ld(&m.dsxz_dz[i - P]), MEM::ld(&m.dsxz_dz[i]), MEM::ld(&m.dsxz_dz[i + P]), MEM::ld(&m.dsxz_dz[i + 2 * P]), g.rdz);
        uz += pc.a_zh[z] * -dminus(MEM::ld(&m.dszz_dz[i - 2 * P]), MEM::ld(&m.dszz_dz[i - P]), MEM::ld(&m.dszz_dz[i]), MEM::ld(&m.dszz_dz[i + P]), g.rdz);
    }
    const float sxz = q.sxz + us;
    const float sxx = q.sxx + ux;
    const float szz = q.szz + uz;
    MEM::st(&f.sxz[i], sxz);
    MEM::st(&f.sxx[i], sxx);
    MEM::st(&f.szz[i], szz);
    bool nz = (sxz != 0.0f) | (sxx != 0.0f) | (szz != 0.0f);
    if (wx || wz) {  // lambda, mu, ave_mu only feed the memory variables, which only exist near the layers
        const float amu = md.ave_mu[i];
        const float lam = md.lam[i], mu = md.mu[i];
        const float l2m = lam + 2.0f * mu;
        if (wx) {
            const float p = pc.b_xh[x] * MEM::ld(&m.dvz_dx[i]) + sxz * amu * g.dt, q2 = pc.b_x[x] * MEM::ld(&m.dvx_dx[i]) + lam * szz * g.dt + l2m * sxx * g.dt;
            MEM::st(&m.dvz_dx[i], p);
            MEM::st(&m.dvx_dx[i], q2);
            nz |= (p != 0.0f) | (q2 != 0.0f);
        }
        if (wz) {
            const float p = pc.b_zh[z] * MEM::ld(&m.dvx_dz[i]) + sxz * amu * g.dt, q2 = pc.b_z[z] * MEM::ld(&m.dvz_dz[i]) + l2m * szz * g.dt + lam * sxx * g.dt;
            MEM::st(&m.dvx_dz[i], p);
            MEM::st(&m.dvz_dz[i], q2);
            nz |= (p != 0.0f) | (q2 != 0.0f);
        }
    }
    return nz;
}
template <class MEM = MemPlain>
__device__ __forceinline__ bool stress_adj_body(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m,
                                                const Media &md, const PmlCoef &pc, bool quiet = false) {
    if (quiet) return false;
    const StressAdjIn q = stress_adj_load<MEM>(g, c, f, md, pc);
    return stress_adj_apply<MEM>(q, g, c, f, m, md, pc);
}

// ---------------------------------------------------------------------------------------------
// Quiet segments (option quiet_skip; Fields::q, fwi_types.hpp).  A wavefield is exactly +0 ahead of its numerical front -- on the
// headline model the forward field fills a fifth of the grid on average over a shot, the adjoint field grows downwards from the
// channels -- and an update whose every input is +0 stores +0 again: the bodies skip it, bit for bit the same arrays.  One bit per
// row segment and field group says "may hold a non-zero value"; it is set (never cleared) by the wave that stores one, and read by
// the waves of LATER launches only (each update reads the other group's map and sets its own group's bits; its own bit is the
// segment's own).  All arguments are wave-uniform.
// ---------------------------------------------------------------------------------------------
typedef const unsigned int __attribute__((address_space(4))) *qmap_t;  // read through the scalar cache: the map an update READS is not written
                                                                        // in the same launch (only the other group's is), its own bit only by itself
__device__ __forceinline__ void q_mark(unsigned int *q, const Grid &g, int z, int xs, bool nz, bool already) {
    if (!already && __ballot(nz) != 0ull && (threadIdx.x & (BX - 1)) == 0) {
        const int r = z + 2;
        atomicOr(&q[(size_t)(xs + 1) * (size_t)g.qzw + (size_t)(r >> 5)], 1u << (r & 31));
    }
}
// quiet-skipping kernels: a wave owns g.qr consecutive rows; its r-th (the row stays wave-uniform: scalar profile loads and PML tests)
__device__ __forceinline__ Cell row_of(const Grid &g, Cell c, int r) {
    c.z = __builtin_amdgcn_readfirstlane(c.z * g.qr + r);
    c.i = (size_t)c.z * (size_t)g.pitch + (size_t)c.x;
    return c;
}
__device__ __forceinline__ int seg_of(const Cell &c) { return __builtin_amdgcn_readfirstlane(c.x) >> 6; }

// The four updates with their maps: DECIDE (read the maps; wave-uniform, scalar loads only), then the body, then the own bit.
// A wave with nothing to do lives as long as its chain of dependent scalar loads: the decision therefore issues EVERY map word it
// may need before it looks at any (no short-circuit: `own || reach` made three dependent round trips of it, 3.5 us per quiet wave),
// and the fused backward kernels decide for both of their updates before they apply either.
struct QDec {
    bool on, own, quiet, no_img;
    int xs;
};
// own_map: the group the update writes; in_map: the group it reads through its stencils; img_map: the adjoint group its imaging
// condition reads at the cell itself (or null); force: something enters the segment from outside the fields (source, residual)
__device__ __forceinline__ QDec q_decide(const Grid &g, const Cell &c, const unsigned int *own_map, const unsigned int *in_map,
                                         const unsigned int *img_map, bool force) {
    QDec d{false, true, false, false, 0};
    const int z = c.z;
    d.on = own_map != nullptr && z >= 2 && z <= g.nzc - 3;
    if (d.on) {
        d.xs = seg_of(c);
        const int r = z + 2, w = r >> 5, sh = r & 31;
        const int col = (d.xs + 1) * g.qzw;
        const qmap_t own_p = (qmap_t)own_map, in_p = (qmap_t)in_map, img_p = (qmap_t)(img_map ? img_map : own_map);
        // ---- loads
        const unsigned int own_w = own_p[col + w];
        const unsigned int img_w = img_p[col + w];
        const unsigned int left_w = in_p[col - g.qzw + w], right_w = in_p[col + g.qzw + w];
        const unsigned int w0 = in_p[col + (z >> 5)], w1 = in_p[col + (z >> 5) + 1];  // bit of row z - 2 is z
        // ---- arithmetic
        const unsigned long long win = (unsigned long long)w0 | ((unsigned long long)w1 << 32);
        const unsigned int reach = ((unsigned int)(win >> (z & 31)) & 0x1fu) | (((left_w | right_w) >> sh) & 1u);
        d.own = ((own_w >> sh) & 1u) != 0;
        d.quiet = !(d.own | force | (reach != 0));
        d.no_img = img_map != nullptr && ((img_w >> sh) & 1u) == 0;
    }
    return d;
}
template <bool FWD>
__device__ __forceinline__ QDec q_dec_stress(const Grid &g, const Cell &c, const Fields &f, const Fields &adj, int z_src, int x_src, float src_amp) {
    const bool src = c.z == z_src && (x_src >> 6) == seg_of(c) && src_amp != 0.0f;
    return q_decide(g, c, f.q ? f.q + g.qn : nullptr, f.q, (!FWD && adj.q) ? adj.q + g.qn : nullptr, src);
}
template <bool FWD>
__device__ __forceinline__ QDec q_dec_velocity(const Grid &g, const Cell &c, const Fields &f, const Fields &adj) {
    return q_decide(g, c, f.q, f.q ? f.q + g.qn : nullptr, (!FWD && adj.q) ? adj.q : nullptr, false);
}
template <bool Q, bool FWD, bool SAVE, class ACC>
__device__ __forceinline__ void stress_update(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc,
                                              float *__restrict__ frame_t, int z_src, int x_src, float src_amp, const Fields &adj,
                                              const ACC &acc, const LineRec &lr) {
    if constexpr (!Q) {
        stress_body<FWD, SAVE>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, acc, lr);
        return;
    }
    const QDec d = q_dec_stress<FWD>(g, c, f, adj, z_src, x_src, src_amp);
    const bool nz = stress_body<FWD, SAVE>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, acc, lr, d.quiet, d.no_img);
    if (d.on) q_mark(f.q + g.qn, g, c.z, d.xs, nz, d.own);
}
template <bool Q, bool FWD, class ACC>
__device__ __forceinline__ void velocity_update(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc,
                                                const float *__restrict__ frame_t, int z_src, int x_src, float src_rxz,
                                                float *__restrict__ stf_grad_it, const Fields &adj, const ACC &acc) {
    if constexpr (!Q) {
        velocity_body<FWD>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_rxz, stf_grad_it, adj, acc);
        return;
    }
    const QDec d = q_dec_velocity<FWD>(g, c, f, adj);
    const bool nz = velocity_body<FWD>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_rxz, stf_grad_it, adj, acc, d.quiet, d.no_img);
    if (d.on) q_mark(f.q, g, c.z, d.xs, nz, d.own);
}
// The two halves of the fused backward step: ONE block of map loads decides for both updates.  Update A works on the forward
// fields (own_a / in_a), update B on the adjoint fields (own_b / in_b); A's imaging condition reads, at the cell itself, the adjoint
// group B reads through its stencils -- the middle bit of B's row window, no load of its own.
struct QDec2 {
    QDec a, b;
};
__device__ __forceinline__ QDec2 q_decide2(const Grid &g, const Cell &c, const unsigned int *own_a, const unsigned int *in_a, bool force_a,
                                           const unsigned int *own_b, const unsigned int *in_b, bool force_b) {
    QDec2 d{QDec{false, true, false, false, 0}, QDec{false, true, false, false, 0}};
    const int z = c.z;
    const bool on = own_a != nullptr && own_b != nullptr && z >= 2 && z <= g.nzc - 3;
    if (on) {
        const int xs = seg_of(c);
        const int r = z + 2, w = r >> 5, sh = r & 31, zw = z >> 5, zs = z & 31;
        const int col = (xs + 1) * g.qzw;
        const qmap_t oa = (qmap_t)own_a, ia = (qmap_t)in_a, ob = (qmap_t)own_b, ib = (qmap_t)in_b;
        // ---- loads
        const unsigned int own_wa = oa[col + w], own_wb = ob[col + w];
        const unsigned int la = ia[col - g.qzw + w], ra = ia[col + g.qzw + w], lb = ib[col - g.qzw + w], rb = ib[col + g.qzw + w];
        const unsigned int a0 = ia[col + zw], a1 = ia[col + zw + 1], b0 = ib[col + zw], b1 = ib[col + zw + 1];  // bit of row z - 2 is z
        // ---- arithmetic
        const unsigned int win_a = (unsigned int)((((unsigned long long)a0 | ((unsigned long long)a1 << 32)) >> zs) & 0x1full);
        const unsigned int win_b = (unsigned int)((((unsigned long long)b0 | ((unsigned long long)b1 << 32)) >> zs) & 0x1full);
        d.a.on = d.b.on = true;
        d.a.xs = d.b.xs = xs;
        d.a.own = ((own_wa >> sh) & 1u) != 0;
        d.b.own = ((own_wb >> sh) & 1u) != 0;
        d.a.quiet = !(d.a.own | force_a | ((win_a | (((la | ra) >> sh) & 1u)) != 0));
        d.b.quiet = !(d.b.own | force_b | ((win_b | (((lb | rb) >> sh) & 1u)) != 0));
        d.a.no_img = (win_b & 4u) == 0;  // row z of the group B reads
    }
    return d;
}
template <class ACC>
__device__ __forceinline__ void bwd_a_quiet(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc,
                                            const float *__restrict__ frame_t, const Fields &adj, const ACC &acc) {
    // A: reverse-time velocity (forward velocity group from the forward stress group; rho imaging reads the adjoint velocities);
    // B: adjoint stress (adjoint stress group from the adjoint velocity group)
    const QDec2 d = q_decide2(g, c, f.q, f.q ? f.q + g.qn : nullptr, false, adj.q ? adj.q + g.qn : nullptr, adj.q, false);
    const bool nz1 = velocity_body<false>(g, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, acc, d.a.quiet, d.a.no_img);
    if (d.a.on) q_mark(f.q, g, c.z, d.a.xs, nz1, d.a.own);
    const bool nz2 = stress_adj_body(g, c, adj, m, md, pc, d.b.quiet);
    if (d.b.on) q_mark(adj.q + g.qn, g, c.z, d.b.xs, nz2, d.b.own);
}
template <class ACC>
__device__ __forceinline__ void bwd_b_quiet(const Grid &g, const Cell &c, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc,
                                            float *__restrict__ frame_t, int z_src, int x_src, float src_amp, const Fields &adj, const ACC &acc,
                                            const LineRec &lr) {
    // A: reverse-time stress (forward stress group from the forward velocity group; lambda / mu imaging reads the adjoint stresses);
    // B: adjoint velocity (adjoint velocity group from the adjoint stress group) + the residual of the step
    const int xs = seg_of(c);
    const bool src = c.z == z_src && (x_src >> 6) == xs && src_amp != 0.0f;
    const bool rec = lr.n && c.z == lr.z && xs * BX + BX - 1 >= lr.x0 - 1 && xs * BX <= lr.x0 + lr.n - 1;  // cells lr.x0 - 1 ... lr.x0 + lr.n - 1
    const QDec2 d = q_decide2(g, c, f.q ? f.q + g.qn : nullptr, f.q, src, adj.q, adj.q ? adj.q + g.qn : nullptr, rec);
    const bool nz1 = stress_body<false, false>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, acc, LineRec{}, d.a.quiet, d.a.no_img);
    if (d.a.on) q_mark(f.q + g.qn, g, c.z, d.a.xs, nz1, d.a.own);
    const bool nz2 = velocity_adj_body(g, c, adj, m, md, pc, lr, d.b.quiet);
    if (d.b.on) q_mark(adj.q, g, c.z, d.b.xs, nz2, d.b.own);
}
// ---------------------------------------------------------------------------------------------
// kernels: one body each (the reference's launch structure) ...
// ---------------------------------------------------------------------------------------------
template <bool FWD, bool SAVE, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_stress(Grid g, Fields f, PmlMem m, Media md, PmlCoef pc, float *__restrict__ frame_t,
                                                 int z_src, int x_src, float src_amp, Fields adj, ImgAcc acc, LineRec lr) {
    if constexpr (Q) {
        const Cell c0 = my_cell(g);
        for (int r = 0; r < g.qr; r++)
            stress_update<Q, FWD, SAVE>(g, row_of(g, c0, r), f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, AccG{acc}, lr);
    } else {
        stress_update<Q, FWD, SAVE>(g, my_cell(g), f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, AccG{acc}, lr);
    }
}
template <bool FWD, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_velocity(Grid g, Fields f, PmlMem m, Media md, PmlCoef pc,
                                                   const float *__restrict__ frame_t, int z_src, int x_src, float src_rxz,
                                                   float *__restrict__ stf_grad_it, Fields adj, ImgAcc acc) {
    if constexpr (Q) {
        const Cell c0 = my_cell(g);
        for (int r = 0; r < g.qr; r++)
            velocity_update<Q, FWD>(g, row_of(g, c0, r), f, m, md, pc, frame_t, z_src, x_src, src_rxz, stf_grad_it, adj, AccG{acc});
    } else {
        velocity_update<Q, FWD>(g, my_cell(g), f, m, md, pc, frame_t, z_src, x_src, src_rxz, stf_grad_it, adj, AccG{acc});
    }
}
__global__ __launch_bounds__(MAXT) void k_velocity_adj(Grid g, Fields f, PmlMem m, Media md, PmlCoef pc) {
    velocity_adj_body(g, my_cell(g), f, m, md, pc, LineRec{});
}
__global__ __launch_bounds__(MAXT) void k_stress_adj(Grid g, Fields f, PmlMem m, Media md, PmlCoef pc) {
    stress_adj_body(g, my_cell(g), f, m, md, pc);
}

// ---------------------------------------------------------------------------------------------
// Backward step fused ACROSS its two independent chains (option "bwd_fuse" = 2, default; 0 = the reference's four
// kernels).  Reverse-time reconstruction and adjoint propagation only meet in the imaging condition, which reads the
// adjoint field of the START of the step.  The adjoint kernels need the OPPOSITE
// coefficient set of the reverse-time kernels of the same field type (adjoint stress uses the buoyancies,
// adjoint velocity uses lambda/mu/ave_mu: el_stress_adj.cu:63-96, el_velocity_adj.cu:69-93).  Pairing
//   k_bwd_a = reverse-time VELOCITY (+ rho imaging, frame restore)  +  adjoint STRESS of the PREVIOUS step
//   k_bwd_b = source_grad + reverse-time STRESS (+ lambda/mu imaging, frame restore) + adjoint VELOCITY + injection
// lets each kernel read one coefficient set only (8 B and 12 B per cell instead of 20 B + 20 B).  Legal
// because the adjoint stress of step t+1 is only consumed by (i) source_grad, (ii) the lambda/mu imaging and
// (iii) the adjoint velocity of step t -- all in k_bwd_b of step t, which runs after k_bwd_a of step t; the rho
// imaging in k_bwd_a reads the adjoint velocity, which the adjoint stress does not modify.  The adjoint stress
// of the very last step (t = 0) is never consumed and is not computed.  Order of operations on every array is
// the reference's (Src/libCUFD.cu:545-631).
// ---------------------------------------------------------------------------------------------
// Arrays arrive as bundles (base pointer + stride) to keep the kernel's SGPR count at or below 80, the limit for
// 8 waves per SIMD (MI355X_MICROARCH.md "Residency"): 37 separate pointers cost 74 SGPRs on their own.
struct BwdArgs {
    float *fields;       // vz, vx, szz, sxx, sxz          (stride n)
    float *mem;          // 8 C-PML memory variables       (stride n)
    float *adj;          // adjoint vz, vx, szz, sxx, sxz  (stride n)
    const float *media;  // lam, mu, ave_mu, byc_a, byc_b  (stride n)
    float *acc;          // lam, mu, xz, a, b              (stride n)
    const float *cz;     // z profiles a, b, 1/K, a_half, b_half, 1/K_half (stride nzc), then the six x profiles (stride nx)
    size_t n;
    unsigned int *qf, *qa;  // quiet-segment maps of the forward / adjoint fields (Fields::q), or null
};
__device__ __forceinline__ Fields fields_of(float *b, size_t n) { return Fields{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n}; }
__device__ __forceinline__ PmlMem mem_of(float *b, size_t n) {
    return PmlMem{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 5 * n, b + 6 * n, b + 7 * n};
}
__device__ __forceinline__ Media media_of(const float *b, size_t n) { return Media{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n, b + 5 * n}; }
__device__ __forceinline__ ImgAcc acc_of(float *b, size_t n) { return ImgAcc{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n}; }
__device__ __forceinline__ PmlCoef coef_of(const float *cz, const float *cx, int nzc, int nx) {
    return PmlCoef{cz, cz + nzc, cz + 2 * nzc, cz + 3 * nzc, cz + 4 * nzc, cz + 5 * nzc,
                   cx, cx + nx,  cx + 2 * nx,  cx + 3 * nx,  cx + 4 * nx,  cx + 5 * nx};
}

template <bool EARLY, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_bwd_a(Grid g, BwdArgs b, const float *__restrict__ frame_t) {
    Fields f = fields_of(b.fields, b.n), adj = fields_of(b.adj, b.n);
    f.q = b.qf;
    adj.q = b.qa;
    const PmlMem m = mem_of(b.mem, b.n);
    const Media md = media_of(b.media, b.n);
    const ImgAcc acc = acc_of(b.acc, b.n);
    const PmlCoef pc = coef_of(b.cz, b.cz + 6 * g.nzc, g.nzc, g.nx);
    const Cell c = my_cell(g);
    if constexpr (Q) {  // (one row per wave here: a loop over rows pushes these kernels into scalar-register spills)
        bwd_a_quiet(g, c, f, m, md, pc, frame_t, adj, AccG{acc});
        return;
    }
    if constexpr (EARLY) {  // adjoint-stress loads in flight together with the reverse-velocity loads
        const StressAdjIn q = stress_adj_load(g, c, adj, md, pc);
        velocity_body<false>(g, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, AccG{acc});
        stress_adj_apply(q, g, c, adj, m, md, pc);
    } else {
        velocity_body<false>(g, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, AccG{acc});
        stress_adj_body(g, c, adj, m, md, pc);
    }
}
template <bool EARLY, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_bwd_b(Grid g, BwdArgs b, float *__restrict__ frame_t, int zx_src /* z<<16 | x */,
                                                float src_amp, float src_rxz, float *__restrict__ stf_grad_it,
                                                int lr_zx /* z<<16 | x0 */, int lr_n, const float *__restrict__ lr_res) {
    const int z_src = zx_src >> 16, x_src = zx_src & 0xffff;
    const LineRec lr{lr_zx >> 16, lr_zx & 0xffff, lr_n, nullptr, nullptr, nullptr, lr_res};
    Fields f = fields_of(b.fields, b.n), adj = fields_of(b.adj, b.n);
    f.q = b.qf;
    adj.q = b.qa;
    const PmlMem m = mem_of(b.mem, b.n);
    const Media md = media_of(b.media, b.n);
    const ImgAcc acc = acc_of(b.acc, b.n);
    const PmlCoef pc = coef_of(b.cz, b.cz + 6 * g.nzc, g.nzc, g.nx);
    const Cell c = my_cell(g);
    // source_grad (utilities.cu:719-730): adjoint stresses after the adjoint stress update of the previous step
    if (c.z == z_src && c.x == x_src) *stf_grad_it = -(adj.szz[c.i] + src_rxz * adj.sxx[c.i]) * g.dt;
    if constexpr (Q) {
        bwd_b_quiet(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, AccG{acc}, lr);
        return;
    }
    if constexpr (EARLY) {  // adjoint-velocity loads in flight together with the reverse-stress loads
        const VelAdjIn q = velocity_adj_load(g, c, adj, md, pc);
        stress_body<false, false>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, AccG{acc}, LineRec{});
        velocity_adj_apply(q, g, c, adj, m, md, pc, lr);
    } else {
        stress_body<false, false>(g, c, f, m, md, pc, frame_t, z_src, x_src, src_amp, adj, AccG{acc}, LineRec{});
        velocity_adj_body(g, c, adj, m, md, pc, lr);
    }
}

// ---------------------------------------------------------------------------------------------
// Batched forms: the block index encodes (tile, shot of the batch) -- my_cell(); per-shot pointers in a ShotDev table in device memory.
// One launch advances EVERY shot of the batch by a half step.  Small grids stop being launch-bound (the reference issues
// 24 launches per shot and time step; the stream form 4; this one 4 / batch), and on the headline grid the three
// concurrent forward passes become one launch whose blocks pack without stream scheduling.  Same bodies as above.
// ---------------------------------------------------------------------------------------------
template <bool SAVE, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_stress_fwd_batch(Grid g, const ShotDev *__restrict__ shots, const float *__restrict__ media,
                                                           const float *__restrict__ cz, size_t n, size_t data_len, int it,
                                                           float src_scale) {
    int ish;
    const Cell c = my_cell(g, &ish);
    const ShotDev &s = shots[ish];
    Fields f = fields_of(s.fields, n);
    f.q = s.quiet;
    const PmlMem m = mem_of(s.mem, n);
    const Media md = media_of(media, n);
    const PmlCoef pc = coef_of(cz, cz + 6 * g.nzc, g.nzc, g.nx);
    float *frame_t = SAVE ? s.frame + (size_t)it * 5 * (size_t)g.frame_len : nullptr;
    // scale*stf[it]*dt rounded like the host's float product of the stream form (no contraction into the later add)
    const float amp = __fmul_rn(__fmul_rn(src_scale, s.stf[it]), g.dt);
    LineRec lr{};
    if ((s.comps & 16) && it >= 1) {  // bit 16: line sampled here; column `it` = velocities at the start of step `it`
        lr.z = s.lr_z;
        lr.x0 = s.lr_x0;
        lr.n = s.lr_n;
        const size_t c0 = (size_t)it * (size_t)s.nrec;
        lr.d_vx = (s.comps & 2) ? s.syn + data_len + c0 : nullptr;
        lr.d_vz = (s.comps & 4) ? s.syn + 2 * data_len + c0 : nullptr;
        lr.d_ett = (s.comps & 8) ? s.syn + 3 * data_len + c0 : nullptr;
    }
    stress_update<Q, true, SAVE>(g, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, Fields{}, AccG{}, lr);
}
template <bool Q = false>
__global__ __launch_bounds__(MAXT) void k_velocity_fwd_batch(Grid g, const ShotDev *__restrict__ shots, const float *__restrict__ media,
                                                             const float *__restrict__ cz, size_t n) {
    int ish;
    const Cell c = my_cell(g, &ish);
    const ShotDev &s = shots[ish];
    Fields f = fields_of(s.fields, n);
    f.q = s.quiet;
    const PmlMem m = mem_of(s.mem, n);
    const Media md = media_of(media, n);
    const PmlCoef pc = coef_of(cz, cz + 6 * g.nzc, g.nzc, g.nx);
    velocity_update<Q, true>(g, c, f, m, md, pc, nullptr, -1, -1, 0.0f, nullptr, Fields{}, AccG{});
}
template <bool EARLY, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_bwd_a_batch(Grid g, const ShotDev *__restrict__ shots, const float *__restrict__ media,
                                                      const float *__restrict__ cz, size_t n, int it) {
    int ish;
    const Cell c = my_cell(g, &ish);
    const ShotDev &s = shots[ish];
    Fields f = fields_of(s.fields, n), adj = fields_of(s.adj, n);
    f.q = s.quiet;
    adj.q = s.quiet ? s.quiet + 2 * (size_t)g.qn : nullptr;
    const PmlMem m = mem_of(s.bmem, n);
    const Media md = media_of(media, n);
    const ImgAcc acc = acc_of(s.acc, n);
    const PmlCoef pc = coef_of(cz, cz + 6 * g.nzc, g.nzc, g.nx);
    const float *frame_t = s.frame + (size_t)it * 5 * (size_t)g.frame_len;
    if constexpr (EARLY) {
        const StressAdjIn q = stress_adj_load(g, c, adj, md, pc);
        velocity_body<false>(g, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, AccG{acc});
        stress_adj_apply(q, g, c, adj, m, md, pc);
    } else {
        if constexpr (Q) {
            bwd_a_quiet(g, c, f, m, md, pc, frame_t, adj, AccG{acc});
        } else {
            velocity_body<false>(g, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, AccG{acc});
            stress_adj_body(g, c, adj, m, md, pc);
        }
    }
}
template <bool EARLY, bool Q = false>
__global__ __launch_bounds__(MAXT) void k_bwd_b_batch(Grid g, const ShotDev *__restrict__ shots, const float *__restrict__ media,
                                                      const float *__restrict__ cz, size_t n, int it, float src_scale) {
    int ish;
    const Cell c = my_cell(g, &ish);
    const ShotDev &s = shots[ish];
    Fields f = fields_of(s.fields, n), adj = fields_of(s.adj, n);
    f.q = s.quiet;
    adj.q = s.quiet ? s.quiet + 2 * (size_t)g.qn : nullptr;
    const PmlMem m = mem_of(s.bmem, n);
    const Media md = media_of(media, n);
    const ImgAcc acc = acc_of(s.acc, n);
    const PmlCoef pc = coef_of(cz, cz + 6 * g.nzc, g.nzc, g.nx);
    float *frame_t = s.frame + (size_t)it * 5 * (size_t)g.frame_len;
    const float amp = __fmul_rn(__fmul_rn(src_scale, s.stf[it]), g.dt);
    const LineRec lr{s.lr_z, s.lr_x0, s.lr_n, nullptr, nullptr, nullptr, s.res + (size_t)it * (size_t)s.nrec};
    if (c.z == s.z_src && c.x == s.x_src) s.stf_grad[it] = -(adj.szz[c.i] + s.src_rxz * adj.sxx[c.i]) * g.dt;  // source_grad
    if constexpr (EARLY) {
        const VelAdjIn q = velocity_adj_load(g, c, adj, md, pc);
        stress_body<false, false>(g, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, AccG{acc}, LineRec{});
        velocity_adj_apply(q, g, c, adj, m, md, pc, lr);
    } else {
        if constexpr (Q) {
            bwd_b_quiet(g, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, AccG{acc}, lr);
        } else {
            stress_body<false, false>(g, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, AccG{acc}, LineRec{});
            velocity_adj_body(g, c, adj, m, md, pc, lr);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Persistent backward time loop (option bwd_fuse = 4): ONE launch advances a shot through a whole backward
// pass.  The grid is occupancy-sized (every workgroup resident at once); a workgroup owns a fixed tile of 64-column row
// segments (host-built plan, persist_plan.hpp: every tile the same size +- 1, edge segments first) and walks it twice per
// time step: phase A = the k_bwd_a bodies, phase B = the k_bwd_b bodies -- the same bodies, the same order of operations on
// every array (Src/libCUFD.cu:545-631), hence bit-identical results.  What fixed ownership buys: the tile's imaging
// accumulators stay in LDS (template mask LMASK) instead of 8 B of HBM read-modify-write each per cell and step, and there is
// no grid fill / drain between the 2 x 3999 phases of a pass.
//
// Synchronisation between tiles (phases are numbered through the pass; flags[tile] = phases whose EDGE part is complete):
//   * a stencil reaches at most two rows / one segment column into a neighbouring tile, and everything a phase reads through
//     a stencil was written in the previous phase -- so a tile may start phase p once every neighbour has finished the edge
//     part of phase p-1 (their new values are there: RAW; they have read my old ones: WAR).  Edge segments come first in a
//     phase, the flag is published when the last wave has seen its edge stores acknowledged, and the interior part hides the
//     latency: the poll at the next phase start normally succeeds at once.  A workgroup barrier per phase orders the tile's own
//     waves.
//   * visibility: band = blockIdx % nband is the XCD (checked at run time: all workgroups of a band must report one XCC_ID),
//     so tiles that exchange halos share an L2 except across the nband - 1 band edges.  Inside a band plain stores are in the
//     shared L2 once acknowledged; the reader drops its CU's vector L1 once per phase (agent-scope acquire) after the poll.
//     Segments next to another band (kSegXband) run the bodies with MemAgent: sc1 loads and write-through stores.
//   * the pass starts with a rendezvous of the whole grid (below): if the grid is not resident at once, or a band is spread over
//     several XCDs, every workgroup leaves before anything is touched and the host runs the two-launch step instead.
//   * every spin is bounded; a time-out later in the pass raises *err, every workgroup leaves, the host reports it.
// Two things the wave timeline showed (profiles/r05_pk_trace.txt): tiles wait for their neighbours every phase, so the slowest tile
// sets the pace of all --
//   * the instruction arbiter serves the OLDEST wave first, and in a launch that never ends the CU's first workgroup stays the older
//     one: its tile ran a third faster than the second workgroup's (17 against 26 us per phase).  Wave priorities (s_setprio) are
//     therefore dealt so that the two workgroups interleave -- each has half of its waves on the upper pair of levels, and which
//     workgroup gets the odd levels alternates from phase to phase (option pk_prio);
//   * tiles that own the x C-PML strips execute the absorbing-layer branches (twice the loads): the host-built tiling cuts by cost,
//     not by count (persist_plan.hpp PlanCost, options pk_wx / pk_wz).
// Segment descriptors are read through the scalar cache: a vector load of one waits with vmcnt(0), i.e. for the previous row segment's
// stores as well (2.6 % of the backward step).
// ---------------------------------------------------------------------------------------------
#ifdef SEPFWI_PK_TRACE
// one-off wave timeline of the loop (build with SEPFWI_HIPCC_FLAGS=-DSEPFWI_PK_TRACE; scripts/gpu_r05_pk_trace.sh, scripts/pk_trace.py):
// per (tile, wave, phase 200..207 of the launch) s_memrealtime at the phase start (after the barrier), at each item start (up to 8), at
// the end of the wave's items and after the closing drain.  Kept in device memory, dumped to $SEPFWI_PK_TRACE at exit.
__device__ unsigned long long *g_pk_trace;
constexpr int kTrTiles = 512, kTrPh0 = 200, kTrPh = 8, kTrSlots = 12;
#endif
constexpr int kPersistSpinLimit = 1 << 21;   // polls of ~1 us: about two seconds (never reached once the start rendezvous has passed)
constexpr int kPersistStartLimit = 1 << 15;  // start rendezvous: ~30 ms

// registers sized for 8 waves per SIMD: two workgroups of 16 waves per CU
template <int LMASK>
__global__ __launch_bounds__(MAXT, 8) void k_bwd_persist(Grid g, const PersistArgs a) {
    extern __shared__ float lds_dyn[];
    __shared__ int next_item, edge_done, abort_flag, start_verdict, cu_slot_s;
    const ShotDev &s = a.s;
    const size_t n = a.n;
    const Fields f = fields_of(s.fields, n), adj = fields_of(s.adj, n);
    const PmlMem m = mem_of(s.bmem, n);
    const Media md = media_of(a.media, n);
    const PmlCoef pc = coef_of(a.cz, a.cz + 6 * g.nzc, g.nzc, g.nx);
    const int band = (int)(blockIdx.x % a.nband);
    const int tile = band * a.per_band + (int)(blockIdx.x / a.nband);
    const TileHdr &h = a.hdr[tile];
    typedef const uint32_t __attribute__((address_space(4))) *seg_table_t;  // constant for the launch: scalar loads
    const seg_table_t segs = (seg_table_t)(a.seg + (size_t)tile * (size_t)a.cap);
    const int nst = h.n_seg, n_edge = h.n_edge, nnb = h.n_nb;
    const int lane = threadIdx.x & (BX - 1);
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), nw = (int)(blockDim.x >> 6);
    lds_float *const lbase = (lds_float *)lds_dyn;
    AccT<LMASK> acc{acc_of(s.acc, n), nullptr, a.cap * BX};
    unsigned int *const my_flag = a.flags + (size_t)tile * 32;

    // ---- start rendezvous: nothing is touched before EVERY workgroup of the grid is known to be resident (they wait for each
    // other all pass long and cannot be pre-empted) and every band is known to sit on one XCD.  ONE word decides for all:
    // the last arriver votes GO, a workgroup that has waited too long (the GPU is busy with something else: another process'
    // kernels, another persistent grid) votes ABORT; whichever compare-and-swap comes first stands, also for late arrivers.
    if (threadIdx.x == 0) {
        next_item = 0;
        edge_done = 0;
        abort_flag = 0;
        unsigned int verdict = kPersistGo;
        if (a.nosync && blockIdx.x == 0) atomicExch(a.band_xcc + 9, kPersistGo);  // (the host reads the decision word)
        if (!a.nosync) {
            unsigned int *arrived = a.band_xcc + 8, *decision = a.band_xcc + 9;
            const unsigned int xcc = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;  // HW_REG_XCC_ID, 4 bits
            const unsigned int seen = atomicCAS(a.band_xcc + band, 0xffffffffu, xcc);  // first comer records, the others compare
            if (seen != 0xffffffffu && seen != xcc) atomicCAS(decision, 0u, kPersistAbortPlacement);
            if (atomicAdd(arrived, 1u) == gridDim.x - 1) atomicCAS(decision, 0u, kPersistGo);
            int spins = 0;
            while ((verdict = __hip_atomic_load(decision, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) == 0u) {
                __builtin_amdgcn_s_sleep(8);
                if (++spins > kPersistStartLimit) atomicCAS(decision, 0u, kPersistAbortResidency);
            }
        }
        start_verdict = (int)verdict;
        // first or second workgroup on this CU?  (arrival order at a per-CU counter kept in the spare words of the flag lines)
        const unsigned int hwid = __builtin_amdgcn_s_getreg(4 | (31 << 11));       // HW_REG_HW_ID: bits 8..15 = CU, shader array, engine
        const unsigned int xcc_ = __builtin_amdgcn_s_getreg(20 | (3 << 11)) & 15u;  // HW_REG_XCC_ID
        const unsigned int key = (xcc_ << 8) | ((hwid >> 8) & 0xffu);
        cu_slot_s = (int)atomicAdd(a.flags + (size_t)(key % gridDim.x) * 32 + 1 + (key / gridDim.x) % 31, 1u);
    }
    __syncthreads();  // (all waves keep their registers meanwhile: a workgroup reduced to one wave would make room for one that does not fit)
    if (start_verdict != (int)kPersistGo) return;
    const int cu_slot = cu_slot_s;
    auto set_prio = [&](int p) {  // (the instruction takes an immediate)
        switch (p & 3) {
            case 0: __builtin_amdgcn_s_setprio(0); break;
            case 1: __builtin_amdgcn_s_setprio(1); break;
            case 2: __builtin_amdgcn_s_setprio(2); break;
            default: __builtin_amdgcn_s_setprio(3); break;
        }
    };
    auto cell_of = [&](uint32_t d) {
        Cell c;
        c.z = __builtin_amdgcn_readfirstlane((int)(d & 0xffffu));
        c.x = (int)((d >> 16) & 0xffu) * BX + lane;
        c.i = (size_t)c.z * (size_t)g.pitch + (size_t)c.x;
        return c;
    };
    // prologue: the tile's accumulators HBM -> LDS (they carry the sum over the shots of the call)
    if constexpr (LMASK != 0) {
        for (int j = wave; j < nst; j += nw) {
            const Cell c = cell_of(segs[j]);
            lds_float *cell = lbase + j * BX + lane;
            int r = 0;
            if constexpr (LMASK & 1) cell[(r++) * acc.stride] = acc.p.lam[c.i];
            if constexpr (LMASK & 2) cell[(r++) * acc.stride] = acc.p.mu[c.i];
            if constexpr (LMASK & 4) cell[(r++) * acc.stride] = acc.p.xz[c.i];
            if constexpr (LMASK & 8) cell[(r++) * acc.stride] = acc.p.a[c.i];
        }
    }
    __syncthreads();

    auto grab = [&]() {  // next work item of the workgroup: (phase, segment) in execution order
        int v = 0;
        if (lane == 0) v = atomicAdd(&next_item, 1);
        return __builtin_amdgcn_readfirstlane(v);
    };
    int w = grab();
    int local = 0;  // phases done in this launch; a.phase0 + local numbers them through the pass
    bool dead = abort_flag != 0;
    for (int it = a.it_hi; it >= a.it_lo && !dead; it--) {
        Grid gs = g;
        if (a.img_every > 1) gs.dt_img = (it % a.img_every == 0) ? (float)a.img_every * g.dt : 0.0f;
        float *frame_t = s.frame + (size_t)it * 5 * (size_t)g.frame_len;
        const float amp = __fmul_rn(__fmul_rn(a.src_scale, s.stf[it]), g.dt);
        const LineRec lr{s.lr_z, s.lr_x0, s.lr_n, nullptr, nullptr, nullptr, s.res + (size_t)it * (size_t)s.nrec};
        for (int ph = 0; ph < 2 && !dead; ph++, local++) {
            const unsigned int phase = (unsigned int)(a.phase0 + local);
            // ---- neighbours through the edge part of the previous phase?  then drop what this CU's L1 still holds of their rows
            if (wave == 0 && !a.nosync) {
                bool ok = true;
                if (phase > 0 && lane < nnb) {
                    const unsigned int *pf = a.flags + (size_t)h.nb[lane] * 32;
                    int spins = 0;
                    while (__hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < phase) {
                        __builtin_amdgcn_s_sleep(4);
                        if (++spins > kPersistSpinLimit ||
                            ((spins & 255) == 0 && __hip_atomic_load(a.err, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0)) {
                            ok = false;
                            break;
                        }
                    }
                }
                if (!__all(ok)) {
                    if (lane == 0) {
                        atomicCAS(a.err, 0, 1);
                        abort_flag = 1;
                    }
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __syncthreads();  // the tile's own waves: everything of the previous phase is stored; poll and invalidate are done
            if (abort_flag) {
                dead = true;
                break;
            }
            const int base = local * nst;
            bool reported = false;
            // waves w, w+4, w+8, w+12 of a workgroup share a SIMD: two of them on levels {2, 3}, two on {0, 1}; the odd levels go to
            // the CU's first workgroup in even phases and to the second in odd ones
            if (a.prio == 1) set_prio(2 * ((wave >> 2) & 1) + ((cu_slot ^ local) & 1));
            else if (a.prio == 2) set_prio(2 * ((wave >> 2) & 1) + (cu_slot & 1));
            else if (a.prio == 3) set_prio(2 * ((wave >> 2) & 1) + ((cu_slot ^ (local % 3 == 0)) & 1));
#ifdef SEPFWI_PK_TRACE
            unsigned long long *tr = nullptr;
            int tr_k = 1;
            if (g_pk_trace && tile < kTrTiles && wave < 16 && local >= kTrPh0 && local < kTrPh0 + kTrPh && a.phase0 == 0)
                tr = g_pk_trace + (((size_t)tile * 16 + wave) * kTrPh + (local - kTrPh0)) * kTrSlots;
            if (tr && lane == 0) tr[0] = __builtin_amdgcn_s_memrealtime();
#endif
            // a wave that has seen its last edge segment of the phase waits for its stores, counts itself in; the last one publishes
            auto report = [&]() {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                int old = 0;
                if (lane == 0) old = atomicAdd(&edge_done, 1);
                old = __builtin_amdgcn_readfirstlane(old);
                if (old + 1 == nw * (local + 1) && lane == 0 && !a.nosync)
                    __hip_atomic_store(my_flag, phase + 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                reported = true;
            };
            for (; w < base + nst; w = grab()) {
                const int j = w - base;
#ifdef SEPFWI_PK_TRACE
                if (tr && lane == 0 && tr_k < 9) tr[tr_k++] = __builtin_amdgcn_s_memrealtime();
#endif
                if (j >= n_edge && !reported) report();
                const uint32_t d = segs[j];
                const Cell c = cell_of(d);
                acc.cell = lbase + j * BX + lane;
                const bool xband = (d & kSegXband) != 0 && !a.nosync;  // wave-uniform
                if (ph == 0) {
                    // phase A: reverse-time velocity (+ rho imaging, frame restore) + adjoint stress of the previous step
                    if (xband) {
                        velocity_body<false, AccT<LMASK>, MemAgent>(gs, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, acc);
                        stress_adj_body<MemAgent>(gs, c, adj, m, md, pc);
                    } else {
                        velocity_body<false>(gs, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, acc);
                        stress_adj_body(gs, c, adj, m, md, pc);
                    }
                } else {
                    // phase B: source_grad + reverse-time stress (+ lambda/mu imaging, frame restore) + adjoint velocity + injection
                    if (c.z == s.z_src && c.x == s.x_src) s.stf_grad[it] = -(adj.szz[c.i] + s.src_rxz * adj.sxx[c.i]) * g.dt;  // source_grad
                    if (xband) {
                        stress_body<false, false, AccT<LMASK>, MemAgent>(gs, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, acc, LineRec{});
                        velocity_adj_body<MemAgent>(gs, c, adj, m, md, pc, lr);
                    } else {
                        stress_body<false, false>(gs, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, acc, LineRec{});
                        velocity_adj_body(gs, c, adj, m, md, pc, lr);
                    }
                }
            }
#ifdef SEPFWI_PK_TRACE
            if (tr && lane == 0) tr[9] = __builtin_amdgcn_s_memrealtime();
#endif
            if (!reported) report();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // this wave's stores of the phase are complete before the next barrier
#ifdef SEPFWI_PK_TRACE
            if (tr && lane == 0) {
                tr[10] = __builtin_amdgcn_s_memrealtime();
                tr[11] = (unsigned long long)tr_k;
            }
#endif
        }
    }
    __syncthreads();

    // epilogue: LDS -> HBM
    if constexpr (LMASK != 0) {
        for (int j = wave; j < nst; j += nw) {
            const Cell c = cell_of(segs[j]);
            lds_float *cell = lbase + j * BX + lane;
            int r = 0;
            if constexpr (LMASK & 1) acc.p.lam[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 2) acc.p.mu[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 4) acc.p.xz[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 8) acc.p.a[c.i] = cell[(r++) * acc.stride];
        }
    }
}

// ---------------------------------------------------------------------------------------------
// receivers.  Seismograms are kept time-major on the device: d[comp][it][rec]  (coalesced for a
// horizontal fibre); they are transposed to the reference's [rec][it] files only on export.
// comps bit mask: 1 pressure, 2 vx, 4 vz, 8 ett.
// ---------------------------------------------------------------------------------------------
__global__ void k_record(Grid g, Fields f, int nrec, const int *__restrict__ rec_idx /* z*pitch+x */,
                         float *__restrict__ d_pr, float *__restrict__ d_vx, float *__restrict__ d_vz,
                         float *__restrict__ d_ett, int comps, int fiber, const float *__restrict__ sens) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const int i = rec_idx[r];
    if (comps & 1) d_pr[r] = f.szz[i] + f.sxx[i];
    const float vx = f.vx[i];
    if (comps & 2) d_vx[r] = vx;
    const float vz = f.vz[i];
    if (comps & 4) d_vz[r] = vz;
    if (!(comps & 8)) return;
    if (sens) {
        // directional channel: ett = s0 exx + s3 ezz + s1 exz (MOD/elasticSolver.py:266-276), strains as one-cell differences in
        // units of "strain x dx" like recording_exx (the z-differences carry dx/dz)
        const float k = g.dx * g.rdz;
        const float exx = vx - f.vx[i - 1];
        const float ezz = (vz - f.vz[i - g.pitch]) * k;
        const float exz = 0.5f * ((f.vx[i + g.pitch] - vx) * k + (f.vz[i + 1] - vz));
        d_ett[r] = sens[3 * r] * exx + sens[3 * r + 1] * ezz + sens[3 * r + 2] * exz;
        return;
    }
    // axial strain over one cell, not divided by the spacing (utilities.cu:600-601): exx for a horizontal fibre,
    // ezz (recording_ezz, utilities.cu:620-629) for a vertical one
    d_ett[r] = fiber ? vz - f.vz[i - g.pitch] : vx - f.vx[i - 1];
}

// res_injection_exx: vx_adj(z,x) += r ; vx_adj(z,x-1) -= r.  Adjacent channels share cells, so the
// two statements are applied through float atomics (the reference's plain +=/-= is racy there,
// utilities.cu:613-614).  Atomic order only permutes a few adds per cell.  With `sens`: the transpose of the
// directional channel above.
__global__ void k_inject(Fields adj, int nrec, const int *__restrict__ rec_idx, const float *__restrict__ res_t,
                         int down /* 0: horizontal fibre, else the pitch: vertical fibre (res_injection_ezz, utilities.cu:632-641) */,
                         const float *__restrict__ sens, int pitch, float dx_dz) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= nrec) return;
    const int i = rec_idx[r];
    const float v = res_t[r];
    if (sens) {
        const float a = sens[3 * r] * v, b = sens[3 * r + 1] * dx_dz * v, c = 0.5f * sens[3 * r + 2] * v;
        atomicAdd(&adj.vx[i], a);
        atomicAdd(&adj.vx[i - 1], -a);
        atomicAdd(&adj.vz[i], b);
        atomicAdd(&adj.vz[i - pitch], -b);
        atomicAdd(&adj.vx[i + pitch], c * dx_dz);
        atomicAdd(&adj.vx[i], -(c * dx_dz));
        atomicAdd(&adj.vz[i + 1], c);
        atomicAdd(&adj.vz[i], -c);
    } else if (down) {
        atomicAdd(&adj.vz[i], v);
        atomicAdd(&adj.vz[i - down], -v);
    } else {
        atomicAdd(&adj.vx[i], v);
        atomicAdd(&adj.vx[i - 1], -v);
    }
}

// residual r = obs - syn (time sample 0 forced to 0) and sum r^2, all time-major [it][rec].
// One double partial per block -> atomicAdd(double).
__global__ void k_residual(const float *__restrict__ obs, const float *__restrict__ syn, float *__restrict__ res,
                           int nrec, long long n, double *__restrict__ sumsq) {
    double s = 0.0;
    for (long long k = (long long)blockIdx.x * blockDim.x + threadIdx.x; k < n; k += (long long)gridDim.x * blockDim.x) {
        float r = (k < nrec) ? 0.0f : (obs[k] - syn[k]);  // first time sample: utilities.cu:159-163
        res[k] = r;
        s += (double)r * (double)r;
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[16];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane == 0) part[w] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int k = 0; k < (int)(blockDim.x >> 6); k++) t += part[k];
        atomicAdd(sumsq, t);
    }
}

// [rows][cols] -> [cols][rows] through a padded LDS tile (used for seismogram import/export).
__global__ void k_transpose(const float *__restrict__ in, float *__restrict__ out, int rows, int cols) {
    __shared__ float tile[32][33];
    int c = blockIdx.x * 32 + threadIdx.x, r0 = blockIdx.y * 32;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int r = r0 + k;
        if (r < rows && c < cols) tile[k][threadIdx.x] = in[(size_t)r * cols + c];
    }
    __syncthreads();
    int orow0 = blockIdx.x * 32, oc = blockIdx.y * 32 + threadIdx.x;
    for (int k = threadIdx.y; k < 32; k += blockDim.y) {
        int orow = orow0 + k;
        if (orow < cols && oc < rows) out[(size_t)orow * rows + oc] = tile[threadIdx.x][k];
    }
}

// ---------------------------------------------------------------------------------------------
// media preparation: boundary arrays (nz, nx) dense [MPa] -> internal pitched arrays [Pa] + averages
// ---------------------------------------------------------------------------------------------
__global__ void k_model_prep(Grid g, const float *__restrict__ Lam_in, const float *__restrict__ Mu_in,
                             const float *__restrict__ Den_in, float *__restrict__ lam, float *__restrict__ mu,
                             float *__restrict__ ave_mu, float *__restrict__ byc_a, float *__restrict__ byc_b,
                             float *__restrict__ rho, unsigned int *__restrict__ cp2_max_bits, int amu_fly) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y * blockDim.y + threadIdx.y;
    float cp2 = 0.0f;
    if (x < g.nx && z < g.nz) {
        const size_t si = (size_t)z * g.nx + x;
        const float L = (float)((double)Lam_in[si] * 1e6);  // libCUFD.cu:73-74
        const float M = (float)((double)Mu_in[si] * 1e6);
        const float D = Den_in[si];
        cp2 = (float)(((double)L + 2.0 * (double)M) / (double)D);  // velInit, utilities.cu:119-120 (squared)
        if (z < g.nzc) {
            const size_t i = (size_t)z * g.pitch + x;
            lam[i] = L;
            mu[i] = M;
            rho[i] = D;
            float am = 0.0f, ba = 1.0f / 1000.0f, bb = 1.0f / 1000.0f;  // Model.cu:67,72-73
            // averages exist on [2, n-3] of the FULL padded grid (utilities.cu:129,146); rows >= nzc are
            // never read by any kernel.
            if (z >= 2 && z <= g.nz - 3 && x >= 2 && x <= g.nx - 3) {
                const double a = M;
                const double b = (double)Mu_in[si + g.nx] * 1e6;
                const double c = (double)Mu_in[si + 1] * 1e6;
                const double d = (double)Mu_in[si + g.nx + 1] * 1e6;
                const float bf = (float)b, cf = (float)c, df = (float)d;
                if (amu_fly)  // the value the stress kernels rebuild on the fly (ave_mu_at): ONE definition of the average per session
                    am = 4.0f * __builtin_amdgcn_rcpf((__builtin_amdgcn_rcpf(M) + __builtin_amdgcn_rcpf(bf)) +
                                                      (__builtin_amdgcn_rcpf(cf) + __builtin_amdgcn_rcpf(df)));
                else if (!(M == 0.0f || bf == 0.0f || cf == 0.0f || df == 0.0f))
                    am = (float)(4.0 / (1.0 / a + 1.0 / (double)bf + 1.0 / (double)cf + 1.0 / (double)df));
                ba = (float)(2.0 / (double)(Den_in[si + g.nx] + D));
                bb = (float)(2.0 / (double)(Den_in[si + 1] + D));
            }
            ave_mu[i] = am;
            byc_a[i] = ba;
            byc_b[i] = bb;
        }
    }
    // max over the whole padded grid for the Courant guard (utilities.cu:225-232).  Cp^2 > 0, so the
    // float bit pattern orders like the value.
    for (int off = 32; off > 0; off >>= 1) cp2 = fmaxf(cp2, __shfl_down(cp2, off, 64));
    if ((threadIdx.y * blockDim.x + threadIdx.x) % 64 == 0 && cp2 > 0.0f) atomicMax(cp2_max_bits, __float_as_uint(cp2));
}

// ---------------------------------------------------------------------------------------------
// gradient finalisation (once per call): gather form of the reference's sprays, written straight
// into the boundary layout (nz, nx) dense -- rows >= nzc are zero.
//   el_stress.cu:108-123 :  gLam = MEGA*acc.lam ; gMu = MEGA*acc.mu + sum_p S(p)/mu(z,x)^2 over the
//       staggered points p in {(z,x),(z-1,x),(z,x-1),(z-1,x-1)} that sprayed onto (z,x)
//   el_velocity.cu:101-110: gDen = A(z,x)+B(z,x)+A(z-1,x)+B(z,x-1), A = acc.a*(-byc_a^2/2), ...
// including the reference's edge tests (the x+1 spray is unconditional).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float xz_weight(const Grid &g, const Media &md, const ImgAcc &acc, int pz, int px) {
    if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
    const size_t p = (size_t)pz * g.pitch + px;
    const float am = md.ave_mu[p];
    if (am == 0.0f) return 0.0f;
    const double h = 1.0 / (double)md.mu[p] + 1.0 / (double)md.mu[p + g.pitch] + 1.0 / (double)md.mu[p + 1] +
                     1.0 / (double)md.mu[p + g.pitch + 1];
    return (float)((double)(acc.xz[p] * am) / h * 1e6);
}

__global__ void k_finalize_gradients(Grid g, Media md, ImgAcc acc, float *__restrict__ gLam, float *__restrict__ gMu,
                                     float *__restrict__ gDen) {
    const int x = blockIdx.x * blockDim.x + threadIdx.x;
    const int z = blockIdx.y * blockDim.y + threadIdx.y;
    if (x >= g.nx || z >= g.nz) return;
    const size_t o = (size_t)z * g.nx + x;
    float gl = 0.0f, gm = 0.0f, gd = 0.0f;
    // targets can lie one column right of the interior (always-true x test of the reference)
    if (z >= g.nPml && z <= g.zmax && x >= g.nPml && x <= g.xmax + 1) {
        const size_t i = (size_t)z * g.pitch + x;
        const bool inside = (x <= g.xmax);
        if (inside) {
            gl = (float)((double)acc.lam[i] * 1e6);
            gm = (float)((double)acc.mu[i] * 1e6);
        }
        // A fluid cell (mu = 0) makes every corner around it a zero-average one, which sprays nothing (el_stress.cu:112: the spray
        // is inside `if (ave_Mu != 0)`); its 1/mu^2 = inf must not meet those zero weights (inf * 0 = NaN).
        const double rmu2 = 1.0 / ((double)md.mu[i] * (double)md.mu[i]);
        auto spray = [&](int pz, int px) {
            const float s = xz_weight(g, md, acc, pz, px);
            if (s != 0.0f) gm += (float)(rmu2 * (double)s);
        };
        spray(z, x);          // own corner
        spray(z - 1, x);      // sprayed down (z+1<=zmax holds: target z<=zmax)
        spray(z, x - 1);      // sprayed right, unconditional
        if (inside) spray(z - 1, x - 1);
        // density
        auto A = [&](int pz, int px) -> float {
            if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
            const size_t p = (size_t)pz * g.pitch + px;
            const double b = md.byc_a[p];
            return (float)((double)acc.a[p] * (-(b * b) / 2.0));
        };
        auto B = [&](int pz, int px) -> float {
            if (pz < g.nPml || pz > g.zmax || px < g.nPml || px > g.xmax) return 0.0f;
            const size_t p = (size_t)pz * g.pitch + px;
            const double b = md.byc_b[p];
            return (float)((double)acc.b[p] * (-(b * b) / 2.0));
        };
        gd = A(z, x) + B(z, x) + A(z - 1, x) + B(z, x - 1);
    }
    gLam[o] = gl;
    gMu[o] = gm;
    gDen[o] = gd;
}

// =============================================================================================
// options + launchers
// =============================================================================================
// Process-wide DEFAULTS of the kernel options (sepfwi_set_option).  Nothing on the launch path reads them: every
// Session::run takes one snapshot (kernel_options()) and hands it to the launchers, so sessions running on different
// host threads (one per GPU) never see a half-updated block.
static std::mutex g_opt_mu;
static KernelOptions g_opt;

KernelOptions kernel_options() {
    std::lock_guard<std::mutex> lock(g_opt_mu);
    return g_opt;
}

namespace {
struct OptField {
    const char *name;
    int KernelOptions::*field;
    int lo, hi;
};
const OptField kOptFields[] = {
    {"bz", &KernelOptions::bz, 1, 16},           {"xcd_remap", &KernelOptions::xcd_remap, 0, 1},
    {"bwd_fuse", &KernelOptions::bwd_fuse, 0, 4}, {"line_fuse", &KernelOptions::line_fuse, 0, 1},
    {"pair_fwd", &KernelOptions::pair_fwd, 0, 1}, {"fwd_lanes", &KernelOptions::fwd_lanes, 1, 4},
    {"early", &KernelOptions::early, 0, 3},       {"rho_fly", &KernelOptions::rho_fly, 0, 3},
    {"amu_fly", &KernelOptions::amu_fly, 0, 3},   {"rk_lazy", &KernelOptions::rk_lazy, 0, 1},
    {"batch", &KernelOptions::batch, 0, 2},       {"batch_f", &KernelOptions::batch_f, 0, 64},
    {"batch_b", &KernelOptions::batch_b, 0, 64},  {"batch_mb", &KernelOptions::batch_mb, 1, 1 << 20},
    {"batch_order", &KernelOptions::batch_order, 0, 1},
    {"probe", &KernelOptions::probe, 0, 1 << 30},
    {"img_every", &KernelOptions::img_every, 1, 64},
    {"obs_cache_mb", &KernelOptions::obs_cache_mb, 0, 1 << 30},
    {"quiet_skip", &KernelOptions::quiet_skip, 0, 1}, {"quiet_rows", &KernelOptions::quiet_rows, 1, 16},
    {"pk_lmask", &KernelOptions::pk_lmask, 0, 16},  {"pk_wpc", &KernelOptions::pk_wpc, 1, 4},
    {"pk_px", &KernelOptions::pk_px, 1, 64},         {"pk_waves", &KernelOptions::pk_waves, 4, 16}, {"pk_order", &KernelOptions::pk_order, 0, 1},         {"pk_nosync", &KernelOptions::pk_nosync, 0, 1},
    {"pk_prio", &KernelOptions::pk_prio, 0, 3}, {"pk_wx", &KernelOptions::pk_wx, 25, 400}, {"pk_wxp", &KernelOptions::pk_wxp, 25, 400}, {"pk_wz", &KernelOptions::pk_wz, 25, 400},
};
}  // namespace

int get_kernel_option(const char *name) {
    const std::string n(name ? name : "");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    for (const OptField &f : kOptFields)
        if (n == f.name) return g_opt.*(f.field);
    return -1;
}

int set_kernel_option(const char *name, int value) {
    const std::string n(name ? name : "");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    for (const OptField &f : kOptFields)
        if (n == f.name) {
            if (value < f.lo || value > f.hi || (n == "bwd_fuse" && (value == 1 || value == 3))) return -1;
            g_opt.*(f.field) = value;
            return 0;
        }
    return -1;
}

static inline Grid tiled(const Grid &g0, const KernelOptions &o, int fly_bit = -1, bool quiet = false) {
    Grid g = g0;
    g.bz = o.bz;
    g.qr = quiet ? o.quiet_rows : 1;
    g.gx = (g.nx + BX - 1) / BX;
    g.gy = (g.nzc + g.bz * g.qr - 1) / (g.bz * g.qr);
    g.xcd_remap = o.xcd_remap;
    g.rho_fly = fly_bit < 0 ? 0 : (o.rho_fly >> fly_bit) & 1;
    g.amu_fly = fly_bit < 0 ? 0 : (o.amu_fly >> fly_bit) & 1;
    g.rk_lazy = o.rk_lazy;
    const int nb = g.gx * g.gy;
    g.nblk = g.xcd_remap ? ((nb + 7) / 8) * 8 : nb;
    return g;
}
static inline dim3 field_grid(const Grid &g) { return dim3(g.nblk); }
#define BLOCK dim3(BX *g.bz)

void launch_stress_fwd(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                       float *frame_t, int z_src, int x_src, float src_amp, LineRec lr) {
    const Grid g = tiled(g0, o, 0, f.q != nullptr);
    Fields none{};
    ImgAcc na{};
    auto k = frame_t ? (f.q ? k_stress<true, true, true> : k_stress<true, true, false>) : (f.q ? k_stress<true, false, true> : k_stress<true, false, false>);
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, f, m, md, pc, frame_t, z_src, x_src, src_amp, none, na, lr);
}

void launch_velocity_fwd(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 0, f.q != nullptr);
    Fields none{};
    ImgAcc na{};
    auto k = f.q ? k_velocity<true, true> : k_velocity<true, false>;
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, f, m, md, pc, (const float *)nullptr, -1, -1, 0.0f, (float *)nullptr, none, na);
}

void launch_velocity_rev(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, Media md, PmlCoef pc,
                         const float *frame_t, int z_src, int x_src, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    PmlMem nm{};
    hipLaunchKernelGGL((k_velocity<false>), field_grid(g), BLOCK, 0, st, g, f, nm, md, pc, frame_t, z_src, x_src,
                       src_rxz, stf_grad_it, adj, acc);
}

void launch_stress_rev(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, Media md, PmlCoef pc, float *frame_t,
                       int z_src, int x_src, float src_amp, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    PmlMem nm{};
    hipLaunchKernelGGL((k_stress<false, false>), field_grid(g), BLOCK, 0, st, g, f, nm, md, pc, frame_t, z_src,
                       x_src, src_amp, adj, acc, LineRec{});
}

void launch_velocity_adj(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 1);
    hipLaunchKernelGGL(k_velocity_adj, field_grid(g), BLOCK, 0, st, g, adj, m, md, pc);
}

void launch_stress_adj(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 1);
    hipLaunchKernelGGL(k_stress_adj, field_grid(g), BLOCK, 0, st, g, adj, m, md, pc);
}

// The fused backward kernels (and every batched kernel) receive their arrays as bundles: base pointer + one stride, the C-PML
// profiles as ONE block (six z profiles of nzc floats, then six x profiles of nx).  The session allocates them that way; a
// caller that does not must hear about it here, not through a kernel reading the wrong array.
static void check_bundles(const Grid &g, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc, const Fields &adj, const ImgAcc &acc) {
    const ptrdiff_t n = f.vx - f.vz;
    const bool fields_ok = f.szz - f.vx == n && f.sxx - f.szz == n && f.sxz - f.sxx == n;
    const bool adj_ok = adj.vx - adj.vz == n && adj.szz - adj.vx == n && adj.sxx - adj.szz == n && adj.sxz - adj.sxx == n;
    const bool mem_ok = m.dvz_dx - m.dvz_dz == n && m.dvx_dz - m.dvz_dx == n && m.dvx_dx - m.dvx_dz == n && m.dszz_dz - m.dvx_dx == n &&
                        m.dsxz_dx - m.dszz_dz == n && m.dsxz_dz - m.dsxz_dx == n && m.dsxx_dx - m.dsxz_dz == n;
    const bool media_ok = md.mu - md.lam == n && md.ave_mu - md.mu == n && md.byc_a - md.ave_mu == n && md.byc_b - md.byc_a == n && md.rho - md.byc_b == n;
    const bool acc_ok = acc.mu - acc.lam == n && acc.xz - acc.mu == n && acc.a - acc.xz == n && acc.b - acc.a == n;
    const ptrdiff_t z = g.nzc, x = g.nx;
    const bool coef_ok = pc.b_z - pc.a_z == z && pc.rK_z - pc.b_z == z && pc.a_zh - pc.rK_z == z && pc.b_zh - pc.a_zh == z && pc.rK_zh - pc.b_zh == z &&
                         pc.a_x - pc.rK_zh == z && pc.b_x - pc.a_x == x && pc.rK_x - pc.b_x == x && pc.a_xh - pc.rK_x == x && pc.b_xh - pc.a_xh == x &&
                         pc.rK_xh - pc.b_xh == x;
    if (!(n > 0 && fields_ok && adj_ok && mem_ok && media_ok && acc_ok && coef_ok))
        throw std::logic_error("bundled kernel launch: arrays are not laid out as base + k * stride (fields " + std::to_string(fields_ok) + ", adjoint " +
                               std::to_string(adj_ok) + ", memories " + std::to_string(mem_ok) + ", media " + std::to_string(media_ok) + ", accumulators " +
                               std::to_string(acc_ok) + ", C-PML profiles " + std::to_string(coef_ok) + ")");
}

void launch_bwd_a(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                  const float *frame_t, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    check_bundles(g, f, m, md, pc, adj, acc);
    const BwdArgs b{f.vz, m.dvz_dz, adj.vz, md.lam, acc.lam, pc.a_z, (size_t)(f.vx - f.vz), f.q, adj.q};
    auto k = f.q ? k_bwd_a<false, true> : (o.early & 1) ? k_bwd_a<true, false> : k_bwd_a<false, false>;
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, b, frame_t);
}

void launch_bwd_b(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t,
                  int z_src, int x_src, float src_amp, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc, LineRec lr,
                  hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled(g0, o, 1);
    check_bundles(g, f, m, md, pc, adj, acc);
    const BwdArgs b{f.vz, m.dvz_dz, adj.vz, md.lam, acc.lam, pc.a_z, (size_t)(f.vx - f.vz), f.q, adj.q};
    auto k = f.q ? k_bwd_b<false, true> : (o.early & 2) ? k_bwd_b<true, false> : k_bwd_b<false, false>;
    if (ev_start)  // timestamps taken by the command processor at kernel begin / end (no launch gap included)
        hipExtLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, ev_start, ev_stop, 0, g, b, frame_t, (z_src << 16) | x_src, src_amp,
                              src_rxz, stf_grad_it, (lr.z << 16) | lr.x0, lr.n, lr.res);
    else
        hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, b, frame_t, (z_src << 16) | x_src, src_amp, src_rxz,
                           stf_grad_it, (lr.z << 16) | lr.x0, lr.n, lr.res);
}

static void (*persist_kernel(int lmask))(Grid, const PersistArgs) {
    switch (lmask) {
        case 0: return k_bwd_persist<0>;
        case 1: return k_bwd_persist<1>;
        case 3: return k_bwd_persist<3>;
        case 7: return k_bwd_persist<7>;
        case 15: return k_bwd_persist<15>;
        default: return nullptr;
    }
}

// 0, or why this grid cannot run the persistent loop (never launches a grid that would not be resident at once: its tiles wait for
// each other)
static int persist_config_ok(const void *k, int nwg, int threads, size_t lds_bytes) {
    if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
    int per_cu = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, lds_bytes) != hipSuccess) return -3;
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    if (per_cu * ncu < nwg) return -4;
    return 0;
}

int launch_bwd_persist(hipStream_t st, const Grid &g0, const KernelOptions &o, const PersistArgs &args, int nwg, int threads, int lmask,
                       size_t lds_bytes, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled(g0, o, 1);
    auto k = persist_kernel(lmask);
    if (!k) return -1;
#ifdef SEPFWI_PK_TRACE
    {
        static unsigned long long *d_tr = nullptr;
        static size_t tr_n = (size_t)kTrTiles * 16 * kTrPh * kTrSlots;
        if (!d_tr && getenv("SEPFWI_PK_TRACE")) {
            (void)hipMalloc((void **)&d_tr, tr_n * sizeof(unsigned long long));
            (void)hipMemset(d_tr, 0, tr_n * sizeof(unsigned long long));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pk_trace), &d_tr, sizeof(d_tr));
            atexit([] {
                std::vector<unsigned long long> h(tr_n);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h.data(), d_tr, tr_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (FILE *fp = fopen(getenv("SEPFWI_PK_TRACE"), "wb")) {
                    fwrite(h.data(), sizeof(unsigned long long), tr_n, fp);
                    fclose(fp);
                }
            });
        }
    }
#endif
    const int rc = persist_config_ok((const void *)k, nwg, threads, lds_bytes);
    if (rc) return rc;
    if (ev_start)
        hipExtLaunchKernelGGL(k, dim3(nwg), dim3(threads), lds_bytes, st, ev_start, ev_stop, 0, g, args);
    else
        hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), lds_bytes, st, g, args);
    return 0;
}

__global__ void k_add_inplace(float *__restrict__ a, const float *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}
void launch_add_inplace(hipStream_t st, float *a, const float *b, size_t n) {
    hipLaunchKernelGGL(k_add_inplace, dim3(4096), dim3(256), 0, st, a, b, n);
}

// ---- batched launchers: one grid over (tile, shot of the batch); the order of the two is Grid::nb / batch_order
static inline Grid tiled_batch(const Grid &g0, const KernelOptions &o, int fly_bit, int nb) {
    Grid g = tiled(g0, o, fly_bit);
    g.nb = nb;
    g.shot_fastest = o.batch_order;
    const int nblk = g.gx * g.gy * g.nb;
    g.nblk = g.xcd_remap ? ((nblk + 7) / 8) * 8 : nblk;
    return g;
}
static void check_shared_bundles(const Grid &g, const Media &md, const PmlCoef &pc) {
    const ptrdiff_t n = md.mu - md.lam, z = g.nzc, x = g.nx;
    const bool media_ok = n > 0 && md.ave_mu - md.mu == n && md.byc_a - md.ave_mu == n && md.byc_b - md.byc_a == n && md.rho - md.byc_b == n;
    const bool coef_ok = pc.b_z - pc.a_z == z && pc.rK_z - pc.b_z == z && pc.a_zh - pc.rK_z == z && pc.b_zh - pc.a_zh == z && pc.rK_zh - pc.b_zh == z &&
                         pc.a_x - pc.rK_zh == z && pc.b_x - pc.a_x == x && pc.rK_x - pc.b_x == x && pc.a_xh - pc.rK_x == x && pc.b_xh - pc.a_xh == x &&
                         pc.rK_xh - pc.b_xh == x;
    if (!(media_ok && coef_ok)) throw std::logic_error("batched kernel launch: media or C-PML profiles are not laid out as one bundle");
}
static inline dim3 batch_grid(const Grid &g) { return dim3(g.nblk); }
void launch_stress_fwd_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                             PmlCoef pc, size_t n, size_t data_len, int it, float src_scale, bool save) {
    const Grid g = tiled_batch(g0, o, 0, nb);
    check_shared_bundles(g, md, pc);
    const bool q = o.quiet_skip != 0;  // (per shot: ShotDev::quiet)
    auto k = save ? (q ? k_stress_fwd_batch<true, true> : k_stress_fwd_batch<true, false>) : (q ? k_stress_fwd_batch<false, true> : k_stress_fwd_batch<false, false>);
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, data_len, it, src_scale);
}
void launch_velocity_fwd_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                               PmlCoef pc, size_t n) {
    const Grid g = tiled_batch(g0, o, 0, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_velocity_fwd_batch<true> : k_velocity_fwd_batch<false>;
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n);
}
void launch_bwd_a_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it) {
    const Grid g = tiled_batch(g0, o, 1, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_bwd_a_batch<false, true> : (o.early & 1) ? k_bwd_a_batch<true, false> : k_bwd_a_batch<false, false>;
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, it);
}
void launch_bwd_b_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it, float src_scale, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled_batch(g0, o, 1, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_bwd_b_batch<false, true> : (o.early & 2) ? k_bwd_b_batch<true, false> : k_bwd_b_batch<false, false>;
    if (ev_start)
        hipExtLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, ev_start, ev_stop, 0, g, shots, md.lam, pc.a_z, n, it, src_scale);
    else
        hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, it, src_scale);
}

void launch_record(hipStream_t st, const Grid &g, Fields f, int nrec, const int *rec_idx, float *d_pr, float *d_vx,
                   float *d_vz, float *d_ett, int comps, const float *sens) {
    if (nrec <= 0) return;
    hipLaunchKernelGGL(k_record, dim3((nrec + 255) / 256), dim3(256), 0, st, g, f, nrec, rec_idx, d_pr, d_vx, d_vz, d_ett,
                       comps, g.fiber, sens);
}

void launch_inject(hipStream_t st, const Grid &g, Fields adj, int nrec, const int *rec_idx, const float *res_t, const float *sens) {
    if (nrec <= 0) return;
    hipLaunchKernelGGL(k_inject, dim3((nrec + 255) / 256), dim3(256), 0, st, adj, nrec, rec_idx, res_t, g.fiber ? g.pitch : 0,
                       sens, g.pitch, g.dx * g.rdz);
}

void launch_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, long long n,
                     double *sumsq) {
    if (n <= 0) return;  // a shot without receivers contributes nothing
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_residual, dim3(blocks), dim3(256), 0, st, obs, syn, res, nrec, n, sumsq);
}

void launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st, in, out, rows, cols);
}

void launch_model_prep(hipStream_t st, const Grid &g, const KernelOptions &o, const float *Lam_in, const float *Mu_in,
                       const float *Den_in, float *lam, float *mu, float *ave_mu, float *byc_a, float *byc_b, float *rho,
                       unsigned int *cp2_max_bits) {
    hipLaunchKernelGGL(k_model_prep, dim3((g.nx + 63) / 64, (g.nz + 3) / 4), dim3(64, 4), 0, st, g, Lam_in, Mu_in, Den_in,
                       lam, mu, ave_mu, byc_a, byc_b, rho, cp2_max_bits, o.amu_fly != 0 ? 1 : 0);
}

void launch_finalize_gradients(hipStream_t st, const Grid &g, Media md, ImgAcc acc, float *gLam, float *gMu,
                               float *gDen) {
    hipLaunchKernelGGL(k_finalize_gradients, dim3((g.nx + 63) / 64, (g.nz + 3) / 4), dim3(64, 4), 0, st, g, md, acc, gLam,
                       gMu, gDen);
}

}  // namespace sepfwi
