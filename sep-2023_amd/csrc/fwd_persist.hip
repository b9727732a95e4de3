// fwd_persist.hip -- the forward time loop of ONE shot as ONE persistent launch (option fwd_fuse=3).
//
// One 1024-thread workgroup per CU owns a band of 2..5 consecutive rows of the grid and keeps its five wavefields in
// REGISTERS for the whole time loop (2000x1000: 256 bands x 4-5 rows x 2112 columns = 8-10 cells per thread).  Per half
// step it needs from outside only
//   * the two rows above and below of the two fields the other half step has just updated: handed over by the two
//     row-neighbours through global halo buffers -- write-through (sc1) stores, `s_waitcnt vmcnt(0)`, a step counter as
//     flag, sc1 polls and sc1 loads (the hand-off rules of MI355X_MICROARCH.md; measured on the bare pattern in
//     scripts/probes/persistent_stencil_probe.hip: 3.7 us per half step, bit-exact);
//   * x-neighbours: the band's rows of those two fields staged in LDS;
//   * coefficients, C-PML memory variables (strips only), boundary frames, seismogram columns: plain global accesses of
//     the owning thread.
// Neighbour-only dependencies make the in-place update legal (a band starts a half step only after both neighbours
// finished the previous one), which tile + halo recomputation cannot.  Replaces, per time step, the two
// launches k_stress<FWD> + k_velocity<FWD> (reference: 12 launches, Src/libCUFD.cu:268-332); arithmetic per cell is
// that of stress_body<FWD> / velocity_body<FWD> in kernels.hip (el_stress.cu:50-87, el_velocity.cu:45-82).
//
// Every spin is bounded and a global abort flag stops all bands, so the grid always drains.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace sepfwi {

using namespace dev;

namespace {

constexpr int PT = 1024;  // threads per workgroup (16 waves, 4 per SIMD -> 128 VGPRs each)
constexpr int RMAX = 5;   // rows per band

__device__ __forceinline__ int flag_load(const int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void flag_store(int *p, int v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void pay_store(float *p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float pay_load(const float *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// Wait until both row-neighbours have published step `want` (flags on 128-B lines of their own).  Thread 0 polls; false if the
// run was aborted (spin limit reached somewhere).
__device__ __forceinline__ bool wait_neighbours(const int *flag, int b, int nb, int want, int *abort_flag, int *ok_lds) {
    if (threadIdx.x == 0) {
        int good = 1;
        for (int q = b - 1; q <= b + 1 && good; q += 2) {
            if (q < 0 || q >= nb) continue;
            long spins = 0;
            while (flag_load(flag + q * 32) < want) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1L << 22) || flag_load(abort_flag)) {
                    good = 0;
                    flag_store(abort_flag, 1);
                    break;
                }
            }
        }
        *ok_lds = good;
    }
    __syncthreads();
    return *ok_lds != 0;
}

// halo buffers: [band][field 0/1][side 0 = first two rows, 1 = last two rows][row 0/1][pitch]
__device__ __forceinline__ size_t halo_off(int band, int field, int side, int row, int pitch) {
    return ((((size_t)band * 2 + field) * 2 + side) * 2 + row) * (size_t)pitch;
}

// value of a band-resident field at band row q (may be -2..nr+1): own registers, or -- read on demand, each is used once or
// twice -- the neighbours' published rows (upper neighbour's last two rows for q = -2, -1; lower neighbour's first two rows
// for q = nr, nr+1)
struct HaloRef {
    const float *up, *dn;  // this field's [side 1] rows of band b-1 and [side 0] rows of band b+1 (null: no neighbour)
    int pitch;
};
template <int C>
__device__ __forceinline__ float zget(const float (&own)[RMAX][C], const HaloRef &h, int q, int c, int x, int nr) {
    if (q < 0) return h.up ? pay_load(h.up + (size_t)(q + 2) * h.pitch + x) : 0.0f;
    if (q >= nr) return h.dn ? pay_load(h.dn + (size_t)(q - nr) * h.pitch + x) : 0.0f;
    return own[q < RMAX ? q : RMAX - 1][c];
}

// the same for a field that lives in LDS
__device__ __forceinline__ float zget_lds(const float *L, int LW, const HaloRef &h, int q, int x, int nr) {
    if (q < 0) return h.up ? pay_load(h.up + (size_t)(q + 2) * h.pitch + x) : 0.0f;
    if (q >= nr) return h.dn ? pay_load(h.dn + (size_t)(q - nr) * h.pitch + x) : 0.0f;
    return L[q * LW + 4 + x];
}

}  // namespace

struct PersistArgs {
    ShotDev s;
    const float *media;  // lam, mu, ave_mu, byc_a, byc_b, rho (stride n)
    const float *cz;     // six z profiles (stride nzc) then six x profiles (stride nx)
    size_t n, data_len;
    float src_scale;
    int nsteps;          // time steps to run (nSteps - 1)
    int nb;              // bands = workgroups
    int save;            // write boundary frames
    float *haloV, *haloS;  // velocity halos (vz, vx) and stress halos (szz, sxz)
    int *flagV, *flagS;    // steps completed per band, one 128-B line each
    int *abort_flag;
};

template <int C>
__global__ __launch_bounds__(PT) void k_fwd_persist(Grid g, PersistArgs a) {
    // LDS: [3][RMAX][LW].  tileA stages the band's rows of vz (stress half step) / sxz (velocity half step) for the x taps;
    // vx and sxx LIVE in LDS for the whole time loop (LVX, LSXX: own-cell read-modify-write by the owning thread, x taps by its
    // neighbours in the other half step) -- that takes 10 values per column off the register file.  vz, szz, sxz are registers.
    extern __shared__ float lds[];
    const int P = g.pitch, LW = P + 8;
    float *tileA = lds, *LVX = lds + RMAX * LW, *LSXX = lds + 2 * RMAX * LW;
    int *ok_lds = (int *)(lds + 3 * RMAX * LW);
    const int tid = threadIdx.x;
    int b = blockIdx.x;
    if ((a.nb & 7) == 0) b = (blockIdx.x & 7) * (a.nb >> 3) + (blockIdx.x >> 3);  // consecutive bands share an XCD (blockIdx % 8)
    const int z0 = (int)((long long)b * g.nzc / a.nb), z1 = (int)((long long)(b + 1) * g.nzc / a.nb), nr = z1 - z0;
    const bool has_up = b > 0, has_dn = b < a.nb - 1;
    const ShotDev &s = a.s;
    const size_t n = a.n;
    const float *lam_ = a.media, *mu_ = a.media + n, *amu_ = a.media + 2 * n, *byca_ = a.media + 3 * n, *bycb_ = a.media + 4 * n,
                *rho_ = a.media + 5 * n;
    const float *a_z = a.cz, *b_z = a.cz + g.nzc, *rK_z = a.cz + 2 * g.nzc, *a_zh = a.cz + 3 * g.nzc, *b_zh = a.cz + 4 * g.nzc,
                *rK_zh = a.cz + 5 * g.nzc;
    const float *cx = a.cz + 6 * g.nzc;
    const float *a_x = cx, *b_x = cx + g.nx, *rK_x = cx + 2 * g.nx, *a_xh = cx + 3 * g.nx, *b_xh = cx + 4 * g.nx, *rK_xh = cx + 5 * g.nx;
    float *m_dvz_dz = s.mem, *m_dvz_dx = s.mem + n, *m_dvx_dz = s.mem + 2 * n, *m_dvx_dx = s.mem + 3 * n;
    float *m_dszz_dz = s.mem + 4 * n, *m_dsxz_dx = s.mem + 5 * n, *m_dsxz_dz = s.mem + 6 * n, *m_dsxx_dx = s.mem + 7 * n;

    float vz[RMAX][C], szz[RMAX][C], sxz[RMAX][C];
#pragma unroll
    for (int r = 0; r < RMAX; r++)
#pragma unroll
        for (int c = 0; c < C; c++) vz[r][c] = szz[r][c] = sxz[r][c] = 0.0f;
    for (int k = tid; k < 3 * RMAX * LW; k += PT) lds[k] = 0.0f;  // fields start from zero; so do the pads left and right of every row

    for (int it = 0; it < a.nsteps; it++) {
        // =====================================================================================
        // stress half step (stress_body<FWD,SAVE>): needs vz, vx of the previous velocity half step
        // =====================================================================================
        if (!wait_neighbours(a.flagV, b, a.nb, it, a.abort_flag, ok_lds)) break;
        const HaloRef hvz{has_up ? a.haloV + halo_off(b - 1, 0, 1, 0, P) : nullptr, has_dn ? a.haloV + halo_off(b + 1, 0, 0, 0, P) : nullptr, P};
        const HaloRef hvx{has_up ? a.haloV + halo_off(b - 1, 1, 1, 0, P) : nullptr, has_dn ? a.haloV + halo_off(b + 1, 1, 0, 0, P) : nullptr, P};
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int x = tid + c * PT;
            const bool on = x < P;
#pragma unroll
            for (int r = 0; r < RMAX; r++)
                if (on) tileA[r * LW + 4 + x] = vz[r][c];
        }
        __syncthreads();
        {
            float *frame_t = a.save ? s.frame + (size_t)it * 5 * (size_t)g.frame_len : nullptr;
            const float amp = __fmul_rn(__fmul_rn(a.src_scale, s.stf[it]), g.dt);  // rounded like the host's float product
            const bool rec = (s.comps & 16) && it >= 1;  // line receivers: column `it` = velocities at the start of step `it`
            const size_t c0 = (size_t)it * (size_t)s.nrec;
            float *d_vx = (rec && (s.comps & 2)) ? s.syn + a.data_len + c0 : nullptr;
            float *d_vz = (rec && (s.comps & 4)) ? s.syn + 2 * a.data_len + c0 : nullptr;
            float *d_ett = (rec && (s.comps & 8)) ? s.syn + 3 * a.data_len + c0 : nullptr;
#pragma unroll
            for (int r = 0; r < RMAX; r++) {
                if (r >= nr) break;
                int z = z0 + r;
                asm volatile("" : "+s"(z));  // opaque per step: keeps the row's address arithmetic out of the loop-invariant set
                const bool pz = in_pml_z(g, z);
                const size_t row = (size_t)z * P;  // uniform: every access below is scalar row base + 32-bit lane offset
#pragma unroll
                for (int c = 0; c < C; c++) {
                    int x = tid + c * PT;
                    asm volatile("" : "+v"(x));  // likewise per cell: frame slots, strip tests, offsets are recomputed, not kept live
                    if (x >= g.nx) continue;
                    const size_t i = row + (unsigned)x;
                    if (frame_t) {  // boundary saving BEFORE this step's update (libCUFD.cu:271-273)
                        const int sl = frame_slot(g, z, x);
                        if (sl >= 0) {
                            const int L = g.frame_len;
                            frame_t[sl] = szz[r][c];
                            frame_t[L + sl] = sxz[r][c];
                            frame_t[2 * L + sl] = LSXX[r * LW + 4 + x];
                            frame_t[3 * L + sl] = vz[r][c];
                            frame_t[4 * L + sl] = LVX[r * LW + 4 + x];
                        }
                    }
                    if (z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3) continue;  // el_stress.cu:52
                    const float *ta = tileA + r * LW + 4 + x, *tb = LVX + r * LW + 4 + x;
                    const float vz0 = vz[r][c], vx0 = tb[0], vxm1 = tb[-1];
                    float dvz_dz = dminus(zget<C>(vz, hvz, r - 2, c, x, nr), zget<C>(vz, hvz, r - 1, c, x, nr), vz0,
                                          zget<C>(vz, hvz, r + 1, c, x, nr), g.rdz);
                    float dvx_dx = dminus(tb[-2], vxm1, vx0, tb[1], g.rdx);
                    float dvx_dz = dplus(zget_lds(LVX, LW, hvx, r - 1, x, nr), vx0, zget_lds(LVX, LW, hvx, r + 1, x, nr),
                                         zget_lds(LVX, LW, hvx, r + 2, x, nr), g.rdz);
                    float dvz_dx = dplus(ta[-1], vz0, ta[1], ta[2], g.rdx);
                    const float lam = lam_[i], mu = mu_[i], amu = amu_[i];
                    if (rec && z == s.lr_z) {  // recording_vx / _vz / _exx, utilities.cu:593-602,645-677
                        const int q = x - s.lr_x0;
                        if (q >= 0 && q < s.lr_n) {
                            if (d_vx) d_vx[q] = vx0;
                            if (d_vz) d_vz[q] = vz0;
                            if (d_ett) d_ett[q] = vx0 - vxm1;
                        }
                    }
                    if (pz) {
                        float p = b_z[z] * m_dvz_dz[i] + a_z[z] * dvz_dz;
                        m_dvz_dz[i] = p;
                        dvz_dz = dvz_dz * rK_z[z] + p;
                        float q = b_zh[z] * m_dvx_dz[i] + a_zh[z] * dvx_dz;
                        m_dvx_dz[i] = q;
                        dvx_dz = dvx_dz * rK_zh[z] + q;
                    }
                    if (x < g.nPml || x > g.nx - g.nPml - 1) {  // el_stress.cu:61,77
                        float p = b_x[x] * m_dvx_dx[i] + a_x[x] * dvx_dx;
                        m_dvx_dx[i] = p;
                        dvx_dx = dvx_dx * rK_x[x] + p;
                        float q = b_xh[x] * m_dvz_dx[i] + a_xh[x] * dvz_dx;
                        m_dvz_dx[i] = q;
                        dvz_dx = dvz_dx * rK_xh[x] + q;
                    }
                    const float l2m = lam + 2.0f * mu;
                    float nzz = szz[r][c] + (l2m * dvz_dz + lam * dvx_dx) * g.dt;
                    float nxx = LSXX[r * LW + 4 + x] + (lam * dvz_dz + l2m * dvx_dx) * g.dt;
                    if (z == s.z_src && x == s.x_src) {  // add_source, utilities.cu:531-538
                        nzz += amp;
                        nxx += amp;
                    }
                    szz[r][c] = nzz;
                    LSXX[r * LW + 4 + x] = nxx;
                    sxz[r][c] = sxz[r][c] + amu * (dvx_dz + dvz_dx) * g.dt;
                    __builtin_amdgcn_sched_barrier(0);  // one cell at a time: keeps the live set at state + one cell's temporaries
                }
            }
        }
        __syncthreads();  // staged rows consumed
        // publish szz (field 0) and sxz (field 1): first two and last two rows, write-through
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int x = tid + c * PT;
            if (x < P) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    pay_store(a.haloS + halo_off(b, 0, 0, k, P) + x, szz[k][c]);
                    pay_store(a.haloS + halo_off(b, 1, 0, k, P) + x, sxz[k][c]);
                    const int rl = nr - 2 + k;
                    float l0 = 0.f, l1 = 0.f;
#pragma unroll
                    for (int r = 0; r < RMAX; r++)
                        if (r == rl) {
                            l0 = szz[r][c];
                            l1 = sxz[r][c];
                        }
                    pay_store(a.haloS + halo_off(b, 0, 1, k, P) + x, l0);
                    pay_store(a.haloS + halo_off(b, 1, 1, k, P) + x, l1);
                }
                // stage sxz for the velocity half step's x taps (sxx lives in LDS)
#pragma unroll
                for (int r = 0; r < RMAX; r++) tileA[r * LW + 4 + x] = sxz[r][c];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // write-through stores drained before the flag
        __syncthreads();
        if (tid == 0) flag_store(a.flagS + b * 32, it + 1);

        // =====================================================================================
        // velocity half step (velocity_body<FWD>): needs szz, sxz of this step's stress half step
        // =====================================================================================
        if (!wait_neighbours(a.flagS, b, a.nb, it + 1, a.abort_flag, ok_lds)) break;
        const HaloRef hzz{has_up ? a.haloS + halo_off(b - 1, 0, 1, 0, P) : nullptr, has_dn ? a.haloS + halo_off(b + 1, 0, 0, 0, P) : nullptr, P};
        const HaloRef hxz{has_up ? a.haloS + halo_off(b - 1, 1, 1, 0, P) : nullptr, has_dn ? a.haloS + halo_off(b + 1, 1, 0, 0, P) : nullptr, P};
#pragma unroll
        for (int r = 0; r < RMAX; r++) {
            if (r >= nr) break;
            int z = z0 + r;
            asm volatile("" : "+s"(z));
            const bool pz = in_pml_z(g, z);
            const size_t row = (size_t)z * P;
#pragma unroll
            for (int c = 0; c < C; c++) {
                int x = tid + c * PT;
                asm volatile("" : "+v"(x));
                if (x >= g.nx) continue;
                if (z < 2 || z > g.nzc - 3 || x < 2 || x > g.nx - 3) continue;  // el_velocity.cu:47
                const size_t i = row + (unsigned)x;
                const float *ta = tileA + r * LW + 4 + x, *tb = LSXX + r * LW + 4 + x;  // sxz, sxx
                const float szz0 = szz[r][c], sxz0 = sxz[r][c], sxx0 = tb[0];
                float dszz_dz = dplus(zget<C>(szz, hzz, r - 1, c, x, nr), szz0, zget<C>(szz, hzz, r + 1, c, x, nr),
                                      zget<C>(szz, hzz, r + 2, c, x, nr), g.rdz);
                float dsxz_dx = dminus(ta[-2], ta[-1], sxz0, ta[1], g.rdx);
                float dsxz_dz = dminus(zget<C>(sxz, hxz, r - 2, c, x, nr), zget<C>(sxz, hxz, r - 1, c, x, nr), sxz0,
                                       zget<C>(sxz, hxz, r + 1, c, x, nr), g.rdz);
                float dsxx_dx = dplus(tb[-1], sxx0, tb[1], tb[2], g.rdx);
                float ba, bb;
                if (g.rho_fly) {
                    const float r0 = rho_[i];
                    ba = 2.0f / (rho_[i + P] + r0);
                    bb = 2.0f / (rho_[i + 1] + r0);
                } else {
                    ba = byca_[i];
                    bb = bycb_[i];
                }
                if (pz) {
                    float p = b_zh[z] * m_dszz_dz[i] + a_zh[z] * dszz_dz;
                    m_dszz_dz[i] = p;
                    dszz_dz = dszz_dz * rK_zh[z] + p;
                    float q = b_z[z] * m_dsxz_dz[i] + a_z[z] * dsxz_dz;
                    m_dsxz_dz[i] = q;
                    dsxz_dz = dsxz_dz * rK_z[z] + q;
                }
                if (x < g.nPml || x > g.nx - g.nPml) {  // el_velocity.cu:56,71 (one column narrower on the right)
                    float p = b_x[x] * m_dsxz_dx[i] + a_x[x] * dsxz_dx;
                    m_dsxz_dx[i] = p;
                    dsxz_dx = dsxz_dx * rK_x[x] + p;
                    float q = b_xh[x] * m_dsxx_dx[i] + a_xh[x] * dsxx_dx;
                    m_dsxx_dx[i] = q;
                    dsxx_dx = dsxx_dx * rK_xh[x] + q;
                }
                vz[r][c] = vz[r][c] + (dszz_dz + dsxz_dx) * ba * g.dt;
                LVX[r * LW + 4 + x] = LVX[r * LW + 4 + x] + (dsxz_dz + dsxx_dx) * bb * g.dt;
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        __syncthreads();  // staged rows consumed (the next stress half step overwrites the tiles)
        // publish vz (field 0) and vx (field 1)
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int x = tid + c * PT;
            if (x < P) {
#pragma unroll
                for (int k = 0; k < 2; k++) {
                    pay_store(a.haloV + halo_off(b, 0, 0, k, P) + x, vz[k][c]);
                    pay_store(a.haloV + halo_off(b, 1, 0, k, P) + x, LVX[k * LW + 4 + x]);
                    const int rl = nr - 2 + k;
                    float l0 = 0.f;
#pragma unroll
                    for (int r = 0; r < RMAX; r++)
                        if (r == rl) l0 = vz[r][c];
                    pay_store(a.haloV + halo_off(b, 0, 1, k, P) + x, l0);
                    pay_store(a.haloV + halo_off(b, 1, 1, k, P) + x, LVX[rl * LW + 4 + x]);
                }
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) flag_store(a.flagV + b * 32, it + 1);
    }

    // final state back to the field arrays (the backward pass and the last seismogram column start from it)
    float *F = s.fields;
#pragma unroll
    for (int r = 0; r < RMAX; r++) {
        if (r >= nr) break;
#pragma unroll
        for (int c = 0; c < C; c++) {
            const int x = tid + c * PT;
            if (x >= P) continue;
            const size_t i = (size_t)(z0 + r) * P + x;
            F[i] = vz[r][c];
            F[n + i] = LVX[r * LW + 4 + x];
            F[2 * n + i] = szz[r][c];
            F[3 * n + i] = LSXX[r * LW + 4 + x];
            F[4 * n + i] = sxz[r][c];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------------
int persist_bands(const Grid &g, int n_cus) {
    // bands of 2..RMAX rows, one workgroup each, all co-resident: at most one per CU
    int nb = g.nzc / 2;
    if (nb > n_cus) nb = n_cus;
    if (nb >= 8) nb &= ~7;  // multiple of 8: consecutive bands on one XCD
    if (nb < 1) return 0;
    if ((g.nzc + nb - 1) / nb > RMAX) return 0;  // grid too tall for register-resident bands
    if (g.pitch > 3 * PT) return 0;              // too wide: more than 3 columns per thread
    return nb;
}
size_t persist_halo_floats(const Grid &g, int nb) { return (size_t)nb * 2 * 2 * 2 * (size_t)g.pitch; }

bool launch_fwd_persist(hipStream_t st, const Grid &g0, const ShotDev &shot, Media md, PmlCoef pc, size_t n, size_t data_len,
                        float src_scale, int nsteps, int nb, bool save, float *haloV, float *haloS, int *flagV, int *flagS,
                        int *abort_flag, int rho_fly) {
    Grid g = g0;
    g.rho_fly = rho_fly;
    PersistArgs a{};
    a.s = shot;
    a.media = md.lam;
    a.cz = pc.a_z;
    a.n = n;
    a.data_len = data_len;
    a.src_scale = src_scale;
    a.nsteps = nsteps;
    a.nb = nb;
    a.save = save ? 1 : 0;
    a.haloV = haloV;
    a.haloS = haloS;
    a.flagV = flagV;
    a.flagS = flagS;
    a.abort_flag = abort_flag;
    const size_t shmem = (size_t)(3 * RMAX * (g.pitch + 8) + 4) * sizeof(float);
    const int C = (g.pitch + PT - 1) / PT;
    auto go = [&](auto kern) -> bool {
        if (shmem > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem) != hipSuccess)
            return false;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, PT, shmem) != hipSuccess || occ < 1) return false;
        hipLaunchKernelGGL(kern, dim3(nb), dim3(PT), shmem, st, g, a);
        return hipGetLastError() == hipSuccess;
    };
    if (C == 1) return go(k_fwd_persist<1>);
    if (C == 2) return go(k_fwd_persist<2>);
    if (C == 3) return go(k_fwd_persist<3>);
    return false;
}

}  // namespace sepfwi
