// fwd_march.hip -- fused forward time step, wave-autonomous z-marching form (gfx950, wave64).
//
// Same job as fwd_fused.hip (one launch per forward step: from_bnd x5 -> el_stress -> add_source ->
// el_velocity -> recorders of Src/libCUFD.cu:268-332; 60 algorithmic bytes per cell) with a different
// execution structure, chosen after profiling the LDS-tiled version:
//
//   * a WAVE (not a block) owns a strip of 64 columns x CZ rows and marches down it one row per iteration;
//   * the column neighbours (z+-1, z+-2) of every field live in per-lane REGISTER WINDOWS that slide with
//     the march, so each v_old / sigma_old / coefficient value is loaded from HBM exactly once per wave;
//   * the new stresses never leave the registers: the velocity update of row j-2 takes its z-neighbours
//     from the sigma_new windows and its x-neighbours from the adjacent lanes (ds_bpermute shuffles);
//   * no LDS allocation, no barrier, no inter-wave dependence: every wave is an independent stream, and the
//     loads of row j+1 are issued before row j is computed (software prefetch), so a handful of waves per CU
//     keeps enough bytes in flight.
// Redundancy: lanes 0,1,62,63 only feed their neighbours (60 of 64 columns are outputs) and each chunk
// recomputes 2+2 stress rows of its neighbours.  Fields are double-buffered exactly as in fwd_fused.hip.
#include <hip/hip_runtime.h>

#include "device_common.hpp"
#include "kernels.hpp"

namespace sepfwi {

using namespace dev;

namespace {

constexpr int MW_X = 60;  // output columns per wave

__device__ __forceinline__ float ldg(const float *base, unsigned off, int imm = 0) {
    return *reinterpret_cast<const float *>(reinterpret_cast<const char *>(base) + off + imm);
}
__device__ __forceinline__ void stg(float *base, unsigned off, float v) {
    *reinterpret_cast<float *>(reinterpret_cast<char *>(base) + off) = v;
}

// Everything one march iteration needs from memory: stress row j and velocity row j-2.
struct RowIn {
    float vz_n, vx_n;              // vz_old(j+1, x), vx_old(j+2, x): the rows entering the windows
    float vz_xm1, vz_xp1, vz_xp2;  // vz_old(j, x-1), (j, x+1), (j, x+2)
    float vx_xm2, vx_xm1, vx_xp1;  // vx_old(j, x-2), (j, x-1), (j, x+1)
    float szz, sxx, sxz;           // sigma_old(j, x)
    float lam, mu, amu;            // media at (j, x)
    float ba, bb;                  // buoyancies at (j-2, x)
};

}  // namespace

template <bool SAVE>
__global__ __launch_bounds__(64) void k_fwd_march(Grid g, FwdFusedArgs a, int CZ, int nstrips, int nchunks, LineRec lr) {
    int t = blockIdx.x;
    if (g.xcd_remap) {
        const int per = (nstrips * nchunks + 7) >> 3;
        t = (t & 7) * per + (t >> 3);
    }
    if (t >= nstrips * nchunks) return;
    const int chunk = t / nstrips, strip = t - chunk * nstrips;
    const int lane = threadIdx.x;
    const int x = strip * MW_X - 2 + lane;
    const int zc0 = chunk * CZ;
    const int zc1 = min(zc0 + CZ, g.nzc);  // rows [zc0, zc1) are this wave's outputs
    const unsigned n = a.n;
    const int P = g.pitch;
    const unsigned P4 = 4u * (unsigned)P;
    const int nzc = g.nzc, nx = g.nx;

    const float *o_vz = a.fo, *o_vx = a.fo + n, *o_szz = a.fo + 2 * n, *o_sxx = a.fo + 3 * n, *o_sxz = a.fo + 4 * n;
    float *n_vz = a.fn, *n_vx = a.fn + n, *n_szz = a.fn + 2 * n, *n_sxx = a.fn + 3 * n, *n_sxz = a.fn + 4 * n;
    const float *c_lam = a.media, *c_mu = a.media + n, *c_amu = a.media + 2 * n, *c_ba = a.media + 3 * n, *c_bb = a.media + 4 * n;
    const float *cz = a.cz, *cx = a.cx;

    const bool col_in = (x >= 0 && x < nx);
    const bool col_comp = (x >= 2 && x <= nx - 3);                 // el_stress.cu:52
    const bool own_col = col_in && lane >= 2 && lane < MW_X + 2;
    // A strip that touches an x-PML applies the C-PML recursion to ALL its lanes: outside the layer a = 0, b = 1,
    // 1/K = 1 and psi stays 0, so the result is unchanged -- and the test becomes wave-uniform.
    const int x_lo = strip * MW_X - 2, x_hi = x_lo + 63;
    const bool strip_px = (x_lo < g.nPml) || (x_hi > nx - g.nPml - 1);  // uniform
    // every load uses a clamped, always-valid address; out-of-range lanes / rows are masked AFTER the load so the
    // march body is branch-free (predicated loads cost one branch each and serialise the prefetch)
    const int xc = min(max(x, 2), nx - 3);
    const unsigned xoff = 4u * (unsigned)xc;

    float ax = cx[xc], bx = cx[nx + xc], rKx = cx[2 * nx + xc];
    float axh = cx[3 * nx + xc], bxh = cx[4 * nx + xc], rKxh = cx[5 * nx + xc];

    auto row_off = [&](int z) -> unsigned { return (unsigned)min(max(z, 0), nzc - 1) * P4 + xoff; };

    auto load_row = [&](int j) -> RowIn {
        RowIn r;
        r.vz_n = ldg(o_vz, row_off(j + 1));
        r.vx_n = ldg(o_vx, row_off(j + 2));
        const unsigned off = row_off(j);
        r.vz_xm1 = ldg(o_vz, off, -4);
        r.vz_xp1 = ldg(o_vz, off, 4);
        r.vz_xp2 = ldg(o_vz, off, 8);
        r.vx_xm2 = ldg(o_vx, off, -8);
        r.vx_xm1 = ldg(o_vx, off, -4);
        r.vx_xp1 = ldg(o_vx, off, 4);
        r.szz = ldg(o_szz, off);
        r.sxx = ldg(o_sxx, off);
        r.sxz = ldg(o_sxz, off);
        r.lam = ldg(c_lam, off);
        r.mu = ldg(c_mu, off);
        r.amu = ldg(c_amu, off);
        const unsigned offv = row_off(j - 2);
        r.ba = ldg(c_ba, offv);
        r.bb = ldg(c_bb, offv);
        return r;
    };
    // rows outside the grid hold zeros: mask the window-filling loads (uniform selects)
    auto in_rows = [&](int z) -> float { return (z >= 0 && z < nzc) ? 1.0f : 0.0f; };

    // ---- prologue: windows for the first stress row j0 = zc0 - 2 ----
    const int j0 = zc0 - 2, j1 = zc1 + 1;  // stress rows j0..j1; velocity rows j0+2 .. j1-2 = zc0 .. zc1-1
    const float cm = col_comp ? 1.0f : 0.0f;  // columns outside [2, nx-3] are never computed: they hold zeros
    float vzw0 = cm * in_rows(j0 - 2) * ldg(o_vz, row_off(j0 - 2)), vzw1 = cm * in_rows(j0 - 1) * ldg(o_vz, row_off(j0 - 1)),
          vzw2 = cm * in_rows(j0) * ldg(o_vz, row_off(j0)), vzw3;
    float vxw0 = cm * in_rows(j0 - 2) * ldg(o_vx, row_off(j0 - 2)), vxw1 = cm * in_rows(j0 - 1) * ldg(o_vx, row_off(j0 - 1)),
          vxw2 = cm * in_rows(j0) * ldg(o_vx, row_off(j0)), vxw3 = cm * in_rows(j0 + 1) * ldg(o_vx, row_off(j0 + 1)), vxw4;
    float szzw0 = 0.f, szzw1 = 0.f, szzw2 = 0.f, szzw3 = 0.f;                // szz_new(j-3..j)
    float sxzw0 = 0.f, sxzw1 = 0.f, sxzw2 = 0.f, sxzw3 = 0.f, sxzw4 = 0.f;  // sxz_new(j-4..j)
    float sxxw0 = 0.f, sxxw1 = 0.f, sxxw2 = 0.f;                            // sxx_new(j-2..j)

    // one march iteration: stress row j and velocity row j-2 from the prefetched inputs `cur`
    auto body = [&](const int j, const RowIn &cur) {
        vzw3 = (j + 1 >= 0 && j + 1 < nzc && col_comp) ? cur.vz_n : 0.f;  // vz_old(j+1)
        vxw4 = (j + 2 >= 0 && j + 2 < nzc && col_comp) ? cur.vx_n : 0.f;  // vx_old(j+2)

        // ---------------- stress row j ----------------
        const bool rin = (j >= 0 && j < nzc);                       // uniform
        const bool own_row = (j >= zc0 && j < zc1);                  // uniform
        const bool comp = rin && j >= 2 && j <= nzc - 3 && col_comp;
        const unsigned off = row_off(j);
        // cells outside the computed region hold zeros (their clamped loads may have fetched a neighbour)
        const float szz_o = comp ? cur.szz : 0.f, sxx_o = comp ? cur.sxx : 0.f, sxz_o = comp ? cur.sxz : 0.f;
        if constexpr (SAVE) {
            if (own_row && own_col) {  // boundary saving of the state BEFORE the update (libCUFD.cu:271-273)
                const int s = frame_slot(g, j, x);
                if (s >= 0) {
                    const int L = g.frame_len;
                    a.frame_t[s] = szz_o;
                    a.frame_t[L + s] = sxz_o;
                    a.frame_t[2 * L + s] = sxx_o;
                    a.frame_t[3 * L + s] = vzw2;
                    a.frame_t[4 * L + s] = vxw2;
                }
            }
        }
        if (lr.n && j == lr.z && own_row) {  // uniform test
            // line receivers: column `it` = velocities at the START of the step (utilities.cu:593-602,645-677)
            const int r = x - lr.x0;
            if (own_col && r >= 0 && r < lr.n) {
                if (lr.d_vx) lr.d_vx[r] = vxw2;
                if (lr.d_vz) lr.d_vz[r] = vzw2;
                if (lr.d_ett) lr.d_ett[r] = vxw2 - cur.vx_xm1;
            }
        }
        float szz, sxx, sxz;
        {
            float dvz_dz = dminus(vzw0, vzw1, vzw2, vzw3, g.rdz);
            float dvx_dx = dminus(cur.vx_xm2, cur.vx_xm1, vxw2, cur.vx_xp1, g.rdx);
            float dvx_dz = dplus(vxw1, vxw2, vxw3, vxw4, g.rdz);
            float dvz_dx = dplus(cur.vz_xm1, vzw2, cur.vz_xp1, cur.vz_xp2, g.rdx);
            const bool store = own_row && own_col && comp;
            if (rin && in_pml_z(g, j)) {  // uniform
                const float p = cz[nzc + j] * ldg(a.mo, off) + cz[j] * dvz_dz;               // psi(dvz_dz)
                dvz_dz = dvz_dz * cz[2 * nzc + j] + p;
                const float q = cz[4 * nzc + j] * ldg(a.mo + 2 * n, off) + cz[3 * nzc + j] * dvx_dz;  // psi(dvx_dz)
                dvx_dz = dvx_dz * cz[5 * nzc + j] + q;
                if (store) {
                    stg(a.mn, off, p);
                    stg(a.mn + 2 * n, off, q);
                }
            }
            if (strip_px) {  // uniform
                const float p = bx * ldg(a.mo + 3 * n, off) + ax * dvx_dx;                     // psi(dvx_dx)
                dvx_dx = dvx_dx * rKx + p;
                const float q = bxh * ldg(a.mo + n, off) + axh * dvz_dx;                       // psi(dvz_dx)
                dvz_dx = dvz_dx * rKxh + q;
                if (store) {
                    stg(a.mn + 3 * n, off, p);
                    stg(a.mn + n, off, q);
                }
            }
            const float l2m = cur.lam + 2.0f * cur.mu;
            float nzz = szz_o + (l2m * dvz_dz + cur.lam * dvx_dx) * g.dt;
            float nxx = sxx_o + (cur.lam * dvz_dz + l2m * dvx_dx) * g.dt;
            if (j == a.z_src && x == a.x_src) {  // add_source, utilities.cu:531-538
                nzz += a.src_amp;
                nxx += a.src_amp;
            }
            const float nxz = sxz_o + cur.amu * (dvx_dz + dvz_dx) * g.dt;
            szz = comp ? nzz : szz_o;
            sxx = comp ? nxx : sxx_o;
            sxz = comp ? nxz : sxz_o;
            if (store) {
                stg(n_szz, off, szz);
                stg(n_sxx, off, sxx);
                stg(n_sxz, off, sxz);
            }
        }
        szzw3 = szz;
        sxzw4 = sxz;
        sxxw2 = sxx;

        // ---------------- velocity row jv = j - 2 (new stresses of rows jv-2 .. jv+2 are in the windows) ----------------
        const int jv = j - 2;
        if (jv >= zc0 && jv < zc1 && jv >= 2 && jv <= nzc - 3) {  // uniform
            // x-neighbours of the new stresses of row jv come from the adjacent lanes
            const float sxz_c = sxzw2, sxx_c = sxxw0;
            const float sxz_m2 = __shfl_up(sxz_c, 2), sxz_m1 = __shfl_up(sxz_c, 1), sxz_p1 = __shfl_down(sxz_c, 1);
            const float sxx_m1 = __shfl_up(sxx_c, 1), sxx_p1 = __shfl_down(sxx_c, 1), sxx_p2 = __shfl_down(sxx_c, 2);
            const unsigned offv = row_off(jv);
            float dszz_dz = dplus(szzw0, szzw1, szzw2, szzw3, g.rdz);
            float dsxz_dx = dminus(sxz_m2, sxz_m1, sxz_c, sxz_p1, g.rdx);
            float dsxz_dz = dminus(sxzw0, sxzw1, sxzw2, sxzw3, g.rdz);
            float dsxx_dx = dplus(sxx_m1, sxx_c, sxx_p1, sxx_p2, g.rdx);
            const bool vstore = own_col && col_comp;
            if (in_pml_z(g, jv)) {  // uniform
                const float p = cz[4 * nzc + jv] * ldg(a.mv, offv) + cz[3 * nzc + jv] * dszz_dz;      // psi(dszz_dz)
                dszz_dz = dszz_dz * cz[5 * nzc + jv] + p;
                const float q = cz[nzc + jv] * ldg(a.mv + 2 * n, offv) + cz[jv] * dsxz_dz;            // psi(dsxz_dz)
                dsxz_dz = dsxz_dz * cz[2 * nzc + jv] + q;
                if (vstore) {
                    stg(a.mv, offv, p);
                    stg(a.mv + 2 * n, offv, q);
                }
            }
            if (strip_px) {  // uniform; el_velocity.cu:56,71 tests x > nx-nPml (one column narrower than the stress test)
                const bool pxv = (x < g.nPml || x > nx - g.nPml);
                const float p = bx * ldg(a.mv + n, offv) + ax * dsxz_dx;                              // psi(dsxz_dx)
                const float q = bxh * ldg(a.mv + 3 * n, offv) + axh * dsxx_dx;                        // psi(dsxx_dx)
                if (pxv) {
                    dsxz_dx = dsxz_dx * rKx + p;
                    dsxx_dx = dsxx_dx * rKxh + q;
                    if (vstore) {
                        stg(a.mv + n, offv, p);
                        stg(a.mv + 3 * n, offv, q);
                    }
                }
            }
            if (vstore) {
                stg(n_vz, offv, vzw0 + (dszz_dz + dsxz_dx) * cur.ba * g.dt);   // vzw0 = vz_old(jv)
                stg(n_vx, offv, vxw0 + (dsxz_dz + dsxx_dx) * cur.bb * g.dt);   // vxw0 = vx_old(jv)
            }
        }

        // ---------------- slide the windows ----------------
        vzw0 = vzw1; vzw1 = vzw2; vzw2 = vzw3;
        vxw0 = vxw1; vxw1 = vxw2; vxw2 = vxw3; vxw3 = vxw4;
        szzw0 = szzw1; szzw1 = szzw2; szzw2 = szzw3;
        sxzw0 = sxzw1; sxzw1 = sxzw2; sxzw2 = sxzw3; sxzw3 = sxzw4;
        sxxw0 = sxxw1; sxxw1 = sxxw2;
    };

    // Software pipeline, two row buffers in ping-pong (no register copies, so no wait at the loop end): the
    // loads of row j+1 are in flight while row j is computed, those of row j+2 while row j+1 is computed.
    RowIn A = load_row(j0);
    for (int j = j0; j <= j1; j += 2) {
        const RowIn B = load_row(j + 1);
        body(j, A);
        A = load_row(j + 2);
        if (j + 1 <= j1) body(j + 1, B);
    }
}

void launch_fwd_march(hipStream_t st, const Grid &g0, const FwdFusedArgs &a, LineRec lr, int xcd_remap) {
    Grid g = g0;
    g.xcd_remap = xcd_remap;
    const int nstrips = (g.nx + MW_X - 1) / MW_X;
    int nchunks = get_kernel_option("march_waves") / nstrips;  // waves in flight: enough per SIMD to overlap issue and memory
    if (nchunks < 1) nchunks = 1;
    int CZ = (g.nzc + nchunks - 1) / nchunks;
    if (CZ < 8) CZ = 8;
    nchunks = (g.nzc + CZ - 1) / CZ;
    const int nb = nstrips * nchunks;
    const dim3 grid(xcd_remap ? ((nb + 7) / 8) * 8 : nb);
    if (a.frame_t)
        hipLaunchKernelGGL((k_fwd_march<true>), grid, dim3(64), 0, st, g, a, CZ, nstrips, nchunks, lr);
    else
        hipLaunchKernelGGL((k_fwd_march<false>), grid, dim3(64), 0, st, g, a, CZ, nstrips, nchunks, lr);
}

}  // namespace sepfwi
