// kernels_bodies.hpp -- the four updates (stress, velocity, adjoint velocity, adjoint stress) as inline bodies
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

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
