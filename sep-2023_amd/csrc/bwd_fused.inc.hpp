// bwd_fused.inc.hpp -- included by kernels.hip (inside namespace sepfwi, after the field-kernel bodies).
//
// The whole backward time step in ONE launch (option bwd_fuse = 3).  Per step the reference runs
//   source_grad -> el_velocity(back) -> to_bnd(v) -> add_source(-) -> el_stress(back) -> to_bnd(s) -> el_velocity_adj ->
//   res_injection_exx -> el_stress_adj                                                    (Src/libCUFD.cu:545-631)
// and the two-launch form (k_bwd_a, k_bwd_b) passes every one of the ten wavefields three times through the memory
// system per step: once as the stencil input of the sibling kernel, once as read-modify-write of its owner.  Here a
// workgroup owns a tile of 64 x NO cells and computes
//   stage 1   reverse-time VELOCITY of step t (+ rho imaging, frame restore)  and  adjoint STRESS of step t+1
//             on the tile widened by 2 cells (68 columns x NR = NO + 4 rows), results to LDS.  A wave walks its R1 rows one
//             after the other (64 lanes = the 64 owned, line-aligned columns), then spends ONE more pass on the 4 halo
//             columns of all its rows (4 R1 lanes active).
//   barrier   (LDS only: loads stay in flight across it)
//   stage 2   source_grad + reverse-time STRESS (+ lambda/mu imaging, frame restore) + adjoint VELOCITY + line injection
//             on the tile, stencil taps of the stage-1 results from LDS; every global store of the step happens here, as
//             full 256-byte row segments
// so every field is read once and written once per step (plus the halo re-reads, which hit L1 / L2).  A tile cannot update
// in place -- a neighbour's halo recomputation needs the step-start values -- so all state ping-pongs between two sets
// of [5 fields | 8 memory variables | 5 adjoint fields] (set_in is only read, set_out only written); the imaging
// accumulators are owned cell by cell and stay in place.
//
// Addressing: every array of a set is reached through ONE buffer descriptor and ONE 32-bit per-lane byte offset (the cell
// two rows up and two columns left of the thread's cell); the array and the row of a tap go into the scalar offset, the
// column into the instruction's immediate.  No 64-bit per-array address pairs: the kernel has to hold the loads of two
// update bodies in registers and still fit 8 waves per SIMD.
//
// C-PML: stage 1's adjoint stress reads the velocity-type memory variables (psi_s: dszz_dz ...) through stencils from the
// IN set, exactly as k_bwd_a does.  Stage 2's adjoint velocity needs stencils of the stress-type memory variables (psi_v:
// dvz_dz ...) AS UPDATED in stage 1 by the neighbouring cells.  Those are not staged in LDS (strips are 6 % of the cells;
// four more LDS arrays would cost every workgroup its occupancy): a strip cell rebuilds psi_v_new at each tap from the IN
// value, the tap's coefficients and the tap's new adjoint stresses in LDS -- the very expression its owner evaluates, so the
// bits are the same.
//
// Expressions and their order are those of stress_body<false>, velocity_body<false>, stress_adj_apply and
// velocity_adj_apply above (tests/test_gpu_parity.py::test_kernel_structures_are_bit_identical).

struct FusedArgs {   // everything the kernel needs and nothing else: its scalar registers are the scarce resource
    const float *set_in;   // [vz vx szz sxx sxz | dvz_dz dvz_dx dvx_dz dvx_dx dszz_dz dsxz_dx dsxz_dz dsxx_dx | adjoint vz vx szz sxx sxz]
    float *set_out;        // same layout, the other ping-pong set
    const float *media;    // lam, mu, ave_mu, byc_a, byc_b, rho
    float *acc;            // lam, mu, xz, a, b (in place)
    const float *cz;       // z profiles a, b, 1/K, a_half, b_half, 1/K_half (stride nzc), then the six x profiles (stride nx)
    const float *frame_t;  // this step's 5 * frame_len block
    float *stf_grad_it;
    const float *lr_res;
    unsigned nb;           // bytes per array (the stride of every bundle)
    int nzc, nx, pitch, nPml;
    int nzBnd, nxBnd, frame_len;
    int gx, ntile;         // tiles per row of tiles, tiles in all
    float dt, rdz, rdx;
    int zx_src;
    float src_amp, src_rxz;
    int lr_zx, lr_n;
    int dbg;               // timing experiments only
};

constexpr int FUSED_NXO = 64;  // owned columns per tile: one full, aligned wave row; the region adds two halo columns on either side

enum : int { S_VZ = 0, S_VX, S_SZZ, S_SXX, S_SXZ, S_DVZ_DZ, S_DVZ_DX, S_DVX_DZ, S_DVX_DX, S_DSZZ_DZ, S_DSXZ_DX, S_DSXZ_DZ, S_DSXX_DX,
              S_AVZ, S_AVX, S_ASZZ, S_ASXX, S_ASXZ };
enum : int { M_LAM = 0, M_MU, M_AMU, M_BYA, M_BYB, M_RHO };
enum : int { A_LAM = 0, A_MU, A_XZ, A_A, A_B };

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float bld(rsrc_t r, int voff, int soff) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
__device__ __forceinline__ void bst(rsrc_t r, float v, int voff, int soff) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

template <int W, int R1>
__global__ __launch_bounds__(64 * W) void k_bwd_fused(FusedArgs g) {
    constexpr int NR = W * R1;        // region rows (stage 1)
    constexpr int NO = NR - 4;        // owned rows (stage 2)
    constexpr int R2 = NO / W;        // owned rows per wave in stage 2
    static_assert(NO % W == 0 && 4 * R1 <= 64, "tile shape");
    constexpr int LC = 68;            // region columns: 2 halo + 64 owned + 2 halo
    __shared__ float lds[5][NR][LC];  // new vz, vx (reconstruction) | new adjoint szz, sxx, sxz
    const int z_src = g.zx_src >> 16, x_src = g.zx_src & 0xffff;
    const LineRec lr{g.lr_zx >> 16, g.lr_zx & 0xffff, g.lr_n, nullptr, nullptr, nullptr, g.lr_res};
    auto pml_z = [&](int z) { return z < g.nPml || z > g.nzc - g.nPml - 1; };
    // C-PML profiles (a, b, 1/K, a_half, b_half, 1/K_half for z, then for x): indexed where they are used, so that no
    // pointer of the twelve lives in scalar registers outside the strips
    enum : int { C_A = 0, C_B, C_RK, C_AH, C_BH, C_RKH };
    auto CZ = [&](int k, int zz) { return g.cz[k * g.nzc + zz]; };
    auto CX = [&](int k, int xx) { return g.cz[6 * g.nzc + k * g.nx + xx]; };
    auto rK_of = [&](int z, int x, float &rKx, float &rKxh, float &rKz, float &rKzh) {  // load_rK
        rKx = rKxh = rKz = rKzh = 1.0f;
        if (x < g.nPml || x > g.nx - g.nPml - 1) {
            rKx = CX(C_RK, x);
            rKxh = CX(C_RKH, x);
        }
        if (pml_z(z)) {
            rKz = CZ(C_RK, z);
            rKzh = CZ(C_RKH, z);
        }
    };
    const float *__restrict__ frame_t = g.frame_t;
    const int P = g.pitch, L = g.frame_len;
    const int zmax = g.nzc - 1 - g.nPml, xmax = g.nx - 1 - g.nPml;
    auto slot_of = [&](int z, int x) {  // frame_slot (device_common.hpp)
        const int zf = z - (g.nPml - 2), xf = x - (g.nPml - 2);
        if (zf < 0 || zf >= g.nzBnd || xf < 0 || xf >= g.nxBnd) return -1;
        if (zf < 5) return zf * g.nxBnd + xf;
        if (zf >= g.nzBnd - 5) return (5 + zf - (g.nzBnd - 5)) * g.nxBnd + xf;
        const int base = 10 * g.nxBnd + (zf - 5) * 10;
        if (xf < 5) return base + xf;
        if (xf >= g.nxBnd - 5) return base + 5 + (xf - (g.nxBnd - 5));
        return -1;
    };
    const int nb = (int)g.nb, rowb = P * 4;
    // the hardware's range check covers scalar offset + lane offset: one descriptor spans a whole bundle
    const rsrc_t RI = make_rsrc(g.set_in, (g.dbg & 16) ? 0u : 18u * g.nb), RO = make_rsrc(g.set_out, (g.dbg & 8) ? 0u : 18u * g.nb),
                 RM = make_rsrc(g.media, (g.dbg & 128) ? 0u : 6u * g.nb), RA = make_rsrc(g.acc, (g.dbg & 64) ? 0u : 5u * g.nb);

    // tile of this block (same XCD banding as my_cell)
    int t = blockIdx.x;
    {
        const int per = (g.ntile + 7) >> 3;
        t = (t & 7) * per + (t >> 3);
    }
    if (t >= g.ntile) return;  // surplus block of the remapped numbering (the whole block: nobody waits at the barrier for it)
    const int ty = t / g.gx, tx = t - ty * g.gx;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    const int x0 = tx * FUSED_NXO - 2, z0 = ty * NO - 2;  // grid position of region cell (0, 0)

    // ---------------------------------------------------------------------------------------------------------------
    // stage 1 for ONE region cell (row r, column c); z may be wave-uniform (row passes) or per-lane (halo pass).  Returns the
    // updated rho-imaging accumulators of the cell (stored after the barrier; no global store before it: a store in flight
    // would make every later wait on a load a wait on the store as well).  Loads are unconditional -- a buffer load
    // outside its bundle returns 0, cells that do not take part are masked by their flags; only the rare ones (C-PML,
    // frame) are predicated.
    // ---------------------------------------------------------------------------------------------------------------
    auto stage1 = [&](const int r, const int c, const bool want_acc, float &acc_a, float &acc_b) {
        const int z = z0 + r, x = x0 + c;
        int vrow[5];  // byte offsets of (z + dz, x - 2), dz = -2..2: a tap adds its column in the immediate, its array in the scalar offset
        vrow[0] = ((z - 2) * P + (x - 2)) * 4;
#pragma unroll
        for (int d = 1; d < 5; d++) vrow[d] = vrow[d - 1] + rowb;
        auto IN = [&](int arr, int dz, int dx) { return bld(RI, vrow[dz + 2] + 4 * (dx + 2), arr * nb); };
        auto MED = [&](int arr, int dz, int dx) { return bld(RM, vrow[dz + 2] + 4 * (dx + 2), arr * nb); };
        const int vo = vrow[2] + 8;  // the cell itself
        const bool in_grid = z >= 0 && z < g.nzc && x >= 0 && x < g.nx;
        const bool interior = (z >= g.nPml && z <= zmax && x >= g.nPml && x <= xmax);
        const bool on = (z >= 2 && z <= g.nzc - 3 && x >= 2 && x <= g.nx - 3);
        const bool px = (x < g.nPml || x > g.nx - g.nPml - 1);
        const int s = in_grid ? slot_of(z, x) : -1;
        // ---- loads: reverse-time velocity body
        const float szz_zm1 = IN(S_SZZ, -1, 0), szz_0 = IN(S_SZZ, 0, 0), szz_zp1 = IN(S_SZZ, 1, 0), szz_zp2 = IN(S_SZZ, 2, 0);
        const float sxz_xm2 = IN(S_SXZ, 0, -2), sxz_xm1 = IN(S_SXZ, 0, -1), sxz_0 = IN(S_SXZ, 0, 0), sxz_xp1 = IN(S_SXZ, 0, 1);
        const float sxz_zm2 = IN(S_SXZ, -2, 0), sxz_zm1 = IN(S_SXZ, -1, 0), sxz_zp1 = IN(S_SXZ, 1, 0);
        const float sxx_xm1 = IN(S_SXX, 0, -1), sxx_0 = IN(S_SXX, 0, 0), sxx_xp1 = IN(S_SXX, 0, 1), sxx_xp2 = IN(S_SXX, 0, 2);
        const float vz0 = IN(S_VZ, 0, 0), vx0 = IN(S_VX, 0, 0);
        // ---- loads: adjoint stress body
        const float avz_xm1 = IN(S_AVZ, 0, -1), avz0 = IN(S_AVZ, 0, 0), avz_xp1 = IN(S_AVZ, 0, 1), avz_xp2 = IN(S_AVZ, 0, 2);
        const float avz_zm2 = IN(S_AVZ, -2, 0), avz_zm1 = IN(S_AVZ, -1, 0), avz_zp1 = IN(S_AVZ, 1, 0);
        const float avx_zm1 = IN(S_AVX, -1, 0), avx0 = IN(S_AVX, 0, 0), avx_zp1 = IN(S_AVX, 1, 0), avx_zp2 = IN(S_AVX, 2, 0);
        const float avx_xm2 = IN(S_AVX, 0, -2), avx_xm1 = IN(S_AVX, 0, -1), avx_xp1 = IN(S_AVX, 0, 1);
        const float asxz0 = IN(S_ASXZ, 0, 0), asxx0 = IN(S_ASXX, 0, 0), aszz0 = IN(S_ASZZ, 0, 0);
        const float ba = MED(M_BYA, 0, 0), bb = MED(M_BYB, 0, 0);   // buoyancies() of the two-launch kernels (stored averages)
        float g_a = 0.f, g_b = 0.f;
        if (want_acc) {
            g_a = bld(RA, vo, A_A * nb);
            g_b = bld(RA, vo, A_B * nb);
        }
        // ---------------- reverse-time velocity + rho imaging + frame restore (velocity_body<false>) ----------------
        float vz_n = 0.f, vx_n = 0.f;
        {
            const float dszz_dz = dplus(szz_zm1, szz_0, szz_zp1, szz_zp2, g.rdz);
            const float dsxz_dx = dminus(sxz_xm2, sxz_xm1, sxz_0, sxz_xp1, g.rdx);
            const float dsxz_dz = dminus(sxz_zm2, sxz_zm1, sxz_0, sxz_zp1, g.rdz);
            const float dsxx_dx = dplus(sxx_xm1, sxx_0, sxx_xp1, sxx_xp2, g.rdx);
            if (interior) {
                vz_n = vz0 - (dszz_dz + dsxz_dx) * ba * g.dt;
                vx_n = vx0 - (dsxz_dz + dsxx_dx) * bb * g.dt;
            }
            acc_a = g_a + -avz0 * (dszz_dz + dsxz_dx) * g.dt;
            acc_b = g_b + -avx0 * (dsxz_dz + dsxx_dx) * g.dt;
        }
        if (s >= 0) {
            vz_n = frame_t[3 * L + s];
            vx_n = frame_t[4 * L + s];
        }
        lds[0][r][c] = vz_n;
        lds[1][r][c] = vx_n;
        // ---------------- adjoint stress of the previous step (stress_adj_load / stress_adj_apply) ----------------
        float sxz_a = 0.f, sxx_a = 0.f, szz_a = 0.f;
        {
            const bool pz = pml_z(z);
            float rKx = 1.0f, rKxh = 1.0f, rKz = 1.0f, rKzh = 1.0f;
            if (on) rK_of(z, x, rKx, rKxh, rKz, rKzh);
            const float dvz_dx = -dplus(avz_xm1, avz0, avz_xp1, avz_xp2, g.rdx);
            const float dvx_dz = -dplus(avx_zm1, avx0, avx_zp1, avx_zp2, g.rdz);
            float us = dvz_dx * rKx * ba * g.dt + dvx_dz * rKz * bb * g.dt;
            const float dvx_dx = -dminus(avx_xm2, avx_xm1, avx0, avx_xp1, g.rdx);
            const float dvz_dz = -dminus(avz_zm2, avz_zm1, avz0, avz_zp1, g.rdz);
            float ux = bb * dvx_dx * rKxh * g.dt;
            float uz = ba * dvz_dz * rKzh * g.dt;
            if (on && px) {
                us += CX(C_A, x) * -dplus(IN(S_DSXZ_DX, 0, -1), IN(S_DSXZ_DX, 0, 0), IN(S_DSXZ_DX, 0, 1), IN(S_DSXZ_DX, 0, 2), g.rdx);
                ux += CX(C_AH, x) * -dminus(IN(S_DSXX_DX, 0, -2), IN(S_DSXX_DX, 0, -1), IN(S_DSXX_DX, 0, 0), IN(S_DSXX_DX, 0, 1), g.rdx);
            }
            if (on && pz) {
                us += CZ(C_A, z) * -dplus(IN(S_DSXZ_DZ, -1, 0), IN(S_DSXZ_DZ, 0, 0), IN(S_DSXZ_DZ, 1, 0), IN(S_DSXZ_DZ, 2, 0), g.rdz);
                uz += CZ(C_AH, z) * -dminus(IN(S_DSZZ_DZ, -2, 0), IN(S_DSZZ_DZ, -1, 0), IN(S_DSZZ_DZ, 0, 0), IN(S_DSZZ_DZ, 1, 0), g.rdz);
            }
            if (on) {
                sxz_a = asxz0 + us;
                sxx_a = asxx0 + ux;
                szz_a = aszz0 + uz;
            }
        }
        lds[2][r][c] = szz_a;
        lds[3][r][c] = sxx_a;
        lds[4][r][c] = sxz_a;
    };

    // row passes: the wave's R1 region rows, 64 owned columns each, one after the other (not unrolled: one row's loads in
    // registers at a time)
    float acc_a[R1], acc_b[R1];
#pragma unroll
    for (int k = 0; k < R1; k++) {
        const int r = w * R1 + k;
        const bool row_own = r >= 2 && r <= NR - 3;  // wave-uniform
        stage1(r, lane + 2, row_own, acc_a[k], acc_b[k]);
    }
    // halo pass: the two columns left and right of the tile, for all rows of the wave at once
    if (lane < 4 * R1) {
        float da, db;
        const int q = lane & 3;
        stage1(w * R1 + (lane >> 2), q < 2 ? q : 64 + q, false, da, db);
    }

    // LDS hand-over only: global loads still in flight stay in flight across the barrier
    if (!(g.dbg & 1)) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    if (g.dbg & 2) return;

    // ---- rho-imaging accumulators of the rows this wave handled in stage 1
    {
        const int x = x0 + 2 + lane;
#pragma unroll
        for (int k = 0; k < R1; k++) {
            const int r = w * R1 + k, z = z0 + r;
            if (r >= 2 && r <= NR - 3 && z >= g.nPml && z <= zmax && x >= g.nPml && x <= xmax) {
                const int vo = (z * P + x) * 4;
                bst(RA, acc_a[k], vo, A_A * nb);
                bst(RA, acc_b[k], vo, A_B * nb);
            }
        }
    }

    // ---------------------------------------------------------------------------------------------------------------
    // stage 2: the wave's R2 owned rows
    // ---------------------------------------------------------------------------------------------------------------
#pragma unroll
    for (int k = 0; k < R2; k++) {
        const int r = 2 + w * R2 + k, c = lane + 2;
        const int z = z0 + r, x = x0 + c;
        if (z >= g.nzc || x >= g.nx) continue;  // z, x >= 0 for owned cells
        const int vo = (z * P + x) * 4;
        auto IN = [&](int arr, int dz, int dx) { return bld(RI, vo + dz * rowb + 4 * dx, arr * nb); };
        auto OUT = [&](int arr, float v) { bst(RO, v, vo, arr * nb); };
        auto MED = [&](int arr, int dz, int dx) { return bld(RM, vo + dz * rowb + 4 * dx, arr * nb); };
        auto VZ = [&](int dr, int dl) { return lds[0][r + dr][c + dl]; };
        auto VX = [&](int dr, int dl) { return lds[1][r + dr][c + dl]; };
        auto AZZ = [&](int dr, int dl) { return lds[2][r + dr][c + dl]; };
        auto AXX = [&](int dr, int dl) { return lds[3][r + dr][c + dl]; };
        auto AXZ = [&](int dr, int dl) { return lds[4][r + dr][c + dl]; };
        const bool interior = (z >= g.nPml && z <= zmax && x >= g.nPml && x <= xmax);
        const bool on = (z >= 2 && z <= g.nzc - 3 && x >= 2 && x <= g.nx - 3);
        const bool pz = pml_z(z), px = (x < g.nPml || x > g.nx - g.nPml - 1);
        const bool wz = (z < g.nPml + 2 || z > g.nzc - g.nPml - 3), wx = (x < g.nPml + 2 || x > g.nx - g.nPml - 3);
        const int s = slot_of(z, x);
        // global loads of the row (the own-cell field values were this block's stage-1 taps a moment ago: L2)
        const float szz_o = IN(S_SZZ, 0, 0), sxx_o = IN(S_SXX, 0, 0), sxz_o = IN(S_SXZ, 0, 0);
        const float avz_o = IN(S_AVZ, 0, 0), avx_o = IN(S_AVX, 0, 0);
        const float lam = MED(M_LAM, 0, 0), mu = MED(M_MU, 0, 0), amu = MED(M_AMU, 0, 0);
        const float g_lam = bld(RA, vo, A_LAM * nb), g_mu = bld(RA, vo, A_MU * nb), g_xz = bld(RA, vo, A_XZ * nb);
        const float za = AZZ(0, 0), xa = AXX(0, 0), sa = AXZ(0, 0);  // NEW adjoint stresses of the own cell

        // source_grad (utilities.cu:719-730): adjoint stresses after the adjoint stress update of the previous step
        if (z == z_src && x == x_src) *g.stf_grad_it = -(za + g.src_rxz * xa) * g.dt;

        // ---------------- stage-1 results of the owned cell to memory ----------------
        if (interior || s >= 0) {
            OUT(S_VZ, VZ(0, 0));
            OUT(S_VX, VX(0, 0));
        }
        if (on) {
            OUT(S_ASXZ, sa);
            OUT(S_ASXX, xa);
            OUT(S_ASZZ, za);
            if (wx || wz) {  // stress-type memory variables (stress_adj_apply), strips widened by the stencil radius
                const float l2m = lam + 2.0f * mu;
                if (wx) {
                    OUT(S_DVZ_DX, CX(C_BH, x) * IN(S_DVZ_DX, 0, 0) + sa * amu * g.dt);
                    OUT(S_DVX_DX, CX(C_B, x) * IN(S_DVX_DX, 0, 0) + lam * za * g.dt + l2m * xa * g.dt);
                }
                if (wz) {
                    OUT(S_DVX_DZ, CZ(C_BH, z) * IN(S_DVX_DZ, 0, 0) + sa * amu * g.dt);
                    OUT(S_DVZ_DZ, CZ(C_B, z) * IN(S_DVZ_DZ, 0, 0) + l2m * za * g.dt + lam * xa * g.dt);
                }
            }
        }

        // ---------------- reverse-time stress + lambda/mu imaging + frame restore (stress_body<false>) ----------------
        if (interior || s >= 0) {
            float szz = 0.f, sxx = 0.f, sxz = 0.f;
            if (interior) {
                szz = szz_o;
                sxx = sxx_o;
                sxz = sxz_o;
                if (z == z_src && x == x_src) {
                    szz -= g.src_amp;
                    sxx -= g.src_amp;
                }
                const float dvz_dz = dminus(VZ(-2, 0), VZ(-1, 0), VZ(0, 0), VZ(1, 0), g.rdz);
                const float dvx_dx = dminus(VX(0, -2), VX(0, -1), VX(0, 0), VX(0, 1), g.rdx);
                const float dvx_dz = dplus(VX(-1, 0), VX(0, 0), VX(1, 0), VX(2, 0), g.rdz);
                const float dvz_dx = dplus(VZ(0, -1), VZ(0, 0), VZ(0, 1), VZ(0, 2), g.rdx);
                const float l2m = lam + 2.0f * mu;
                szz -= (l2m * dvz_dz + lam * dvx_dx) * g.dt;
                sxx -= (lam * dvz_dz + l2m * dvx_dx) * g.dt;
                sxz -= amu * (dvx_dz + dvz_dx) * g.dt;
                bst(RA, g_lam + -(za + xa) * (dvz_dz + dvx_dx) * g.dt, vo, A_LAM * nb);
                bst(RA, g_mu + -2.0f * (za * dvz_dz + xa * dvx_dx) * g.dt, vo, A_MU * nb);
                bst(RA, g_xz + -sa * (dvx_dz + dvz_dx) * g.dt, vo, A_XZ * nb);
            }
            if (s >= 0) {
                szz = frame_t[s];
                sxz = frame_t[L + s];
                sxx = frame_t[2 * L + s];
            }
            OUT(S_SZZ, szz);
            OUT(S_SXX, sxx);
            OUT(S_SXZ, sxz);
        }

        // ---------------- adjoint velocity + line injection (velocity_adj_load / velocity_adj_apply) ----------------
        if (on) {
            float rKx, rKxh, rKz, rKzh;
            rK_of(z, x, rKx, rKxh, rKz, rKzh);
            const float l2m = lam + 2.0f * mu;
            const float dszz_dx = -dplus(AZZ(0, -1), za, AZZ(0, 1), AZZ(0, 2), g.rdx);
            const float dsxx_dx = -dplus(AXX(0, -1), xa, AXX(0, 1), AXX(0, 2), g.rdx);
            const float dsxz_dz = -dminus(AXZ(-2, 0), AXZ(-1, 0), sa, AXZ(1, 0), g.rdz);
            float upd = lam * dszz_dx * rKx * g.dt + l2m * dsxx_dx * rKx * g.dt + amu * rKzh * dsxz_dz * g.dt;
            const float dszz_dz = -dplus(AZZ(-1, 0), za, AZZ(1, 0), AZZ(2, 0), g.rdz);
            const float dsxx_dz = -dplus(AXX(-1, 0), xa, AXX(1, 0), AXX(2, 0), g.rdz);
            const float dsxz_dx = -dminus(AXZ(0, -2), AXZ(0, -1), sa, AXZ(0, 1), g.rdx);
            float upz = l2m * dszz_dz * rKz * g.dt + lam * dsxx_dz * rKz * g.dt + amu * rKxh * dsxz_dx * g.dt;
            if (px) {
                // psi_v as its owner has just updated it (stress_adj_apply, strips widened by 2): rebuilt at the tap
                auto new_dvx_dx = [&](int d) {  // tap (z, x + d)
                    const float old = IN(S_DVX_DX, 0, d);
                    if (x + d < 2 || x + d > g.nx - 3) return old;  // never updated there
                    const float l = MED(M_LAM, 0, d), m2 = l + 2.0f * MED(M_MU, 0, d);
                    return CX(C_B, x + d) * old + l * AZZ(0, d) * g.dt + m2 * AXX(0, d) * g.dt;
                };
                auto new_dvz_dx = [&](int d) {
                    const float old = IN(S_DVZ_DX, 0, d);
                    if (x + d < 2 || x + d > g.nx - 3) return old;
                    return CX(C_BH, x + d) * old + AXZ(0, d) * MED(M_AMU, 0, d) * g.dt;
                };
                upd += CX(C_A, x) * -dplus(new_dvx_dx(-1), new_dvx_dx(0), new_dvx_dx(1), new_dvx_dx(2), g.rdx);
                upz += CX(C_AH, x) * -dminus(new_dvz_dx(-2), new_dvz_dx(-1), new_dvz_dx(0), new_dvz_dx(1), g.rdx);
            }
            if (pz) {
                auto new_dvx_dz = [&](int d) {  // tap (z + d, x)
                    const float old = IN(S_DVX_DZ, d, 0);
                    if (z + d < 2 || z + d > g.nzc - 3) return old;
                    return CZ(C_BH, z + d) * old + AXZ(d, 0) * MED(M_AMU, d, 0) * g.dt;
                };
                auto new_dvz_dz = [&](int d) {
                    const float old = IN(S_DVZ_DZ, d, 0);
                    if (z + d < 2 || z + d > g.nzc - 3) return old;
                    const float l = MED(M_LAM, d, 0), m2 = l + 2.0f * MED(M_MU, d, 0);
                    return CZ(C_B, z + d) * old + m2 * AZZ(d, 0) * g.dt + l * AXX(d, 0) * g.dt;
                };
                upd += CZ(C_AH, z) * -dminus(new_dvx_dz(-2), new_dvx_dz(-1), new_dvx_dz(0), new_dvx_dz(1), g.rdz);
                upz += CZ(C_A, z) * -dplus(new_dvz_dz(-1), new_dvz_dz(0), new_dvz_dz(1), new_dvz_dz(2), g.rdz);
            }
            const float vx = avx_o + upd;
            const float vz = avz_o + upz;
            float vs = vx;
            if (lr.n && z == lr.z) {  // res_injection_exx, utilities.cu:605-615
                const int rr = x - lr.x0;
                if (rr >= 0 && rr < lr.n) vs += lr.res[rr];
                if (rr + 1 >= 0 && rr + 1 < lr.n) vs -= lr.res[rr + 1];
            }
            OUT(S_AVX, vs);
            OUT(S_AVZ, vz);
            if (px || pz) {
                const float bb = MED(M_BYB, 0, 0), ba = MED(M_BYA, 0, 0);
                if (px) {
                    OUT(S_DSXX_DX, CX(C_BH, x) * IN(S_DSXX_DX, 0, 0) + bb * vx * g.dt);
                    OUT(S_DSXZ_DX, CX(C_B, x) * IN(S_DSXZ_DX, 0, 0) + ba * vz * g.dt);
                }
                if (pz) {
                    OUT(S_DSXZ_DZ, CZ(C_B, z) * IN(S_DSXZ_DZ, 0, 0) + bb * vx * g.dt);
                    OUT(S_DSZZ_DZ, CZ(C_BH, z) * IN(S_DSZZ_DZ, 0, 0) + ba * vz * g.dt);
                }
            }
        }
    }
}
