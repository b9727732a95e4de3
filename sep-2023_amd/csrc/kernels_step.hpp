// kernels_step.hpp -- the per-step kernels: one body each, the fused backward pairs, the batched forms
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

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
