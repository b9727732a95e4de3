// kernels_quiet.hpp -- quiet-segment maps (option quiet_skip): decide / apply helpers around the bodies
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

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
