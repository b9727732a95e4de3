// kernels_persist.hpp -- k_bwd_persist, the persistent backward time loop
// Part of the ONE translation unit kernels.hip (included there, inside namespace sepfwi): the kernels share their bodies as
// inline functions, and every kernel structure must compile them identically (bit-identical results).

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
// Instances: k_bwd_persist<LMASK> (a fused line of channels, or none) and <LMASK, GINJ> (any other receiver geometry) are what the
// shipped library launches.  <.., MS> (several shots per launch), <.., QS> (quiet row segments) and the pk_lock item order are
// experiments that are bit-identical (MS, QS) and lost on time (profiles/EXPERIMENTS.md #47-#49): they are compiled in the
// -DSEPFWI_PROBES build only, where their tests and A/B scripts run.
// ---------------------------------------------------------------------------------------------
#ifdef SEPFWI_PK_TRACE
// one-off wave timeline of the loop (build with SEPFWI_HIPCC_FLAGS=-DSEPFWI_PK_TRACE; scripts/gpu_r05_pk_trace.sh, scripts/pk_trace.py):
// per (tile, wave, phase 200..207 of the launch) s_memrealtime at the phase start (after the barrier), at each item start (up to 8), at
// the end of the wave's items and after the closing drain.  Kept in device memory, dumped to $SEPFWI_PK_TRACE at exit.
__device__ unsigned long long *g_pk_trace;
constexpr int kTrTiles = 512, kTrPh0 = 200, kTrPh = 8, kTrSlots = 12;
#endif
#ifdef SEPFWI_PK_FAULT
// Fault-injection build (tests/test_gpu_parity.py::test_loop_failure_path_reports_and_recovers; libsepfwi_fault.so): tile number
// SEPFWI_PK_FAULT stops publishing its phases after the 40th, so that its neighbours run into the time limit inside a pass --
// the one path of the loop that a healthy run never takes.  The limit is shortened to keep the test short.
constexpr int kPersistSpinLimit = 1 << 14;
#else
constexpr int kPersistSpinLimit = 1 << 21;   // polls of ~1 us: about two seconds (never reached once the start rendezvous has passed)
#endif
constexpr int kPersistStartLimit = 1 << 15;  // start rendezvous: ~30 ms
constexpr unsigned int kFlagPhase = 0x0fffffffu;  // a tile's flag word: phases completed (low 28 bits) | QS: summary of its edge segments (top 4)

// registers sized for 8 waves per SIMD: two workgroups of 16 waves per CU
// GINJ: the shot's receivers are not a fused horizontal line (scattered or strided channels, a vertical fibre, directional
// sensitivities): the residual of the step, folded per target cell beforehand (k_inject_values), is added by the lane that owns the
// cell right after its adjoint-velocity update -- res_injection_exx / _ezz (Src/utilities.cu:605-641) without a launch of its own.
// MS: ONE launch carries several shots through their backward passes (grids far below the headline's size, where a single shot cannot
// feed 512 tiles: the reference's own experiments are 101 x 201 cells x 19 shots).  The tiles cut a VIRTUAL grid -- the shots of the
// batch stacked row-wise -- so a row segment's descriptor also names its shot (bits 26..31); the shots' arrays lie at constant
// strides (PersistArgs::ms), their scalars (source, line of channels) in the ShotDev table of the batched launches.  Everything else
// -- phases, flags, LDS accumulators, the bodies -- is the single-shot loop's.
// QS (option quiet_skip inside the loop): updates of row segments whose every input is exactly +0 are left out, bit for
// bit the same arrays.  The tile keeps one word per row segment in LDS -- bit g set: group g of the segment may hold a non-zero value
// (g = 0 forward velocities, 1 forward stresses, 2 adjoint velocities, 3 adjoint stresses; the forward bits start from the forward
// pass' maps, the adjoint bits from zero; bits are only ever set) -- and decides per item from its own word and the words of the six
// row segments its stencils reach (host-built table of their positions in the tile: PersistArgs::q).  Of a segment in ANOTHER tile
// only a summary is known: the OR over that tile's edge segments, which travels in the top four bits of the tile's phase flag -- the
// poll that orders the phases reads it anyway -- so next to a tile that holds values the edge segments are computed (never skipped on
// a guess: a skipped update is one whose inputs are provably +0).  The bits an update reads are written in the OTHER phase only.
template <int LMASK, bool GINJ = false, bool MS = false, bool QS = false>
__global__ __launch_bounds__(MAXT, 8) void k_bwd_persist(Grid g, const PersistArgs a) {
    extern __shared__ float lds_dyn[];
    __shared__ int next_item, edge_done, abort_flag, start_verdict, cu_slot_s;
    __shared__ unsigned int nb_sum_s, edge_sum_s;  // QS: OR of the neighbours' summaries (this phase); OR over this tile's edge segments
    const ShotDev &s = a.s;
    const size_t n = a.n;
#ifdef SEPFWI_PROBES
    const bool nosync = a.nosync != 0;  // timing experiments (WRONG results): only a library built with -DSEPFWI_PROBES has the switch
#else
    constexpr bool nosync = false;
#endif
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
        if (nosync && blockIdx.x == 0) atomicExch(a.band_xcc + 9, kPersistGo);  // (the host reads the decision word)
        if (!nosync) {
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
    bool inj_tile = false;  // GINJ: does this tile own cells of the adjoint source?
    if constexpr (GINJ) inj_tile = __builtin_amdgcn_readfirstlane((int)a.injp->tile_has[tile]) != 0;
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
    // (MS: the accumulators of the segment's own shot)
    auto acc_arrays = [&](uint32_t d) {
        if constexpr (MS) return acc_of(s.acc + (size_t)(d >> 26) * a.ms.bwd_stride, n);
        else return acc.p;
    };
    if constexpr (LMASK != 0) {
        for (int j = wave; j < nst; j += nw) {
            const uint32_t d = segs[j];
            const Cell c = cell_of(d);
            const ImgAcc ap = acc_arrays(d);
            lds_float *cell = lbase + j * BX + lane;
            int r = 0;
            if constexpr (LMASK & 1) cell[(r++) * acc.stride] = ap.lam[c.i];
            if constexpr (LMASK & 2) cell[(r++) * acc.stride] = ap.mu[c.i];
            if constexpr (LMASK & 4) cell[(r++) * acc.stride] = ap.xz[c.i];
            if constexpr (LMASK & 8) cell[(r++) * acc.stride] = ap.a[c.i];
            if constexpr (LMASK & 16) cell[(r++) * acc.stride] = ap.b[c.i];
        }
    }
    unsigned int *const qs = (unsigned int *)(lds_dyn + (size_t)__builtin_popcount(LMASK) * (size_t)acc.stride);  // QS: one word per row segment
    if constexpr (QS) {
        if (threadIdx.x == 0) {
            nb_sum_s = 0u;
            edge_sum_s = 0u;
        }
        __syncthreads();
        for (int j = (int)threadIdx.x; j < nst; j += (int)blockDim.x) {
            const uint32_t d = segs[j];
            const int z = (int)(d & 0xffffu), xs = (int)((d >> 16) & 0xffu), r = z + 2;
            unsigned int st0 = 0u;
            if (z >= 2 && z <= g.nzc - 3) {  // the forward pass' maps (Fields::q: column xs + 1 of qzw words, bit z + 2)
                const size_t wi = (size_t)(xs + 1) * (size_t)g.qzw + (size_t)(r >> 5);
                st0 = ((a.q.maps[wi] >> (r & 31)) & 1u) | (((a.q.maps[(size_t)g.qn + wi] >> (r & 31)) & 1u) << 1);
            }
            qs[j] = st0;
            if (j < n_edge && st0) atomicOr(&edge_sum_s, st0);
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
        // one work item: phase ph of the time step on row segment j of the tile
        auto run_item = [&](int ph, int j, bool sync) {
            const uint32_t d = segs[j];
            const Cell c = cell_of(d);
            acc.cell = lbase + j * BX + lane;
            const bool xband = (d & kSegXband) != 0 && sync;  // wave-uniform
            if constexpr (MS) {
                // the segment's shot: its arrays at constant strides from the first shot's, its scalars from the table (scalar loads)
                typedef const ShotDev __attribute__((address_space(4))) *shots_t;
                const int sh = (int)(d >> 26);
                const shots_t q = (shots_t)a.ms.shots + sh;
                const size_t so = (size_t)sh * a.ms.state_stride, bo = (size_t)sh * a.ms.bwd_stride;
                const Fields f1 = fields_of(s.fields + so, n), adj1 = fields_of(s.adj + bo, n);
                const PmlMem m1 = mem_of(s.bmem + bo, n);
                AccT<LMASK> acc1{acc_of(s.acc + bo, n), acc.cell, acc.stride};
                float *frame1 = s.frame + (size_t)sh * a.ms.frame_stride + (size_t)it * 5 * (size_t)g.frame_len;
                if (ph == 0) {
                    if (xband) {
                        velocity_body<false, AccT<LMASK>, MemAgent>(gs, c, f1, m1, md, pc, frame1, -1, -1, 0.0f, nullptr, adj1, acc1);
                        stress_adj_body<MemAgent>(gs, c, adj1, m1, md, pc);
                    } else {
                        velocity_body<false>(gs, c, f1, m1, md, pc, frame1, -1, -1, 0.0f, nullptr, adj1, acc1);
                        stress_adj_body(gs, c, adj1, m1, md, pc);
                    }
                } else {
                    const int z_src = q->z_src, x_src = q->x_src, nrec1 = q->nrec;
                    const float amp1 = __fmul_rn(__fmul_rn(a.src_scale, s.stf[(size_t)sh * (size_t)g.nSteps + it]), g.dt);
                    const LineRec lr1{q->lr_z, q->lr_x0, q->lr_n, nullptr, nullptr, nullptr, s.res + (size_t)sh * a.ms.res_stride + (size_t)it * (size_t)nrec1};
                    if (c.z == z_src && c.x == x_src)
                        s.stf_grad[(size_t)sh * (size_t)g.nSteps + it] = -(adj1.szz[c.i] + q->src_rxz * adj1.sxx[c.i]) * g.dt;  // source_grad
                    if (xband) {
                        stress_body<false, false, AccT<LMASK>, MemAgent>(gs, c, f1, m1, md, pc, frame1, z_src, x_src, amp1, adj1, acc1, LineRec{});
                        velocity_adj_body<MemAgent>(gs, c, adj1, m1, md, pc, lr1);
                    } else {
                        stress_body<false, false>(gs, c, f1, m1, md, pc, frame1, z_src, x_src, amp1, adj1, acc1, LineRec{});
                        velocity_adj_body(gs, c, adj1, m1, md, pc, lr1);
                    }
                }
                return;
            }
            if constexpr (QS) {
                // decide for both updates of the item from the tile's words (wave-uniform), apply, mark
                const int z = c.z, xs = (int)((d >> 16) & 0xffu);
                const bool on = z >= 2 && z <= g.nzc - 3;
                typedef const unsigned long long __attribute__((address_space(4))) *nbt_t;
                const unsigned long long nbw = ((nbt_t)a.q.nbr)[(size_t)tile * (size_t)a.cap + (size_t)j];
                unsigned int v = 0u;
                if (lane < 7) {
                    const unsigned int idx = lane == 6 ? (unsigned int)j : (unsigned int)((nbw >> (8 * lane)) & 0xffull);
                    v = idx == 0xffu ? 0u : idx == 0xfeu ? nb_sum_s : qs[idx];  // 0xff: no such row segment; 0xfe: another tile's
                }
                unsigned int own = 0u, any = 0u;
#pragma unroll
                for (int b = 0; b < 4; b++) {
                    const unsigned long long bal = __ballot((v >> b) & 1u);
                    own |= ((bal >> 6) & 1ull) ? (1u << b) : 0u;
                    any |= (bal & 0x7full) ? (1u << b) : 0u;
                }
                bool q1 = false, q2 = false, ni = false;
                int g1, g2;
                bool nz1, nz2;
                if (ph == 0) {  // reverse-time velocity: writes group 0, reads 1, images with 2; adjoint stress: writes 3, reads 2
                    g1 = 0;
                    g2 = 3;
                    if (on) {
                        q1 = !((own & 1u) | (any & 2u));
                        ni = !(own & 4u);
                        q2 = !((own & 8u) | (any & 4u));
                    }
                    if (xband) {
                        nz1 = velocity_body<false, AccT<LMASK>, MemAgent>(gs, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, acc, q1, ni);
                        nz2 = stress_adj_body<MemAgent>(gs, c, adj, m, md, pc, q2);
                    } else {
                        nz1 = velocity_body<false>(gs, c, f, m, md, pc, frame_t, -1, -1, 0.0f, nullptr, adj, acc, q1, ni);
                        nz2 = stress_adj_body(gs, c, adj, m, md, pc, q2);
                    }
                } else {  // reverse-time stress: writes 1, reads 0, images with 3, the source enters; adjoint velocity: writes 2, reads 3, the residual enters
                    g1 = 1;
                    g2 = 2;
                    if (c.z == s.z_src && c.x == s.x_src) s.stf_grad[it] = -(adj.szz[c.i] + s.src_rxz * adj.sxx[c.i]) * g.dt;  // source_grad
                    if (on) {
                        const bool src = z == s.z_src && (s.x_src >> 6) == xs && amp != 0.0f;
                        const bool rec = lr.n && z == lr.z && xs * BX + BX - 1 >= lr.x0 - 1 && xs * BX <= lr.x0 + lr.n - 1;  // cells lr.x0 - 1 ... lr.x0 + lr.n - 1
                        q1 = !((own & 2u) | (any & 1u) | (src ? 1u : 0u));
                        ni = !(own & 8u);
                        q2 = !((own & 4u) | (any & 8u) | (rec ? 1u : 0u));
                    }
                    if (xband) {
                        nz1 = stress_body<false, false, AccT<LMASK>, MemAgent>(gs, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, acc, LineRec{}, q1, ni);
                        nz2 = velocity_adj_body<MemAgent>(gs, c, adj, m, md, pc, lr, q2);
                    } else {
                        nz1 = stress_body<false, false>(gs, c, f, m, md, pc, frame_t, s.z_src, s.x_src, amp, adj, acc, LineRec{}, q1, ni);
                        nz2 = velocity_adj_body(gs, c, adj, m, md, pc, lr, q2);
                    }
                }
                if (on) {  // a stored non-zero value sets the segment's bit (and the tile's summary, for an edge segment)
                    unsigned int set = 0u;
                    if (!((own >> g1) & 1u) && __ballot(nz1) != 0ull) set |= 1u << g1;
                    if (!((own >> g2) & 1u) && __ballot(nz2) != 0ull) set |= 1u << g2;
                    if (set && lane == 0) {
                        atomicOr(&qs[j], set);
                        if (j < n_edge) atomicOr(&edge_sum_s, set);
                    }
                }
                return;
            }
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
                if constexpr (GINJ) {
                    // the step's adjoint source at the cells of this segment (rows and columns outside the updated region included:
                    // the adjoint stress update reads them through its stencils).  Only tiles that own target cells look anything up,
                    // and the tables are reached through ONE pointer (a.injp), read in this rare branch: six more pointers alive in
                    // the loop cost every row segment of every tile scalar spills.
                    // (Folding the value into the body's own store instead of this read-modify-write was built twice and is slower:
                    // 57.4 us per backward step with a second inlined copy of the adjoint-velocity body, 56.4 with one copy and a
                    // per-lane select -- 44 B of scratch per lane at the 64-register budget -- against 54.2; EXPERIMENTS #50.)
                    if (inj_tile) {
                        typedef const InjArgs __attribute__((address_space(4))) *inj_t;
                        typedef const int __attribute__((address_space(4))) *ctab_t;
                        const inj_t ia = (inj_t)a.injp;
                        const int slot = ((ctab_t)ia->lookup)[c.z * ia->nseg + (int)((d >> 16) & 0xffu)];
                        if (slot >= 0 && c.x < g.nx) {
                            const InjSeg q = ia->segs[slot];
                            const float *val_t = ia->val + (size_t)it * (size_t)ia->ntgt;
                            const unsigned long long below = (1ull << lane) - 1ull;
                            if ((q.mask[0] >> lane) & 1ull) {
                                const float v = val_t[q.base[0] + __popcll(q.mask[0] & below)];
                                if (xband) MemAgent::st(&adj.vx[c.i], MemAgent::ld(&adj.vx[c.i]) + v);
                                else adj.vx[c.i] += v;
                            }
                            if ((q.mask[1] >> lane) & 1ull) {
                                const float v = val_t[q.base[1] + __popcll(q.mask[1] & below)];
                                if (xband) MemAgent::st(&adj.vz[c.i], MemAgent::ld(&adj.vz[c.i]) + v);
                                else adj.vz[c.i] += v;
                            }
                        }
                    }
                }
            }
        };
#ifdef SEPFWI_PROBES
        if (a.lock > 0) {
            // Timing probe (WRONG results: no synchronisation at all; profiles/EXPERIMENTS.md #47): the two phases of the time step
            // interleaved along the tile's walk order -- item 2k is phase A of segment k, item 2k + 1 phase B of segment k - D -- so that
            // what phase A stored is read by phase B while it is still in the XCD's L2 (the ceiling of a dataflow form in which a tile's
            // phase B trails its phase A by the stencil's reach instead of a whole pass over the tile).
            const int D = a.lock & 0xff, span = 2 * (nst + D), base = (local >> 1) * span;
            const bool rev = (a.lock & 0x100) != 0 && ((local >> 1) & 1) != 0;  // every other time step walks the tile backwards
            for (; w < base + span; w = grab()) {
                const int mm = w - base, ph = mm & 1;
                int j = (mm >> 1) - (ph ? D : 0);
                if (j < 0 || j >= nst) continue;
                if (rev) j = nst - 1 - j;
                run_item(ph, j, false);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (!(a.lock & 0x200)) __syncthreads();
            local += 2;
            continue;
        }
#endif
        for (int ph = 0; ph < 2 && !dead; ph++, local++) {
            const unsigned int phase = (unsigned int)(a.phase0 + local);
            // ---- neighbours through the edge part of the previous phase?  then drop what this CU's L1 still holds of their rows
            if (wave == 0 && !nosync) {
                bool ok = true;
                unsigned int seen = 0u;
                if (phase > 0 && lane < nnb) {
                    const unsigned int *pf = a.flags + (size_t)h.nb[lane] * 32;
                    int spins = 0;
                    while (((seen = __hip_atomic_load(pf, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) & kFlagPhase) < phase) {
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
                if constexpr (QS) {  // what the neighbours' edge segments may hold (bits only grow: a summary from a later phase is a superset)
                    const unsigned int sm = seen >> 28;
                    const unsigned int u = (__ballot(sm & 1u) ? 1u : 0u) | (__ballot(sm & 2u) ? 2u : 0u) | (__ballot(sm & 4u) ? 4u : 0u) | (__ballot(sm & 8u) ? 8u : 0u);
                    if (lane == 0) nb_sum_s = u;
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
                bool publish = old + 1 == nw * (local + 1) && lane == 0 && !nosync;
#ifdef SEPFWI_PK_FAULT
                if (tile == (SEPFWI_PK_FAULT) && phase >= 40u) publish = false;
#endif
                if (publish) {
                    unsigned int word = phase + 1u;
                    if constexpr (QS) word |= edge_sum_s << 28;  // (every edge item's marks precede its wave's count in `edge_done`: LDS operations are in order)
                    __hip_atomic_store(my_flag, word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                reported = true;
            };
            for (; w < base + nst; w = grab()) {
                const int j = w - base;
#ifdef SEPFWI_PK_TRACE
                if (tr && lane == 0 && tr_k < 9) tr[tr_k++] = __builtin_amdgcn_s_memrealtime();
#endif
                if (j >= n_edge && !reported) report();
                run_item(ph, j, !nosync);
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
            const uint32_t d = segs[j];
            const Cell c = cell_of(d);
            const ImgAcc ap = acc_arrays(d);
            lds_float *cell = lbase + j * BX + lane;
            int r = 0;
            if constexpr (LMASK & 1) ap.lam[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 2) ap.mu[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 4) ap.xz[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 8) ap.a[c.i] = cell[(r++) * acc.stride];
            if constexpr (LMASK & 16) ap.b[c.i] = cell[(r++) * acc.stride];
        }
    }
}
