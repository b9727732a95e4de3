// kernels.hpp -- host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "fwi_types.hpp"

namespace sepfwi {

// Kernel / scheduling options.  sepfwi_set_option() edits the process-wide defaults under a mutex; every cufd call works
// on ONE snapshot taken at its start (Session::run), which the launchers receive by reference.
struct KernelOptions {
    int bz = 2;           // waves (rows) per block of the field kernels
    int xcd_remap = 1;    // 1: each XCD gets a contiguous band of tiles
    int bwd_fuse = 4;     // backward step: 0 the reference's four kernels (+ k_inject), 2 cross-chain pairs k_bwd_a / k_bwd_b, 4 the
                          // persistent time loop for every receiver geometry where the grid feeds its tiles (else, and with quiet_skip, 2)
    int line_fuse = 1;    // 1: line receivers are sampled / injected inside the field kernels
    int pair_fwd = 1;     // 1: forward passes of several shots run concurrently (one stream each, or one batched launch)
    int fwd_lanes = 3;    // how many (1..4): 3 x 5 fields + 5 media arrays still sit in the Infinity Cache; 4 lanes lose
    int early = 0;        // fused backward kernels issue the loads of their second update first: bit 0 k_bwd_a, bit 1 k_bwd_b
    int rho_fly = 1;      // buoyancy averages rebuilt from the density: bit 0 forward velocity kernel, bit 1 backward kernels
    int amu_fly = 1;      // harmonic mean of mu rebuilt from mu: bit 0 forward stress kernel (+3.8 %), bit 1 backward kernels (-1 %)
    int rk_lazy = 1;      // adjoint kernels load 1/K only inside the C-PML layers (it is exactly 1 elsewhere)
    int batch = 2;        // shots of a call advance in batched launches: 0 never (one stream per forward lane), 1 always,
                          // 2 when at least three backward passes fit the cache budget together (two with the two-launch backward step)
    int batch_f = 0, batch_b = 0;  // explicit forward / backward batch sizes (0: from batch_mb)
    int batch_mb = 200;   // Infinity-Cache budget [MB] that sizes a batch: (5 B + 5) arrays forward, (15 B + 5) backward
    int batch_order = 1;  // batched launches: 0 shot-major block order, 1 the shots of one tile back to back (L2 reuse of the media)
    int batch_split = 2;  // batched launches: a batch as this many sub-batches on streams of their own (1..3): launches of different queues
                          // overlap their fill and drain (+5 % at 1000x500, +16 % on a 100x200 notebook-sized problem with 19 shots)
    int probe = 0;        // >0: time every probe-th k_bwd_b launch with HIP events (bench.py roofline)
    int quiet_skip = 0;   // 1: updates of row segments whose every input is exactly +0 are left out (the fields ahead of the wave front; same bits).
                          // Used by the forward kernels and the two-launch backward step (which then replaces the loop: EXPERIMENTS #49);
                          // line receivers (or none) only
    int quiet_rows = 4;   //   rows per wave of the quiet-skipping FORWARD kernels (a quiet wave costs its dispatch whatever it skips)
    int obs_cache_mb = 0; // HBM budget [MB] of the observed-data store (0: unlimited; a parameter-file key of the same name wins)
    // persistent backward time loop (bwd_fuse = 4; kernels.hip k_bwd_persist)
    int pk_lmask = 16;    //   imaging accumulators kept in LDS: 1 lam, 3 + mu, 7 + xz, 15 + a, 31 + b (16: as many as fit, in that order)
    int pk_wpc = 2;       //   workgroups (16 waves each) per CU
    int pk_px = 3;        //   strip width of the tiling in row segments (persist_plan.hpp)
    int pk_waves = 16;    //   waves per workgroup
    int pk_order = 2;     //   1: edge segments first in every phase (the flag goes out early), 0: strip order, the flag goes out at the end of the phase;
                          //   2: by tile size -- strip order from 56 row segments per tile on (its locality wins: -2.5 % at the headline's 69, -3.6 % at 100),
                          //   edge-first below (the early flag wins: +2 % for strip order at 36), profiles/EXPERIMENTS.md #52
    int pk_nosync = 0;    //   (-DSEPFWI_PROBES builds only) 1: no synchronisation between tiles -- WRONG RESULTS, timing experiments
    int pk_lock = 0;      //   (-DSEPFWI_PROBES builds only) > 0: the two phases of a time step interleaved, timing only (kernels_persist.hpp)
    int pk_snake = 1;     //   (-DSEPFWI_PROBES builds only) 0: every strip of the tiling is walked top-down
    int pk_ms = 0;        //   (-DSEPFWI_PROBES builds only) 1: a backward sub-batch of the batched schedule as ONE multi-shot persistent launch
    int pk_quiet = 0;     //   (-DSEPFWI_PROBES builds only) 1: with quiet_skip on, the loop's quiet-segment variant instead of the two-launch step
    int pk_prio = 1;      //   1: wave priorities dealt so that the two workgroups of a CU interleave (the arbiter serves the oldest wave first)
    int pk_wx = 150, pk_wxp = 150, pk_wz = 115;  //   tiling by cost: a row segment across the edge of / wholly inside the x C-PML layers, a row inside the z layers, in percent of a plain one
    int img_every = 1;    // imaging condition on every k-th backward step with weight k dt (1 = every step, the reference; k > 1 is an
                          // opt-in quadrature of the same time integral, exact for wavefields sampled above twice their bandwidth)
};
KernelOptions kernel_options();                       // snapshot of the defaults
int get_kernel_option(const char *name);              // -1: unknown
int set_kernel_option(const char *name, int value);   // 0 or -1

void launch_stress_fwd(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                       float *frame_t, int z_src, int x_src, float src_amp, LineRec lr);
void launch_velocity_fwd(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc);
void launch_velocity_rev(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, Media md, PmlCoef pc,
                         const float *frame_t, int z_src, int x_src, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc);
void launch_stress_rev(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, Media md, PmlCoef pc, float *frame_t,
                       int z_src, int x_src, float src_amp, Fields adj, ImgAcc acc);
void launch_velocity_adj(hipStream_t st, const Grid &g, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc);
void launch_stress_adj(hipStream_t st, const Grid &g, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc);
void launch_bwd_a(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                  const float *frame_t, Fields adj, ImgAcc acc);
void launch_bwd_b(hipStream_t st, const Grid &g, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t,
                  int z_src, int x_src, float src_amp, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc, LineRec lr,
                  hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// batched forms (one grid over tiles x shots of the batch; ShotDev table in device memory)
void launch_stress_fwd_batch(hipStream_t st, const Grid &g, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                             PmlCoef pc, size_t n, size_t data_len, int it, float src_scale, bool save);
void launch_velocity_fwd_batch(hipStream_t st, const Grid &g, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                               PmlCoef pc, size_t n);
void launch_bwd_a_batch(hipStream_t st, const Grid &g, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it);
void launch_bwd_b_batch(hipStream_t st, const Grid &g, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it, float src_scale, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
// persistent backward time loop (tiles: persist_plan.hpp).  persist_config_check: 0, or < 0 when this grid cannot be resident at once /
// the LDS does not fit (once per configuration).  launch_bwd_persist: the launch of a checked configuration (0, or -1: no such kernel).
int persist_config_check(int nwg, int threads, int lmask, size_t lds_bytes, bool multi_shot = false);
int launch_bwd_persist(hipStream_t st, const Grid &g, const KernelOptions &o, const PersistArgs &args, int nwg, int threads, int lmask,
                       size_t lds_bytes, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_add_inplace(hipStream_t st, float *a, const float *b, size_t n);
void launch_record(hipStream_t st, const Grid &g, Fields f, int nrec, const int *rec_idx, float *d_pr, float *d_vx,
                   float *d_vz, float *d_ett, int comps, const float *sens = nullptr);
// sens: null (straight fibre along x, or along z when g.fiber) or nrec x 3 directional sensitivities (s_xx, s_zz, s_xz)
void launch_inject(hipStream_t st, const Grid &g, Fields adj, int nrec, const int *rec_idx, const float *res_t,
                   const float *sens = nullptr);
// batched receivers (one launch for the shots of a batch that are NOT served inside the field kernels): seismogram column / residual column
void launch_record_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, int max_nrec, size_t n, size_t data_len, int column);
void launch_inject_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, int max_nrec, size_t n, int it);
// the residual [it][rec] folded per injection target and time step: val[it][t] = sum over the target's entries of w r[it][rec] (inject_plan.hpp)
void launch_inject_values(hipStream_t st, const float *res, int nrec, int nSteps, const int *tgt_start, const int *ent_rec, const float *ent_w, int ntgt,
                          float *val);
void launch_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, long long n,
                     double *sumsq);
void launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols);
void launch_model_prep(hipStream_t st, const Grid &g, const KernelOptions &o, const float *Lam_in, const float *Mu_in,
                       const float *Den_in, float *lam, float *mu, float *ave_mu, float *byc_a, float *byc_b, float *rho,
                       unsigned int *cp2_max_bits);
void launch_finalize_gradients(hipStream_t st, const Grid &g, Media md, ImgAcc acc, float *gLam, float *gMu,
                               float *gDen);

}  // namespace sepfwi
