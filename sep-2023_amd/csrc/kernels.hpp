// kernels.hpp -- host-callable launchers of the gfx950 kernels in kernels.hip.
#pragma once
#include <hip/hip_runtime.h>

#include "fwi_types.hpp"

namespace sepfwi {

void launch_stress_fwd(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t,
                       int z_src, int x_src, float src_amp, LineRec lr);
void launch_velocity_fwd(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc);
void launch_velocity_rev(hipStream_t st, const Grid &g, Fields f, Media md, PmlCoef pc, const float *frame_t, int z_src,
                         int x_src, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc);
void launch_stress_rev(hipStream_t st, const Grid &g, Fields f, Media md, PmlCoef pc, float *frame_t, int z_src,
                       int x_src, float src_amp, Fields adj, ImgAcc acc);
void launch_velocity_adj(hipStream_t st, const Grid &g, Fields adj, PmlMem m, Media md, PmlCoef pc);
void launch_stress_adj(hipStream_t st, const Grid &g, Fields adj, PmlMem m, Media md, PmlCoef pc);
void launch_bwd_velocity(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc, const float *frame_t,
                         int z_src, int x_src, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc, LineRec lr);
void launch_bwd_stress(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t, int z_src,
                       int x_src, float src_amp, Fields adj, ImgAcc acc, hipEvent_t ev_start = nullptr,
                       hipEvent_t ev_stop = nullptr);
void launch_bwd_a(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc, const float *frame_t, Fields adj,
                  ImgAcc acc, bool acc_nt = false);
void launch_bwd_b(hipStream_t st, const Grid &g, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t, int z_src,
                  int x_src, float src_amp, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc, LineRec lr,
                  hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr, bool acc_nt = false);
// batched forms (grid.y = shot of the batch; ShotDev table in device memory)
void launch_stress_fwd_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, Media md, PmlCoef pc, size_t n,
                             size_t data_len, int it, float src_scale, bool save);
void launch_velocity_fwd_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, Media md, PmlCoef pc, size_t n);
void launch_bwd_a_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, Media md, PmlCoef pc, size_t n, int it);
void launch_bwd_b_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, Media md, PmlCoef pc, size_t n, int it,
                        float src_scale, hipEvent_t ev_start = nullptr, hipEvent_t ev_stop = nullptr);
void launch_add_inplace(hipStream_t st, float *a, const float *b, size_t n);
int get_kernel_option_bwd_fuse();
int get_kernel_option(const char *name);
// fused forward step (fwd_fused.hip)
void fwd_fused_tile_shape(int *rows, int *cols);
void launch_fwd_fused(hipStream_t st, const Grid &g, const FwdFusedArgs &a, int xcd_remap);
void launch_fwd_march(hipStream_t st, const Grid &g, const FwdFusedArgs &a, LineRec lr, int xcd_remap);
void launch_record(hipStream_t st, const Grid &g, Fields f, int nrec, const int *rec_idx, float *d_pr, float *d_vx,
                   float *d_vz, float *d_ett, int comps);
void launch_inject(hipStream_t st, Fields adj, int nrec, const int *rec_idx, const float *res_t, int down = 0);
void launch_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, long long n,
                     double *sumsq);
void launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols);
void launch_model_prep(hipStream_t st, const Grid &g, const float *Lam_in, const float *Mu_in, const float *Den_in,
                       float *lam, float *mu, float *ave_mu, float *byc_a, float *byc_b, float *rho, unsigned int *cp2_max_bits);
void launch_finalize_gradients(hipStream_t st, const Grid &g, Media md, ImgAcc acc, float *gLam, float *gMu,
                               float *gDen);

// persistent forward time loop (fwd_persist.hip): one launch per shot
int persist_bands(const Grid &g, int n_cus);            // 0: grid not supported
size_t persist_halo_floats(const Grid &g, int nb);      // per halo buffer
bool launch_fwd_persist(hipStream_t st, const Grid &g, const ShotDev &shot, Media md, PmlCoef pc, size_t n, size_t data_len,
                        float src_scale, int nsteps, int nb, bool save, float *haloV, float *haloS, int *flagV, int *flagS,
                        int *abort_flag, int rho_fly);

// run-time kernel options ("bz": rows per block 1..16, "xcd_remap": 0/1); returns 0 or -1
int set_kernel_option(const char *name, int value);

}  // namespace sepfwi
