/*
 * sepfwi.h -- C ABI of libsepfwi.so: MI355X-native 2-D elastic staggered-grid forward/adjoint
 * propagator (velocity-stress, C-PML, boundary-saving reconstruction, DAS axial-strain receivers).
 *
 * This is the drop-in boundary for the hot path of seisfwi/SEP-2023 "TorchFWI-DAS".  Every entry
 * point names the reference interface it replaces (paths relative to
 * DAS_Waveform_Inversion/Ops/FWI/ in the reference repository).  Plain pointers and sizes only; no
 * torch types, no C++ types.  All functions return 0 on success and a negative SEPFWI_E* code on
 * failure; sepfwi_last_error() then holds a message (the reference printf()s and exit(1)s instead:
 * Src/utilities.h:28-36, Src/utilities.cu:12-16,237-240).
 *
 * Pointer arguments documented as "host or device" may be either: transfers use
 * hipMemcpyDefault, so a torch CPU tensor's data_ptr and a torch HIP tensor's data_ptr both work.
 */
#ifndef SEPFWI_H_
#define SEPFWI_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SEPFWI_OK 0
#define SEPFWI_EINVAL (-1)   /* bad argument / inconsistent sizes                         */
#define SEPFWI_EIO (-2)      /* parameter, survey or data file missing / unreadable       */
#define SEPFWI_ECOURANT (-3) /* Courant number > 1            (Src/utilities.cu:225-241)  */
#define SEPFWI_EHIP (-4)     /* HIP runtime error (no device, out of memory, launch fault)*/
#define SEPFWI_EJSON (-5)    /* malformed parameter / survey JSON                         */

/* calc_id values (Src/libCUFD.cu:22-25, Src/Parameter.cpp:125-137) */
#define SEPFWI_CALC_MISFIT 0   /* forward + residual -> misfit                      */
#define SEPFWI_CALC_GRADIENT 1 /* + boundary saving, adjoint, gradients             */
#define SEPFWI_CALC_OBSERVE 2  /* forward only, write Shot_{pr,vx,vz,ett}{id}.bin   */
/* Extension (no counterpart in the reference): forward only, and the axial-strain gather of every shot goes straight into the
 * session's HBM store of observed data -- what SEPFWI_CALC_OBSERVE followed by reading Shot_ett{id}.bin back would leave there,
 * bit for bit, without the four files per shot (synthetic studies, benchmarks).  No other output. */
#define SEPFWI_CALC_OBSERVE_TO_STORE 3

/* Message of the last failure on the calling thread (never NULL). */
const char *sepfwi_last_error(void);

/* Library version, e.g. 100 = 0.1.0 */
int sepfwi_version(void);

/* Number of visible HIP devices, or a negative error code. */
int sepfwi_device_count(void);

/*
 * Replaces:  extern "C" void cufd(float *misfit, float *grad_Lambda, float *grad_Mu,
 *                float *grad_Den, float *grad_stf, const float *Lambda, const float *Mu,
 *                const float *Den, const float *stf, int calc_id, const int gpu_id,
 *                const int group_size, const int *shot_ids, const string para_fname)
 *            Src/libCUFD.h:6-10, Src/libCUFD.cu:32-820
 * called from Src/Torch_Fwi.cpp:31,86,132.  Same argument order and meaning; std::string became
 * const char*, void became an error code.
 *
 *   Lambda, Mu   [MPa], Den [kg/m^3]: (nz, nx) row-major float32, nz/nx the padded sizes of the
 *                parameter file (FWI_ops.py:124-127).  Host or device.
 *   stf          (nSrc, nSteps) row-major; row shot_ids[i] is shot i's source (Src/Src_Rec.cu:9,132).
 *                Host or device.
 *   shot_ids     group_size ints, host.
 *   misfit       1 float (calc_id 0,1): 0.5 * sum over shots of sum r_ett^2 (libCUFD.cu:427,776).
 *   grad_Lambda, grad_Mu, grad_Den   (nz, nx) row-major, OVERWRITTEN with this call's sum over its
 *                shots (calc_id 1).  Gradients are w.r.t. MPa for Lambda/Mu.  Host or device.
 *   grad_stf     (group_size, nSteps): row i = shot_ids[i] (local position, libCUFD.cu:671-673).
 *   para_fname   one-line JSON written by fwi_utils.paraGen (fwi_utils.py:46-83); names the survey
 *                JSON (fwi_utils.py:87-124) and the data directory holding
 *                Shot_{pr,vx,vz,ett}{id}.bin, float32 [nrec][nSteps] (libCUFD.cu:216-223,755-769).
 *                One optional key beyond the reference's schema: "das_fiber": "horizontal" (default: ett = exx,
 *                recording_exx / res_injection_exx) or "vertical" (ett = ezz, recording_ezz / res_injection_ezz,
 *                Src/utilities.cu:620-641, which the reference reaches only through a source edit).  Further optional keys
 *                (INTEGRATION.md): "obs_pack_fname" -- one packed file of the survey's observed axial-strain gathers instead
 *                of four files per shot; "if_win", "filter", "if_cross_misfit", "if_src_update" -- the data-conditioning
 *                chain of Src/utilities.cu:733-1325, dormant in the reference's driver, live here for the axial-strain gathers.
 *
 * Unlike the reference, device state (fields, PML profiles, boundary buffers, observed data) is kept
 * in a per-(para_fname, gpu_id) session between calls; sepfwi_release_all() frees it.
 */
int sepfwi_cufd(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den,
                float *grad_stf, const float *Lambda, const float *Mu, const float *Den,
                const float *stf, int calc_id, int gpu_id, int group_size, const int *shot_ids,
                const char *para_fname);

/* Same as sepfwi_cufd, but all launches go to `hip_stream` (a hipStream_t, may be NULL), and with `async` != 0 and every output
 * pointer (misfit, gradients) a device pointer the call does not end with a device synchronisation: the outputs are complete when the
 * work queued on the stream is -- the caller synchronises its stream before reading them
 * (tests/test_gpu_parity.py::test_c_abi_on_a_caller_stream_without_final_synchronisation).  `async` is NOT a promise that the host
 * returns early: the call synchronises the stream internally between the passes of a shot (to form the misfit on the host side of
 * the forward pass, and once per backward pass to learn whether the persistent loop's grid started -- a pass that does not start must
 * be re-issued as per-step launches), so it returns when all but the final gradient kernels have run. */
int sepfwi_cufd_stream(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den,
                       float *grad_stf, const float *Lambda, const float *Mu, const float *Den,
                       const float *stf, int calc_id, int gpu_id, int group_size,
                       const int *shot_ids, const char *para_fname, void *hip_stream, int async);

/* Frees every cached session (device memory, cached observed data) of this process. */
void sepfwi_release_all(void);

/*
 * Observed axial-strain data of one shot from memory instead of Shot_ett{id}.bin (the reference re-reads
 * four files per shot on every call, Src/libCUFD.cu:216-223).  `ett` is [nrec][nSteps] float32 like the file, host or
 * device pointer; the session keeps a time-major copy in HBM and uses it for every later misfit / gradient call of that
 * shot until sepfwi_invalidate_observed() or sepfwi_release_all().  No file is needed for shots set this way.
 */
int sepfwi_set_observed(const char *para_fname, int gpu_id, int shot_id, const float *ett, int nrec, int nSteps);

/* Drops cached observed data (e.g. after the Shot_*.bin files were rewritten by another tool). */
void sepfwi_invalidate_observed(void);

/*
 * Host helpers exported for parity tests (they are what the session uses internally).
 *   sepfwi_cpml_profiles: the six 1-D C-PML arrays K, a, b, K_half, a_half, b_half of length N.
 *                         Replaces cpmlInit, Src/utilities.cu:243-359.
 *   sepfwi_stf_taper:     in-place sin^2/cos^2 taper of one trace.  Replaces the 5-argument
 *                         cuda_window, Src/utilities.cu:844-884 (ratio 0.001, Src/Src_Rec.cu:137).
 *   sepfwi_shot_split:    start offsets (ngpu+1 ints) of the contiguous shot blocks per GPU.
 *                         Replaces the sepBars logic of Src/Torch_Fwi.cpp:59-60,78-80.
 */
int sepfwi_cpml_profiles(float *K, float *a, float *b, float *K_half, float *a_half,
                         float *b_half, int N, int nPml, float dh, float f0, float dt);
int sepfwi_stf_taper(float *trace, int nt, float dt, float ratio);
int sepfwi_shot_split(int group_size, int ngpu, int *starts);

/*
 * Statistics of the most recent sepfwi_cufd* call on (para_fname, gpu_id): kernel time measured
 * with hipEvents on the session's stream, cell-update counts as defined in SURVEY.md section 8(d).
 */
typedef struct sepfwi_stats {
    double fwd_ms;            /* forward time loops, all shots of the call                    */
    double bwd_ms;            /* backward (reconstruction + adjoint + imaging) time loops     */
    double total_ms;          /* whole call, host wall clock                                  */
    double cell_updates;      /* N_c * (nSteps-1) * shots * (1 or 3)                          */
    long long fwd_steps;      /* forward time steps executed                                  */
    long long bwd_steps;      /* backward time steps executed                                 */
    long long launches;       /* kernel launches issued                                       */
    long long device_bytes;   /* device memory held by the session                            */
    int n_c;                  /* computed cells per step (nz-nPad)*(nx)  [PML included]       */
    double probe_kernel_us;   /* option "probe">0: mean duration of the sampled k_bwd_b launches (HIP events) */
    long long probe_calls;    /* number of sampled launches                                     */
    long long obs_device_bytes; /* observed-data store: gathers resident in HBM (part of device_bytes)            */
    long long obs_host_bytes;   /* ... and in the pinned host tier (only with a budget, key / option "obs_cache_mb") */
    long long obs_evictions;    /* gathers moved HBM -> host tier since the store was created                       */
    long long persist_steps;    /* backward time steps of the call that ran inside the persistent loop (option bwd_fuse = 4) */
    long long quiet_active;     /* option quiet_skip: row segments of the call's last shot whose stresses ever held a value ...  */
    long long quiet_total;      /* ... of this many (0: the option was off or the shot's receivers are not a fused line)         */
} sepfwi_stats;
int sepfwi_get_stats(const char *para_fname, int gpu_id, sepfwi_stats *out);

/*
 * Extension (no counterpart in the reference, whose driver has one way of running a backward step, Src/libCUFD.cu:545-631):
 * why the most recent backward passes of (para_fname, gpu_id) did NOT run in the persistent time loop -- "" while they did (or no
 * gradient call has asked yet): the grid is too small for the loop's tiles, the configuration cannot be resident at once, the start
 * rendezvous found the GPU busy, ...  Up to len - 1 characters into `why`, always terminated.  For multi-GPU runs: a rank that fell
 * back to per-step launches is 10 % slower than its peers and must be visible (bench.py prints every rank's string).
 */
int sepfwi_loop_status(const char *para_fname, int gpu_id, char *why, int len);

/*
 * Options (process-wide defaults; every sepfwi_cufd* call takes ONE snapshot of them when it starts).  Names, defaults, meaning:
 *   bwd_fuse 4    backward step: 0 the reference's four kernels + injection, 2 two fused launches, 4 the persistent time loop (else 2)
 *   batch 2       shots of a call: 0 one stream per forward lane, 1 batched launches, 2 chosen by grid size
 *   quiet_skip 0  1: updates of 64-cell row segments whose every input is exactly +0 are left out (bit-identical)
 *   obs_cache_mb 0  HBM budget of the observed-data store in MB (0: unlimited; the parameter-file key of the same name wins)
 *   probe 0       > 0: every probe-th backward launch is timed with HIP events (sepfwi_stats.probe_kernel_us)
 *   img_every 1   k > 1: imaging condition on every k-th backward step with weight k dt -- an opt-in quadrature, gradients within
 *                 1e-4 of every-step imaging; the ONLY option that changes results beyond round-off of the parity tolerances
 * Tuning knobs and timing-only switches exist only in a library built with -DSEPFWI_PROBES (csrc/kernels.hip); here they are unknown
 * names.  Returns SEPFWI_EINVAL for unknown names or values; sepfwi_get_option returns the current default or -1.
 */
int sepfwi_set_option(const char *name, int value);
int sepfwi_get_option(const char *name);

/*
 * Fused parameterisation maps for HIP-resident tensors: what the reference's nn.Modules compute with a
 * dozen elementwise torch kernels per iteration on the CPU -- replicate padding (fwi_utils.py:31-44 with the identity
 * resize), mask blend P_m = Mask P_pad + (1 - Mask) P_ref (FWI_ops.py:120-122), the Lame map -- in ONE launch, and the
 * whole chain rule back to the (nz, nx) parameters (Lame derivatives, mask, transpose of the padding) in ONE.
 *   kind: 0 (Vp, Vs, Den) FWI_ops.py:124-125 | 1 (Lambda, Mu, Den) :204 | 2 (IP, IS, Den) :261-262 |
 *         3 (Vp, Vs, IP) :326-328 | 4 (Vp, Vs, IS) :389-391 | 5 (porosity, clay content, water saturation) Voigt-Reuss-Hill
 *         :451-497 | 6 the same triple, Biot-Gassmann :567-611
 *   A, B, C            (nz, nx) physical grid;  *_ref, Mask, Lambda, Mu, Den, gLambda, gMu, gDen:
 *                      (nz + 2 nPml + nPad, nx + 2 nPml) padded grid; gA, gB, gC: (nz, nx).  All float32 row-major DEVICE
 *                      pointers of one device; launched on hip_stream (NULL: the default stream), not synchronised.
 */
int sepfwi_param_forward(int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                         const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, float *Lambda,
                         float *Mu, float *Den, void *hip_stream);
int sepfwi_param_backward(int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                          const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask,
                          const float *gLambda, const float *gMu, const float *gDen, float *gA, float *gB, float *gC,
                          void *hip_stream);

/*
 * Test hook: wavefield `which` (0..4: vz, vx, szz, sxx, sxz; 5..9: their adjoint twins) of forward lane `lane` as the last
 * sepfwi_cufd* call on (para_fname, gpu_id) left it, dense (nz - nPad, nx) row-major float32, host or device pointer.
 * After a gradient call the forward fields are the reverse-time RECONSTRUCTION run back to time step 0, i.e. they must
 * have returned to the zero initial state up to float32 round-off: the size-independent parity
 * property checked at the full 2000 x 1000 x 4000 size, where the CPU oracle cannot go.
 */
int sepfwi_debug_field(const char *para_fname, int gpu_id, int lane, int which, float *out);

#ifdef __cplusplus
}
#endif
#endif /* SEPFWI_H_ */
