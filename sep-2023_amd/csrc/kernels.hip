// kernels.hip -- hand-written gfx950 (CDNA4, wave64) kernels of the 2-D elastic propagator.
//
// What each kernel replaces in the reference (paths relative to DAS_Waveform_Inversion/Ops/FWI/Src):
//   k_stress<FWD>    el_stress(isFor=true)   el_stress.cu:50-87    + from_bnd x5 (utilities.cu:362-392,
//                    Boundary.cu:57-80) + add_source (utilities.cu:524-552)
//   k_velocity<FWD>  el_velocity(isFor=true) el_velocity.cu:45-82
//   k_velocity<REV>  el_velocity(isFor=false) el_velocity.cu:87-113 + source_grad (utilities.cu:719-730)
//                    + to_bnd(vz,vx) (utilities.cu:395-425)
//   k_stress<REV>    add_source(-) + el_stress(isFor=false) el_stress.cu:92-125 + to_bnd(szz,sxz,sxx)
//   k_velocity_adj   el_velocity_adj.cu:57-102
//   k_stress_adj     el_stress_adj.cu:53-97
//   k_bwd_a          k_velocity<REV> + k_stress_adj of the previous step in one launch   } the default backward step:
//   k_bwd_b          source_grad + k_stress<REV> + k_velocity_adj + line injection       } two launches
//   k_bwd_velocity / k_bwd_stress   the other legal pairing (option bwd_fuse=1)
//   k_record         recording, recording_vx, recording_vz, recording_exx / _ezz (utilities.cu:593-602,620-629,645-703)
//   k_inject         res_injection_exx / _ezz (utilities.cu:605-615,632-641)
//   k_residual       gpuMinus + cuda_cal_objective (utilities.cu:154-205)
//   k_model_prep     host transpose x MEGA (libCUFD.cu:71-77) + velInit/aveMuInit/aveBycInit
//                    (utilities.cu:109-152, Model.cu:66-87)
//   k_finalize_gradients  the atomicAdd sprays of el_stress.cu:112-123 / el_velocity.cu:105-110 in
//                    gather form, and the D2H transpose of libCUFD.cu:718-724 (not needed here)
//
// The device code lives in the kernels_*.hpp files included below -- ONE translation unit, because every kernel structure shares the
// update bodies as inline functions: kernels_device.hpp (cell / tile helpers, accumulator and memory-scope policies),
// kernels_bodies.hpp (the four updates), kernels_quiet.hpp (quiet_skip), kernels_step.hpp (per-step and batched kernels),
// kernels_persist.hpp (k_bwd_persist), kernels_aux.hpp (receivers, residual, model preparation, gradient finalisation); this file
// keeps the options and the launchers.
//
// Arrays are row-major, x fastest, pitch `g.pitch` (fwi_types.hpp).  A wave covers 64 consecutive
// x of one row, so every global access of a wave is one 256-B line-aligned segment (plus the +-1/+-2
// shifted re-reads that hit the same lines in the vector L1).  Every cell update reads all its operands before its
// first store (one memory round trip per wave, DESIGN.md 3.1 "loads first").
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "device_common.hpp"
#include "kernels.hpp"
#include "persist_plan.hpp"

namespace sepfwi {
#include "kernels_device.hpp"
#include "kernels_bodies.hpp"
#include "kernels_quiet.hpp"
#include "kernels_step.hpp"
#include "kernels_persist.hpp"
#include "kernels_aux.hpp"

// =============================================================================================
// options + launchers
// =============================================================================================
// Process-wide DEFAULTS of the kernel options (sepfwi_set_option).  Nothing on the launch path reads them: every
// Session::run takes one snapshot (kernel_options()) and hands it to the launchers, so sessions running on different
// host threads (one per GPU) never see a half-updated block.
static std::mutex g_opt_mu;
static KernelOptions g_opt;

KernelOptions kernel_options() {
    std::lock_guard<std::mutex> lock(g_opt_mu);
    return g_opt;
}

namespace {
struct OptField {
    const char *name;
    int KernelOptions::*field;
    int lo, hi;
};
// The public table: what a user may set.  Nothing here changes results beyond the parity tolerances (img_every is the one documented
// quadrature, include/sepfwi.h).  Everything else -- tile shapes, wave counts, launch structures kept for A/B measurements and for the
// bit-identity tests of every selectable structure, and the timing-only switches that give WRONG results -- exists only in a library
// built with -DSEPFWI_PROBES (libsepfwi_probes.so: sepfwi/_native.py, scripts/ab_bench.py).
const OptField kOptFields[] = {
    {"bwd_fuse", &KernelOptions::bwd_fuse, 0, 4},
    {"batch", &KernelOptions::batch, 0, 2},
    {"img_every", &KernelOptions::img_every, 1, 64},
    {"quiet_skip", &KernelOptions::quiet_skip, 0, 1},
    {"obs_cache_mb", &KernelOptions::obs_cache_mb, 0, 1 << 30},
    {"probe", &KernelOptions::probe, 0, 1 << 30},
#ifdef SEPFWI_PROBES
    {"bz", &KernelOptions::bz, 1, 16},           {"xcd_remap", &KernelOptions::xcd_remap, 0, 1},
    {"line_fuse", &KernelOptions::line_fuse, 0, 1},
    {"pair_fwd", &KernelOptions::pair_fwd, 0, 1}, {"fwd_lanes", &KernelOptions::fwd_lanes, 1, 4},
    {"early", &KernelOptions::early, 0, 3},       {"rho_fly", &KernelOptions::rho_fly, 0, 3},
    {"amu_fly", &KernelOptions::amu_fly, 0, 3},   {"rk_lazy", &KernelOptions::rk_lazy, 0, 1},
    {"batch_f", &KernelOptions::batch_f, 0, 64},
    {"batch_b", &KernelOptions::batch_b, 0, 64},  {"batch_mb", &KernelOptions::batch_mb, 1, 1 << 20},
    {"batch_order", &KernelOptions::batch_order, 0, 1}, {"batch_split", &KernelOptions::batch_split, 1, 3},
    {"quiet_rows", &KernelOptions::quiet_rows, 1, 16},
    {"pk_lmask", &KernelOptions::pk_lmask, 0, 31},  {"pk_wpc", &KernelOptions::pk_wpc, 1, 4},
    {"pk_px", &KernelOptions::pk_px, 1, 64},         {"pk_waves", &KernelOptions::pk_waves, 4, 16}, {"pk_order", &KernelOptions::pk_order, 0, 2},
    {"pk_prio", &KernelOptions::pk_prio, 0, 3}, {"pk_wx", &KernelOptions::pk_wx, 25, 400}, {"pk_wxp", &KernelOptions::pk_wxp, 25, 400}, {"pk_wz", &KernelOptions::pk_wz, 25, 400},
    // timing experiments only -- WRONG results: no synchronisation between tiles; phases of a time step interleaved (pk_lock: bits 0-7 the
    // distance D in row segments by which phase B trails phase A, 0x100 alternate walk direction per time step, 0x200 no barrier per step);
    // pk_snake 0: every strip of the tiling is walked top-down
    {"pk_nosync", &KernelOptions::pk_nosync, 0, 1}, {"pk_lock", &KernelOptions::pk_lock, 0, 0x3ff}, {"pk_snake", &KernelOptions::pk_snake, 0, 1},
    // an experiment that lost (EXPERIMENTS #48): the batched schedule's backward sub-batches as ONE multi-shot persistent launch
    {"pk_ms", &KernelOptions::pk_ms, 0, 1},
    // another one (EXPERIMENTS #49): with quiet_skip on, the persistent loop in its quiet-segment variant instead of the two-launch step
    {"pk_quiet", &KernelOptions::pk_quiet, 0, 1},
#endif
};
}  // namespace

int get_kernel_option(const char *name) {
    const std::string n(name ? name : "");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    for (const OptField &f : kOptFields)
        if (n == f.name) return g_opt.*(f.field);
    return -1;
}

int set_kernel_option(const char *name, int value) {
    const std::string n(name ? name : "");
    std::lock_guard<std::mutex> lock(g_opt_mu);
    for (const OptField &f : kOptFields)
        if (n == f.name) {
            if (value < f.lo || value > f.hi || (n == "bwd_fuse" && (value == 1 || value == 3))) return -1;
            g_opt.*(f.field) = value;
            return 0;
        }
    return -1;
}

static inline Grid tiled(const Grid &g0, const KernelOptions &o, int fly_bit = -1, bool quiet = false) {
    Grid g = g0;
    g.bz = o.bz;
    g.qr = quiet ? o.quiet_rows : 1;
    g.gx = (g.nx + BX - 1) / BX;
    g.gy = (g.nzc + g.bz * g.qr - 1) / (g.bz * g.qr);
    g.xcd_remap = o.xcd_remap;
    g.rho_fly = fly_bit < 0 ? 0 : (o.rho_fly >> fly_bit) & 1;
    g.amu_fly = fly_bit < 0 ? 0 : (o.amu_fly >> fly_bit) & 1;
    g.rk_lazy = o.rk_lazy;
    const int nb = g.gx * g.gy;
    g.nblk = g.xcd_remap ? ((nb + 7) / 8) * 8 : nb;
    return g;
}
static inline dim3 field_grid(const Grid &g) { return dim3(g.nblk); }
#define BLOCK dim3(BX *g.bz)

void launch_stress_fwd(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                       float *frame_t, int z_src, int x_src, float src_amp, LineRec lr) {
    const Grid g = tiled(g0, o, 0, f.q != nullptr);
    Fields none{};
    ImgAcc na{};
    auto k = frame_t ? (f.q ? k_stress<true, true, true> : k_stress<true, true, false>) : (f.q ? k_stress<true, false, true> : k_stress<true, false, false>);
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, f, m, md, pc, frame_t, z_src, x_src, src_amp, none, na, lr);
}

void launch_velocity_fwd(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 0, f.q != nullptr);
    Fields none{};
    ImgAcc na{};
    auto k = f.q ? k_velocity<true, true> : k_velocity<true, false>;
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, f, m, md, pc, (const float *)nullptr, -1, -1, 0.0f, (float *)nullptr, none, na);
}

void launch_velocity_rev(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, Media md, PmlCoef pc,
                         const float *frame_t, int z_src, int x_src, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    PmlMem nm{};
    hipLaunchKernelGGL((k_velocity<false>), field_grid(g), BLOCK, 0, st, g, f, nm, md, pc, frame_t, z_src, x_src,
                       src_rxz, stf_grad_it, adj, acc);
}

void launch_stress_rev(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, Media md, PmlCoef pc, float *frame_t,
                       int z_src, int x_src, float src_amp, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    PmlMem nm{};
    hipLaunchKernelGGL((k_stress<false, false>), field_grid(g), BLOCK, 0, st, g, f, nm, md, pc, frame_t, z_src,
                       x_src, src_amp, adj, acc, LineRec{});
}

void launch_velocity_adj(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 1);
    hipLaunchKernelGGL(k_velocity_adj, field_grid(g), BLOCK, 0, st, g, adj, m, md, pc);
}

void launch_stress_adj(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields adj, PmlMem m, Media md, PmlCoef pc) {
    const Grid g = tiled(g0, o, 1);
    hipLaunchKernelGGL(k_stress_adj, field_grid(g), BLOCK, 0, st, g, adj, m, md, pc);
}

// The fused backward kernels (and every batched kernel) receive their arrays as bundles: base pointer + one stride, the C-PML
// profiles as ONE block (six z profiles of nzc floats, then six x profiles of nx).  The session allocates them that way; a
// caller that does not must hear about it here, not through a kernel reading the wrong array.
static void check_bundles(const Grid &g, const Fields &f, const PmlMem &m, const Media &md, const PmlCoef &pc, const Fields &adj, const ImgAcc &acc) {
    const ptrdiff_t n = f.vx - f.vz;
    const bool fields_ok = f.szz - f.vx == n && f.sxx - f.szz == n && f.sxz - f.sxx == n;
    const bool adj_ok = adj.vx - adj.vz == n && adj.szz - adj.vx == n && adj.sxx - adj.szz == n && adj.sxz - adj.sxx == n;
    const bool mem_ok = m.dvz_dx - m.dvz_dz == n && m.dvx_dz - m.dvz_dx == n && m.dvx_dx - m.dvx_dz == n && m.dszz_dz - m.dvx_dx == n &&
                        m.dsxz_dx - m.dszz_dz == n && m.dsxz_dz - m.dsxz_dx == n && m.dsxx_dx - m.dsxz_dz == n;
    const bool media_ok = md.mu - md.lam == n && md.ave_mu - md.mu == n && md.byc_a - md.ave_mu == n && md.byc_b - md.byc_a == n && md.rho - md.byc_b == n;
    const bool acc_ok = acc.mu - acc.lam == n && acc.xz - acc.mu == n && acc.a - acc.xz == n && acc.b - acc.a == n;
    const ptrdiff_t z = g.nzc, x = g.nx;
    const bool coef_ok = pc.b_z - pc.a_z == z && pc.rK_z - pc.b_z == z && pc.a_zh - pc.rK_z == z && pc.b_zh - pc.a_zh == z && pc.rK_zh - pc.b_zh == z &&
                         pc.a_x - pc.rK_zh == z && pc.b_x - pc.a_x == x && pc.rK_x - pc.b_x == x && pc.a_xh - pc.rK_x == x && pc.b_xh - pc.a_xh == x &&
                         pc.rK_xh - pc.b_xh == x;
    if (!(n > 0 && fields_ok && adj_ok && mem_ok && media_ok && acc_ok && coef_ok))
        throw std::logic_error("bundled kernel launch: arrays are not laid out as base + k * stride (fields " + std::to_string(fields_ok) + ", adjoint " +
                               std::to_string(adj_ok) + ", memories " + std::to_string(mem_ok) + ", media " + std::to_string(media_ok) + ", accumulators " +
                               std::to_string(acc_ok) + ", C-PML profiles " + std::to_string(coef_ok) + ")");
}

void launch_bwd_a(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc,
                  const float *frame_t, Fields adj, ImgAcc acc) {
    const Grid g = tiled(g0, o, 1);
    check_bundles(g, f, m, md, pc, adj, acc);
    const BwdArgs b{f.vz, m.dvz_dz, adj.vz, md.lam, acc.lam, pc.a_z, (size_t)(f.vx - f.vz), f.q, adj.q};
    auto k = f.q ? k_bwd_a<false, true> : (o.early & 1) ? k_bwd_a<true, false> : k_bwd_a<false, false>;
    hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, b, frame_t);
}

void launch_bwd_b(hipStream_t st, const Grid &g0, const KernelOptions &o, Fields f, PmlMem m, Media md, PmlCoef pc, float *frame_t,
                  int z_src, int x_src, float src_amp, float src_rxz, float *stf_grad_it, Fields adj, ImgAcc acc, LineRec lr,
                  hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled(g0, o, 1);
    check_bundles(g, f, m, md, pc, adj, acc);
    const BwdArgs b{f.vz, m.dvz_dz, adj.vz, md.lam, acc.lam, pc.a_z, (size_t)(f.vx - f.vz), f.q, adj.q};
    auto k = f.q ? k_bwd_b<false, true> : (o.early & 2) ? k_bwd_b<true, false> : k_bwd_b<false, false>;
    if (ev_start)  // timestamps taken by the command processor at kernel begin / end (no launch gap included)
        hipExtLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, ev_start, ev_stop, 0, g, b, frame_t, (z_src << 16) | x_src, src_amp,
                              src_rxz, stf_grad_it, (lr.z << 16) | lr.x0, lr.n, lr.res);
    else
        hipLaunchKernelGGL(k, field_grid(g), BLOCK, 0, st, g, b, frame_t, (z_src << 16) | x_src, src_amp, src_rxz,
                           stf_grad_it, (lr.z << 16) | lr.x0, lr.n, lr.res);
}

// kind: 0 fused line of channels (or none), 1 general receivers (GINJ), 2 several shots per launch (MS), 3 quiet row segments (QS)
static void (*persist_kernel(int lmask, int kind))(Grid, const PersistArgs) {
#ifdef SEPFWI_PROBES  // the multi-shot and the quiet-segment instances exist in the probe build only: both are bit-identical and both measured
                      // slower than what the shipped library does instead (profiles/EXPERIMENTS.md #48, #49)
#define SEPFWI_PK(M) (kind == 2 ? k_bwd_persist<M, false, true> : kind == 3 ? k_bwd_persist<M, false, false, true> : kind == 1 ? k_bwd_persist<M, true> : k_bwd_persist<M>)
#else
#define SEPFWI_PK(M) (kind >= 2 ? nullptr : kind == 1 ? k_bwd_persist<M, true> : k_bwd_persist<M>)
#endif
    switch (lmask) {
        case 0: return SEPFWI_PK(0);
        case 1: return SEPFWI_PK(1);
        case 3: return SEPFWI_PK(3);
        case 7: return SEPFWI_PK(7);
        case 15: return SEPFWI_PK(15);
        case 31: return SEPFWI_PK(31);
        default: return nullptr;
    }
#undef SEPFWI_PK
}

// 0, or why this grid cannot run the persistent loop (never launch a grid that would not be resident at once: its tiles wait for
// each other): -1 no kernel instance for this LDS mask, -2 the LDS request is refused, -3 the occupancy query fails, -4 fewer
// workgroups fit the device than the grid has.  Asked ONCE per configuration (Session::persist_ready); it also raises the kernel's
// dynamic-LDS limit, which the launches rely on.
int persist_config_check(int nwg, int threads, int lmask, size_t lds_bytes, bool multi_shot) {
    int dev = 0, ncu = 0;
    (void)hipGetDevice(&dev);
    (void)hipDeviceGetAttribute(&ncu, hipDeviceAttributeMultiprocessorCount, dev);
    for (int kind : {0, 1, 3}) {  // single shot: every instance of the configuration (fused line of channels / general receivers / quiet segments)
        if (multi_shot) kind = 2;
        const void *k = (const void *)persist_kernel(lmask, kind);
        if (!k && kind == 3) continue;  // (probe build only)
        if (!k) return -1;
        if (lds_bytes > 64 * 1024 && hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_bytes) != hipSuccess) return -2;
        int per_cu = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, threads, lds_bytes) != hipSuccess) return -3;
        if (per_cu * ncu < nwg) return -4;
        if (multi_shot) break;
    }
    return 0;
}

int launch_bwd_persist(hipStream_t st, const Grid &g0, const KernelOptions &o, const PersistArgs &args, int nwg, int threads, int lmask,
                       size_t lds_bytes, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled(g0, o, 1);
    auto k = persist_kernel(lmask, args.ms.nshot > 0 ? 2 : args.q.maps != nullptr ? 3 : args.injp != nullptr ? 1 : 0);
    if (!k) return -1;
#ifdef SEPFWI_PK_TRACE
    {
        static unsigned long long *d_tr = nullptr;
        static size_t tr_n = (size_t)kTrTiles * 16 * kTrPh * kTrSlots;
        if (!d_tr && getenv("SEPFWI_PK_TRACE")) {
            (void)hipMalloc((void **)&d_tr, tr_n * sizeof(unsigned long long));
            (void)hipMemset(d_tr, 0, tr_n * sizeof(unsigned long long));
            (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pk_trace), &d_tr, sizeof(d_tr));
            atexit([] {
                std::vector<unsigned long long> h(tr_n);
                (void)hipDeviceSynchronize();
                (void)hipMemcpy(h.data(), d_tr, tr_n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
                if (FILE *fp = fopen(getenv("SEPFWI_PK_TRACE"), "wb")) {
                    fwrite(h.data(), sizeof(unsigned long long), tr_n, fp);
                    fclose(fp);
                }
            });
        }
    }
#endif
    if (ev_start)
        hipExtLaunchKernelGGL(k, dim3(nwg), dim3(threads), lds_bytes, st, ev_start, ev_stop, 0, g, args);
    else
        hipLaunchKernelGGL(k, dim3(nwg), dim3(threads), lds_bytes, st, g, args);
    return 0;
}

__global__ void k_add_inplace(float *__restrict__ a, const float *__restrict__ b, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] += b[i];
}
void launch_add_inplace(hipStream_t st, float *a, const float *b, size_t n) {
    hipLaunchKernelGGL(k_add_inplace, dim3(4096), dim3(256), 0, st, a, b, n);
}

// ---- batched launchers: one grid over (tile, shot of the batch); the order of the two is Grid::nb / batch_order
static inline Grid tiled_batch(const Grid &g0, const KernelOptions &o, int fly_bit, int nb) {
    Grid g = tiled(g0, o, fly_bit);
    g.nb = nb;
    g.shot_fastest = o.batch_order;
    const int nblk = g.gx * g.gy * g.nb;
    g.nblk = g.xcd_remap ? ((nblk + 7) / 8) * 8 : nblk;
    return g;
}
static void check_shared_bundles(const Grid &g, const Media &md, const PmlCoef &pc) {
    const ptrdiff_t n = md.mu - md.lam, z = g.nzc, x = g.nx;
    const bool media_ok = n > 0 && md.ave_mu - md.mu == n && md.byc_a - md.ave_mu == n && md.byc_b - md.byc_a == n && md.rho - md.byc_b == n;
    const bool coef_ok = pc.b_z - pc.a_z == z && pc.rK_z - pc.b_z == z && pc.a_zh - pc.rK_z == z && pc.b_zh - pc.a_zh == z && pc.rK_zh - pc.b_zh == z &&
                         pc.a_x - pc.rK_zh == z && pc.b_x - pc.a_x == x && pc.rK_x - pc.b_x == x && pc.a_xh - pc.rK_x == x && pc.b_xh - pc.a_xh == x &&
                         pc.rK_xh - pc.b_xh == x;
    if (!(media_ok && coef_ok)) throw std::logic_error("batched kernel launch: media or C-PML profiles are not laid out as one bundle");
}
static inline dim3 batch_grid(const Grid &g) { return dim3(g.nblk); }
void launch_stress_fwd_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                             PmlCoef pc, size_t n, size_t data_len, int it, float src_scale, bool save) {
    const Grid g = tiled_batch(g0, o, 0, nb);
    check_shared_bundles(g, md, pc);
    const bool q = o.quiet_skip != 0;  // (per shot: ShotDev::quiet)
    auto k = save ? (q ? k_stress_fwd_batch<true, true> : k_stress_fwd_batch<true, false>) : (q ? k_stress_fwd_batch<false, true> : k_stress_fwd_batch<false, false>);
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, data_len, it, src_scale);
}
void launch_velocity_fwd_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md,
                               PmlCoef pc, size_t n) {
    const Grid g = tiled_batch(g0, o, 0, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_velocity_fwd_batch<true> : k_velocity_fwd_batch<false>;
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n);
}
void launch_bwd_a_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it) {
    const Grid g = tiled_batch(g0, o, 1, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_bwd_a_batch<false, true> : (o.early & 1) ? k_bwd_a_batch<true, false> : k_bwd_a_batch<false, false>;
    hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, it);
}
void launch_bwd_b_batch(hipStream_t st, const Grid &g0, const KernelOptions &o, const ShotDev *shots, int nb, Media md, PmlCoef pc,
                        size_t n, int it, float src_scale, hipEvent_t ev_start, hipEvent_t ev_stop) {
    const Grid g = tiled_batch(g0, o, 1, nb);
    check_shared_bundles(g, md, pc);
    auto k = o.quiet_skip ? k_bwd_b_batch<false, true> : (o.early & 2) ? k_bwd_b_batch<true, false> : k_bwd_b_batch<false, false>;
    if (ev_start)
        hipExtLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, ev_start, ev_stop, 0, g, shots, md.lam, pc.a_z, n, it, src_scale);
    else
        hipLaunchKernelGGL(k, batch_grid(g), BLOCK, 0, st, g, shots, md.lam, pc.a_z, n, it, src_scale);
}

void launch_record(hipStream_t st, const Grid &g, Fields f, int nrec, const int *rec_idx, float *d_pr, float *d_vx,
                   float *d_vz, float *d_ett, int comps, const float *sens) {
    if (nrec <= 0) return;
    hipLaunchKernelGGL(k_record, dim3((nrec + 255) / 256), dim3(256), 0, st, g, f, nrec, rec_idx, d_pr, d_vx, d_vz, d_ett,
                       comps, g.fiber, sens);
}

void launch_inject(hipStream_t st, const Grid &g, Fields adj, int nrec, const int *rec_idx, const float *res_t, const float *sens) {
    if (nrec <= 0) return;
    hipLaunchKernelGGL(k_inject, dim3((nrec + 255) / 256), dim3(256), 0, st, adj, nrec, rec_idx, res_t, g.fiber ? g.pitch : 0,
                       sens, g.pitch, g.dx * g.rdz);
}

void launch_record_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, int max_nrec, size_t n, size_t data_len, int column) {
    if (nb <= 0 || max_nrec <= 0) return;
    hipLaunchKernelGGL(k_record_batch, dim3((max_nrec + 255) / 256, nb), dim3(256), 0, st, g, shots, n, data_len, column);
}
void launch_inject_batch(hipStream_t st, const Grid &g, const ShotDev *shots, int nb, int max_nrec, size_t n, int it) {
    if (nb <= 0 || max_nrec <= 0) return;
    hipLaunchKernelGGL(k_inject_batch, dim3((max_nrec + 255) / 256, nb), dim3(256), 0, st, g, shots, n, it);
}

void launch_inject_values(hipStream_t st, const float *res, int nrec, int nSteps, const int *tgt_start, const int *ent_rec, const float *ent_w, int ntgt,
                          float *val) {
    if (ntgt <= 0 || nSteps <= 0) return;
    hipLaunchKernelGGL(k_inject_values, dim3((ntgt + 255) / 256, nSteps), dim3(256), 0, st, res, nrec, tgt_start, ent_rec, ent_w, ntgt, val);
}

void launch_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, long long n,
                     double *sumsq) {
    if (n <= 0) return;  // a shot without receivers contributes nothing
    int blocks = (int)((n + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(k_residual, dim3(blocks), dim3(256), 0, st, obs, syn, res, nrec, n, sumsq);
}

void launch_transpose(hipStream_t st, const float *in, float *out, int rows, int cols) {
    if (rows <= 0 || cols <= 0) return;
    hipLaunchKernelGGL(k_transpose, dim3((cols + 31) / 32, (rows + 31) / 32), dim3(32, 8), 0, st, in, out, rows, cols);
}

void launch_model_prep(hipStream_t st, const Grid &g, const KernelOptions &o, const float *Lam_in, const float *Mu_in,
                       const float *Den_in, float *lam, float *mu, float *ave_mu, float *byc_a, float *byc_b, float *rho,
                       unsigned int *cp2_max_bits) {
    hipLaunchKernelGGL(k_model_prep, dim3((g.nx + 63) / 64, (g.nz + 3) / 4), dim3(64, 4), 0, st, g, Lam_in, Mu_in, Den_in,
                       lam, mu, ave_mu, byc_a, byc_b, rho, cp2_max_bits, o.amu_fly != 0 ? 1 : 0);
}

void launch_finalize_gradients(hipStream_t st, const Grid &g, Media md, ImgAcc acc, float *gLam, float *gMu,
                               float *gDen) {
    hipLaunchKernelGGL(k_finalize_gradients, dim3((g.nx + 63) / 64, (g.nz + 3) / 4), dim3(64, 4), 0, st, g, md, acc, gLam,
                       gMu, gDen);
}

}  // namespace sepfwi
