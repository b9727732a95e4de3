// session_run.cpp -- Session::run, the cufd call (Src/libCUFD.cu:170-724), split into its passes:
//   prepare_media / prepare_buffers     set-up of one call (libCUFD.cu:39-165, the part that is not kept between calls)
//   forward_*  / residual*              forward time loop of one shot and its misfit (libCUFD.cu:268-332, 410-427)
//   after_forward                       seismogram files / HBM store / scratch dumps (libCUFD.cu:732-769)
//   backward_* / backward               boundary-saving adjoint time loop of one shot (libCUFD.cu:500-675); as one persistent launch:
//                                       session_persist.cpp
//   run_streams                         the stream schedule of a call's shots; the batched one: session_batched.cpp
//   write_outputs                       gradient finalisation and read-back (libCUFD.cu:710-724,775-779)
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>

#include "device_alloc.hpp"
#include "hip_check.hpp"
#include "kernels.hpp"
#include "session.hpp"

namespace sepfwi {

static const char *kComp[4] = {"pr", "vx", "vz", "ett"};  // libCUFD.cu:216-223,755-769

static std::string shot_file(const Params &p, int comp, int id) {
    return p.data_dir_name + "/Shot_" + kComp[comp] + std::to_string(id) + ".bin";
}

// Device that owns `p`, or -1 for host memory.  A pointer on ANOTHER device than the session's (the single-process
// ngpu > 1 path handing GPU-0 tensors to the session of GPU i) is staged like host memory: the kernels only ever touch
// memory of their own device, peer access is never assumed.
static int ptr_device(const void *p) {
    if (!p) return -1;
    hipPointerAttribute_t attr;
    hipError_t e = hipPointerGetAttributes(&attr, p);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // clear: plain host memory is reported as an error on some ROCm versions
        return -1;
    }
    return (attr.type == hipMemoryTypeDevice || attr.type == hipMemoryTypeManaged) ? attr.device : -1;
}

// A NULL stream argument means the legacy default stream, which is what torch's default stream is on ROCm: the call's
// own (non-blocking) stream must not start before the work already queued there -- the Lame maps that produced
// Lambda/Mu/Den, the zero-fill of the gradient tensors -- has finished.
void Session::order_after_null_stream(hipStream_t st) {
    HIP_OK(hipEventRecord(ev_order_, nullptr));
    HIP_OK(hipStreamWaitEvent(st, ev_order_, 0));
}

// ---- set-up of one call ------------------------------------------------------------------------------------------------
// media: boundary arrays -> internal layout, averages, Courant guard (utilities.cu:225-241, libCUFD.cu:90).  Inputs that do
// not live on this session's device (host memory, or another GPU's memory) are staged.
void Session::prepare_media(Call &c, const float *Lambda, const float *Mu, const float *Den) {
    hipStream_t st = c.st;
    const size_t n = cells_, dense = (size_t)par_.nz * (size_t)par_.nx;
    const float *dL = Lambda, *dM = Mu, *dD = Den;
    if (ptr_device(Lambda) != gpu_id_) { HIP_OK(hipMemcpyAsync(in_stage_, Lambda, dense * sizeof(float), hipMemcpyDefault, st)); dL = in_stage_; }
    if (ptr_device(Mu) != gpu_id_) { HIP_OK(hipMemcpyAsync(in_stage_ + dense, Mu, dense * sizeof(float), hipMemcpyDefault, st)); dM = in_stage_ + dense; }
    if (ptr_device(Den) != gpu_id_) { HIP_OK(hipMemcpyAsync(in_stage_ + 2 * dense, Den, dense * sizeof(float), hipMemcpyDefault, st)); dD = in_stage_ + 2 * dense; }
    HIP_OK(hipMemsetAsync(cp2_bits_, 0, sizeof(unsigned int), st));
    launch_model_prep(st, g_, c.opt, dL, dM, dD, media_, media_ + n, media_ + 2 * n, media_ + 3 * n, media_ + 4 * n, media_ + 5 * n, cp2_bits_);
    launches_++;
    unsigned int bits = 0;
    HIP_OK(hipMemcpyAsync(&bits, cp2_bits_, sizeof(bits), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    float cp2;
    std::memcpy(&cp2, &bits, sizeof(cp2));
    const float vmax = (float)std::sqrt((double)cp2);
    const float dh_min = (par_.dz < par_.dx) ? par_.dz : par_.dx;
    const float courant = (float)((double)(vmax * par_.dt * sqrtf(2.0f)) * (1.0 / 24.0 + 9.0 / 8.0) / (double)dh_min);
    if (!(courant <= 1.0f)) throw CourantError("Courant number " + std::to_string(courant) + " > 1 (vmax " + std::to_string(vmax) + " m/s)");
}

// boundary-saving storage (Boundary.cu:29-41, allocated on the first gradient call), zeroed accumulators (Model.cu:68-71) and
// misfit, the tapered source traces (row shot_ids[i] of stf, Src_Rec.cu:130-137), the source-gradient rows
void Session::prepare_buffers(Call &c, const float *stf) {
    hipStream_t st = c.st;
    const int nSteps = par_.nSteps;
    const size_t n = cells_;
    if (c.with_adj && !frame_) {
        const size_t fb = (size_t)nSteps * 5 * (size_t)g_.frame_len * sizeof(float);
        HIP_OK(dev_malloc((void **)&frame_, fb));
        device_bytes_ += (long long)fb;
    }
    if (c.with_adj) HIP_OK(hipMemsetAsync(acc_buf_, 0, 5 * n * sizeof(float), st));
    if (c.if_res) HIP_OK(hipMemsetAsync(scal_, 0, 4 * sizeof(double), st));
    c.stf_rows.resize((size_t)c.group_size * nSteps);
    for (int i = 0; i < c.group_size; i++) {
        HIP_OK(hipMemcpy(c.stf_rows.data() + (size_t)i * nSteps, stf + (size_t)c.shot_ids[i] * nSteps, nSteps * sizeof(float), hipMemcpyDefault));
        stf_taper(c.stf_rows.data() + (size_t)i * nSteps, nSteps, par_.dt, 0.001f);
    }
    c.src_scale = (float)std::pow(1500.0, 2);  // utilities.cu:531
    if (c.with_adj) {  // source-time-function gradients of all shots of the call, one row each
        const size_t need = (size_t)c.group_size * nSteps;
        if (need > stf_grad_len_) {
            if (stf_grad_) (void)hipFree(stf_grad_);
            stf_grad_ = nullptr;
            HIP_OK(dev_malloc((void **)&stf_grad_, need * sizeof(float)));
            device_bytes_ += (long long)((need - stf_grad_len_) * sizeof(float));
            stf_grad_len_ = need;
        }
        HIP_OK(hipMemsetAsync(stf_grad_, 0, need * sizeof(float), st));
    }
}

void Session::use_state(ShotCtx &x, float *state) const {
    const size_t n = cells_;
    x.state = state;
    float *b = state;
    x.fld = Fields{b, b + n, b + 2 * n, b + 3 * n, b + 4 * n};
    x.fld.q = x.quiet;
    x.mem = PmlMem{b + 5 * n, b + 6 * n, b + 7 * n, b + 8 * n, b + 9 * n, b + 10 * n, b + 11 * n, b + 12 * n};
}

// Shot `is` of the call in stream lane `lane` (0: the session's own state on the call's stream).  Acquires the shot's observed
// gather (held in HBM until the group of shots is through: ObservedStore::release_all) unless with_obs is false (geometry only).
Session::ShotCtx Session::make_ctx(const Call &c, int is, int lane, hipStream_t lane_st, bool with_obs) {
    const Grid &g = g_;
    ShotCtx x{};
    x.is = is;
    x.id = c.shot_ids[is];
    x.sh = &survey_.shots[x.id];
    x.nrec = x.sh->nrec;
    x.rec = rec_idx_ + rec_off_[x.id];
    x.sens = (sens_ && !x.sh->sens.empty()) ? sens_ + 3 * (size_t)rec_off_[x.id] : nullptr;
    x.stf_s = c.stf_rows.data() + (size_t)is * par_.nSteps;
    x.d_obs = (c.if_res && with_obs) ? obs_->acquire(x.id, x.nrec, c.st) : nullptr;
    x.scratch = c.with_adj && !par_.scratch_dir_name.empty();  // libCUFD.cu:732-752
    x.comps = (c.if_res || c.to_store) ? (x.scratch ? 9 : 8) : 15;
    // horizontal line of consecutive channels inside the computed region?
    const Shot &sh = *x.sh;
    bool is_line = par_.fiber == 0 && !x.sens && x.nrec > 0 && sh.z_rec[0] >= 2 && sh.z_rec[0] <= g.nzc - 3 && sh.x_rec[0] >= 3 && sh.x_rec[0] + x.nrec - 1 <= g.nx - 3;
    for (int r = 1; r < x.nrec && is_line; r++) is_line = (sh.z_rec[r] == sh.z_rec[0] && sh.x_rec[r] == sh.x_rec[0] + r);
    if (is_line) {
        x.line.z = sh.z_rec[0];
        x.line.x0 = sh.x_rec[0];
        x.line.n = x.nrec;
    }
    x.quiet = quiet_wanted(c, x) ? quiet_slot(lane) : nullptr;
    use_state(x, lane ? xl_[lane].state : state_);
    x.frame = lane ? xl_[lane].frame : frame_;
    x.syn = lane ? xl_[lane].syn : syn_;
    x.res = lane ? xl_[lane].res : res_;
    x.st = lane_st;
    return x;
}

// ---- forward pass of one shot (stream form) ------------------------------------------------------------------------------
// zero the 5 fields + 8 memory variables (libCUFD.cu:175-194); data column 0 stays 0 (:205-209)
void Session::forward_init(const ShotCtx &x) {
    HIP_OK(hipMemsetAsync(x.state, 0, 13 * cells_ * sizeof(float), x.st));
    if (x.quiet) {
        HIP_OK(hipMemsetAsync(x.quiet, 0, 2 * (size_t)g_.qn * sizeof(unsigned int), x.st));  // nothing holds a value yet
        quiet_last_ = x.quiet;
    }
    for (int k = 0; k < 4; k++)
        if ((x.comps >> k) & 1) HIP_OK(hipMemsetAsync(syn_of(x, k), 0, (size_t)x.nrec * sizeof(float), x.st));
}

// seismogram column `column` of the shot's present state (recording*, utilities.cu:593-602,645-703)
void Session::record_column(const ShotCtx &x, int column) {
    const size_t col = (size_t)column * x.nrec;
    launch_record(x.st, g_, x.fld, x.nrec, x.rec, syn_of(x, 0) + col, syn_of(x, 1) + col, syn_of(x, 2) + col, syn_of(x, 3) + col, x.comps, x.sens);
    launches_++;
}

// one forward time step (libCUFD.cu:268-332); inl: the line of channels is sampled inside k_stress
void Session::forward_step(const Call &c, const ShotCtx &x, int it, bool inl) {
    float *frame_t = c.with_adj ? x.frame + (size_t)it * 5 * (size_t)g_.frame_len : nullptr;
    const float amp = c.src_scale * x.stf_s[it] * par_.dt;
    LineRec lr{};
    if (inl && it >= 1) {
        lr = x.line;
        const size_t c0 = (size_t)it * x.nrec;
        lr.d_vx = (x.comps & 2) ? syn_of(x, 1) + c0 : nullptr;
        lr.d_vz = (x.comps & 4) ? syn_of(x, 2) + c0 : nullptr;
        lr.d_ett = (x.comps & 8) ? syn_of(x, 3) + c0 : nullptr;
    }
    launch_stress_fwd(x.st, g_, c.opt, x.fld, x.mem, md_, pc_, frame_t, x.sh->z_src, x.sh->x_src, amp, lr);
    launch_velocity_fwd(x.st, g_, c.opt, x.fld, x.mem, md_, pc_);
    launches_ += 2;
    if (!inl) record_column(x, it + 1);
}

// residual + misfit of the axial-strain component (libCUFD.cu:413,418,427)
void Session::residual(const ShotCtx &x) {
    launch_residual(x.st, x.d_obs, syn_of(x, 3), x.res, x.nrec, (long long)x.nrec * par_.nSteps, scal_);
    launches_++;
}

// the same with the data-conditioning chain (libCUFD.cu:353-457 as its commented lines compose it), on the MAIN stream:
// the scratch gathers and the FFT work space are shared by the shots of a call
void Session::residual_conditioned(const Call &c, const ShotCtx &x) {
    if (x.nrec <= 0) return;
    hipStream_t st = c.st;
    const int nSteps = par_.nSteps;
    const size_t tot = (size_t)rec_off_.back() + 1, off = (size_t)rec_off_[x.id];
    launch_transpose(st, syn_of(x, 3), xpose_, nSteps, x.nrec);  // [it][rec] -> [rec][it]
    condition_gather(st, xpose_, x.id, x.nrec);
    if (par_.if_src_update) cond_->source_update(st, x.d_obs, xpose_, x.nrec, par_.dt);  // libCUFD.cu:383-390
    if (par_.if_cross_misfit)
        cond_->cross_residual(st, x.d_obs, xpose_, xpose2_, x.nrec, win_ + 2 * tot + off, x.sh->src_weight, scal_);
    else
        cond_->l2_residual(st, x.d_obs, xpose_, xpose2_, x.nrec, scal_);
    if (par_.if_src_update) cond_->source_update_adj(st, xpose2_, x.nrec, par_.dt);  // libCUFD.cu:430-433
    if (par_.has_filter) cond_->bandpass(st, xpose2_, x.nrec, par_.dt, par_.filter);  // adjoint of the (zero-phase) filter
    if (par_.if_win)
        cond_->window(st, xpose2_, x.nrec, par_.dt, win_ + off, win_ + tot + off, win_ + 2 * tot + off, x.sh->src_weight, 0.005f);
    else
        cond_->window(st, xpose2_, x.nrec, par_.dt, nullptr, nullptr, nullptr, 1.0f, 0.005f);
    launch_transpose(st, xpose2_, x.res, x.nrec, nSteps);  // [rec][it] -> [it][rec]: the adjoint source
    launches_ += 8;
}

// ---- what a forward pass leaves behind -----------------------------------------------------------------------------------
// observe: export the four gathers as [nrec][nSteps] files (libCUFD.cu:755-769)
void Session::export_gathers(const Call &c, const ShotCtx &x) {
    hipStream_t st = c.st;
    const size_t cnt = (size_t)x.nrec * par_.nSteps;
    for (int k = 0; k < 4; k++) {
        launch_transpose(st, syn_of(x, k), xpose_, par_.nSteps, x.nrec);  // [it][rec] -> [rec][it]
        HIP_OK(hipMemcpyAsync(h_io_, xpose_, cnt * sizeof(float), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        const std::string fn = shot_file(par_, k, x.id);
        FILE *fp = fopen(fn.c_str(), "wb");
        if (!fp) throw IoError("cannot write '" + fn + "'");  // utilities.cu:22-31
        const size_t w = fwrite(h_io_, sizeof(float), cnt, fp);
        fclose(fp);
        if (w != cnt) throw IoError("short write on '" + fn + "'");
    }
    obs_->forget(x.id);  // a cached gather of this shot is stale: the file just changed
}

// optional scratch dumps of the PRESSURE component, [nrec][nSteps] float32 (libCUFD.cu:732-745): Syn_Shot{id}.bin,
// CondObs_Shot{id}.bin (observed data, unconditioned here as there) and Residual_Shot{id}.bin = obs - syn with the first time
// sample zeroed (gpuMinus, utilities.cu:154-167)
void Session::scratch_dumps(const Call &c, const ShotCtx &x) {
    hipStream_t st = c.st;
    const int nSteps = par_.nSteps;
    const size_t cnt = (size_t)x.nrec * nSteps;
    launch_transpose(st, syn_of(x, 0), xpose_, nSteps, x.nrec);
    HIP_OK(hipMemcpyAsync(h_io_, xpose_, cnt * sizeof(float), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    std::vector<float> obs_pr(cnt);
    {
        const std::string fn = shot_file(par_, 0, x.id);
        FILE *fp = fopen(fn.c_str(), "rb");
        if (!fp) throw IoError("cannot read observed data '" + fn + "'");
        const size_t got = fread(obs_pr.data(), sizeof(float), cnt, fp);
        fclose(fp);
        if (got != cnt) throw IoError("short read on '" + fn + "'");
    }
    auto dump = [&](const char *stem, const float *data) {
        const std::string fn = par_.scratch_dir_name + "/" + stem + std::to_string(x.id) + ".bin";
        FILE *fp = fopen(fn.c_str(), "wb");
        if (!fp) throw IoError("cannot write '" + fn + "'");
        const size_t w = fwrite(data, sizeof(float), cnt, fp);
        fclose(fp);
        if (w != cnt) throw IoError("short write on '" + fn + "'");
    };
    dump("Syn_Shot", h_io_);
    dump("CondObs_Shot", obs_pr.data());
    for (int r = 0; r < x.nrec; r++) {
        float *o = obs_pr.data() + (size_t)r * nSteps;
        const float *sy = h_io_ + (size_t)r * nSteps;
        o[0] = 0.0f;
        for (int t = 1; t < nSteps; t++) o[t] = o[t] - sy[t];
    }
    dump("Residual_Shot", obs_pr.data());
}

void Session::after_forward(Call &c, const ShotCtx &x) {
    if (c.to_store)  // calc_id 3: the modelled axial-strain gather becomes the shot's observed data, exactly as sepfwi_set_observed
        obs_->put_device_gather(x.id, syn_of(x, 3), x.nrec, c.st);  // would install the Shot_ett file of calc_id 2
    else if (!c.if_res)
        export_gathers(c, x);
    else if (x.scratch)
        scratch_dumps(c, x);
}

// ---- backward pass of one shot (stream form) -----------------------------------------------------------------------------
// adjoint fields + all eight memory variables restart from zero (:503-515); the two pre-loop adjoint launches (:520-542) act on
// all-zero arrays and change nothing.
void Session::backward_init(const BwdLane &L) {
    HIP_OK(hipMemsetAsync(L.bm.dvz_dz, 0, 8 * cells_ * sizeof(float), L.s));
    HIP_OK(hipMemsetAsync(L.adj.vz, 0, 5 * cells_ * sizeof(float), L.s));
}

// HIP-event pair for this step's k_bwd_b launch (option probe: every probe-th step), or null
hipEvent_t *Session::probe_pair(Call &c, int it) {
    if (c.opt.probe <= 0 || c.n_probe >= kProbePairs || (it % c.opt.probe) != 0) return nullptr;
    return &probe_ev_[2 * c.n_probe++];
}

void Session::collect_probes(Call &c) {  // after a synchronisation of the main stream
    for (int k = 0; k < c.n_probe; k++) {
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, probe_ev_[2 * k], probe_ev_[2 * k + 1]));
        probe_us_ += 1e3 * ms;
        probe_calls_++;
    }
    c.n_probe = 0;
}

// one backward time step, the reference's order (libCUFD.cu:545-631)
void Session::backward_step(Call &c, const ShotCtx &x, const BwdLane &L, int it) {
    const Grid &g = g_;
    const KernelOptions &opt = c.opt;
    const bool inj_inl = x.line.n > 0 && opt.line_fuse != 0;
    const Shot &sh = *x.sh;
    float *frame_t = x.frame + (size_t)it * 5 * (size_t)g.frame_len;
    float *sg = stf_grad_ + (size_t)x.is * par_.nSteps + it;
    const float amp = c.src_scale * x.stf_s[it] * par_.dt;
    const float *res_t = x.res + (size_t)it * x.nrec;
    LineRec lr{};
    if (inj_inl) {
        lr = x.line;
        lr.res = res_t;
    }
    Grid gs = g;  // this step's imaging weight (option img_every)
    if (opt.img_every > 1) gs.dt_img = (it % opt.img_every == 0) ? (float)opt.img_every * g.dt : 0.0f;
    if (opt.bwd_fuse != 0) {
        hipEvent_t *ev = probe_pair(c, it);
        Fields adj = L.adj;  // (the adjoint maps follow the shot's lane; the residual enters inside k_bwd_b, which marks the channels' segments)
        adj.q = x.quiet ? x.quiet + 2 * (size_t)g.qn : nullptr;
        launch_bwd_a(L.s, gs, opt, x.fld, L.bm, md_, pc_, frame_t, adj, L.acc);
        launch_bwd_b(L.s, gs, opt, x.fld, L.bm, md_, pc_, frame_t, sh.z_src, sh.x_src, amp, (float)sh.src_rxz, sg, adj, L.acc, lr, ev ? ev[0] : nullptr,
                     ev ? ev[1] : nullptr);
        if (!inj_inl) launch_inject(L.s, g, L.adj, x.nrec, x.rec, res_t, x.sens);
        launches_ += inj_inl ? 2 : 3;
    } else {  // the reference's launch structure
        launch_velocity_rev(L.s, gs, opt, x.fld, md_, pc_, frame_t, sh.z_src, sh.x_src, (float)sh.src_rxz, sg, L.adj, L.acc);
        launch_stress_rev(L.s, gs, opt, x.fld, md_, pc_, frame_t, sh.z_src, sh.x_src, amp, L.adj, L.acc);
        launch_velocity_adj(L.s, g, opt, L.adj, L.bm, md_, pc_);
        launch_inject(L.s, g, L.adj, x.nrec, x.rec, res_t, x.sens);
        launch_stress_adj(L.s, g, opt, L.adj, L.bm, md_, pc_);
        launches_ += 5;
    }
}

// The backward pass of one shot: ONE persistent launch where the configuration allows it (session_persist.cpp), else -- or when the
// loop's start rendezvous says the grid is not resident at once, which leaves everything untouched -- one backward_step per time step.
void Session::backward(Call &c, const ShotCtx &x) {
    hipStream_t st = c.st;
    const BwdLane L{st, mem_, adj_, acc_};
    const bool eligible = persist_ready(c, x);
    HIP_OK(hipEventRecord(ev_[2], st));
    backward_init(L);
    if (x.quiet) HIP_OK(hipMemsetAsync(x.quiet + 2 * (size_t)g_.qn, 0, 2 * (size_t)g_.qn * sizeof(unsigned int), st));  // the adjoint maps
    const bool looped = eligible && backward_persistent(c, x, L);
    if (!looped)
        for (int it = par_.nSteps - 2; it >= 0; it--) backward_step(c, x, L, it);
    HIP_OK(hipEventRecord(ev_[3], st));
    bwd_steps_ += (long long)(par_.nSteps - 1);
    HIP_OK(hipStreamSynchronize(st));
    collect_probes(c);
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev_[2], ev_[3]));
    bwd_ms_ += ms;
    if (looped) persist_check_pass(pk_);
}

// ---- stream schedule: up to fwd_lanes forward passes side by side (their kernel-boundary gaps and tails fill each other:
// x1.28 on the forward loops with three lanes), then their backward passes one after the other (two of them together lose
// 13-20 %)
void Session::run_streams(Call &c) {
    hipStream_t st = c.st;
    const int nSteps = par_.nSteps;
    int n_lanes = c.opt.pair_fwd ? c.opt.fwd_lanes : 1;
    n_lanes = std::max(1, std::min(std::min(n_lanes, c.group_size), (int)kMaxLanes));
    if (c.if_res) n_lanes = obs_->max_group((size_t)std::max(1, survey_.max_nrec) * nSteps * sizeof(float), n_lanes);
    if (n_lanes >= 2) ensure_lanes(n_lanes, c.with_adj);
    for (int is = 0; is < c.group_size;) {
        const int np = std::min(n_lanes, c.group_size - is);
        ShotCtx ctx[kMaxLanes];
        ctx[0] = make_ctx(c, is, 0, st);
        for (int k = 1; k < np; k++) ctx[k] = make_ctx(c, is + k, k, xl_[k].stream);

        // forward time loop(s), libCUFD.cu:268-332
        HIP_OK(hipEventRecord(ev_[0], st));
        for (int k = 1; k < np; k++) HIP_OK(hipStreamWaitEvent(xl_[k].stream, ev_[0], 0));  // extra lanes start after everything queued so far
        for (int k = 0; k < np; k++) forward_init(ctx[k]);
        bool inl[kMaxLanes];
        for (int k = 0; k < np; k++) inl[k] = forward_inline(c, ctx[k]);
        for (int it = 0; it <= nSteps - 2; it++)
            for (int k = 0; k < np; k++) forward_step(c, ctx[k], it, inl[k]);
        for (int k = 0; k < np; k++)
            if (inl[k]) record_column(ctx[k], nSteps - 1);
        if (c.if_res && !cond_on_)
            for (int k = 0; k < np; k++) residual(ctx[k]);
        for (int k = 1; k < np; k++) {  // join: the main stream continues when the extra lanes are done
            HIP_OK(hipEventRecord(xl_[k].join, xl_[k].stream));
            HIP_OK(hipStreamWaitEvent(st, xl_[k].join, 0));
        }
        if (c.if_res && cond_on_)
            for (int k = 0; k < np; k++) residual_conditioned(c, ctx[k]);
        HIP_OK(hipEventRecord(ev_[1], st));
        fwd_steps_ += (long long)np * (nSteps - 1);
        HIP_OK(hipStreamSynchronize(st));
        float ms = 0.f;
        HIP_OK(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
        fwd_ms_ += ms;
        obs_->release_all();  // the residuals are formed: the group's observed gathers may leave HBM again

        for (int k = 0; k < np; k++) after_forward(c, ctx[k]);
        if (c.with_adj)
            for (int k = 0; k < np; k++) backward(c, ctx[k]);
        is += np;
    }
}

// ---- outputs: written in place when they live on this device, staged otherwise (host memory, another GPU) ------------------
void Session::write_outputs(Call &c, float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf) {
    hipStream_t st = c.st;
    const size_t dense = (size_t)par_.nz * (size_t)par_.nx;
    if (c.with_adj && grad_stf) {  // rows indexed by local shot position (libCUFD.cu:671-673)
        std::vector<float> h_gstf((size_t)c.group_size * par_.nSteps);
        HIP_OK(hipMemcpy(h_gstf.data(), stf_grad_, h_gstf.size() * sizeof(float), hipMemcpyDeviceToHost));
        HIP_OK(hipMemcpy(grad_stf, h_gstf.data(), h_gstf.size() * sizeof(float), hipMemcpyDefault));
    }
    if (c.with_adj) {
        const bool devL = ptr_device(grad_Lambda) == gpu_id_, devM = ptr_device(grad_Mu) == gpu_id_, devD = ptr_device(grad_Den) == gpu_id_;
        float *oL = devL ? grad_Lambda : grad_stage_, *oM = devM ? grad_Mu : grad_stage_ + dense, *oD = devD ? grad_Den : grad_stage_ + 2 * dense;
        launch_finalize_gradients(st, g_, md_, acc_, oL, oM, oD);
        launches_++;
        if (!devL) HIP_OK(hipMemcpyAsync(grad_Lambda, oL, dense * sizeof(float), hipMemcpyDefault, st));
        if (!devM) HIP_OK(hipMemcpyAsync(grad_Mu, oM, dense * sizeof(float), hipMemcpyDefault, st));
        if (!devD) HIP_OK(hipMemcpyAsync(grad_Den, oD, dense * sizeof(float), hipMemcpyDefault, st));
    }
    if (c.if_res && misfit) {
        double sumsq = 0.0;
        HIP_OK(hipMemcpyAsync(&sumsq, scal_, sizeof(double), hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
        const float mf = (float)(0.5 * sumsq);  // libCUFD.cu:776
        HIP_OK(hipMemcpy(misfit, &mf, sizeof(float), hipMemcpyDefault));
    }
}

// ---- the cufd call -------------------------------------------------------------------------------------------------------
void Session::run(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf, const float *Lambda, const float *Mu,
                  const float *Den, const float *stf, int calc_id, int group_size, const int *shot_ids, hipStream_t ext_stream, bool async) {
    std::lock_guard<std::mutex> lock(mu_);
    const auto t_begin = std::chrono::steady_clock::now();
    HIP_OK(hipSetDevice(gpu_id_));
    Call c;
    c.opt = kernel_options();  // ONE snapshot for the whole call
    c.st = ext_stream ? ext_stream : own_stream_;
    if (!ext_stream) order_after_null_stream(c.st);
    c.if_res = (calc_id == 0 || calc_id == 1);  // Parameter.cpp:125-137
    c.with_adj = (calc_id == 1);
    c.to_store = (calc_id == SEPFWI_CALC_OBSERVE_TO_STORE);  // observe, but into the HBM store instead of the four files
    c.group_size = group_size;
    c.shot_ids = shot_ids;
    launches_ = 0;
    fwd_ms_ = bwd_ms_ = 0.0;
    probe_us_ = 0.0;
    probe_calls_ = 0;
    fwd_steps_ = bwd_steps_ = persist_steps_ = 0;
    quiet_active_ = quiet_total_ = 0;
    quiet_last_ = nullptr;
    for (int i = 0; i < group_size; i++) {
        const int id = shot_ids[i];
        if (id < 0 || id >= (int)survey_.shots.size() || !survey_.shots[id].present)
            throw std::invalid_argument("shot id " + std::to_string(id) + " is not in the survey file");
    }
    // HBM budget of the observed-data store: parameter key "obs_cache_mb", else the option of the same name (0: unlimited)
    const long long mb = par_.obs_cache_mb > 0 ? par_.obs_cache_mb : c.opt.obs_cache_mb;
    obs_->set_budget_bytes(mb * 1000000LL);
    obs_->release_all();

    prepare_media(c, Lambda, Mu, Den);
    prepare_buffers(c, stf);
    const size_t gather_bytes = (size_t)std::max(1, survey_.max_nrec) * par_.nSteps * sizeof(float);
    if (c.if_res && obs_->budget_bytes() == 0)  // observed data of every shot of the call resident before the time loops start
        for (int is = 0; is < group_size; is++) (void)obs_->acquire(shot_ids[is], survey_.shots[shot_ids[is]].nrec, c.st);
    obs_->release_all();

    // Batch sizes from the Infinity-Cache budget: a forward batch keeps 5 fields per shot + 5 media arrays resident, a backward
    // batch 15 arrays per shot + 5 (2000x500: 7 and 2; a 101x201 notebook problem: all its shots at once).  Where fewer than three
    // backward passes fit (two-launch step: two) the stream schedule runs them one by one -- as the persistent loop where it is
    // eligible: measured fwd+adj at 1000 steps, batched / streams in Gcell-updates/s: 2000x500 (2 fit) 73.5 / 79.6, 1500x500 (3) 77.1 /
    // 75.0, 1000x700 (3) 73.2 / 72.5, 2000x300 (4) 73.7 / 66.1, 1000x500 (5) 69.0 / 63.9 (profiles/r05_other_grids.txt).
    const double arr_mb = (double)cells_ * sizeof(float) / 1.0e6, budget = (double)c.opt.batch_mb;
    int Bf = (int)((budget / arr_mb - 5.0) / 5.0), Bb = (int)((budget / arr_mb - 5.0) / 15.0);
    const int bb_min = c.opt.bwd_fuse == 4 ? 3 : 2;
    const bool batched = c.opt.bwd_fuse != 0 && group_size >= 1 &&
                         (c.opt.batch == 1 || (c.opt.batch == 2 && (c.with_adj ? Bb >= bb_min : Bf >= 8)));  // forward-only calls: streams until kernels are launch-bound
    last_batched_ = batched;
    if (batched) {
        if (c.opt.batch_f > 0) Bf = c.opt.batch_f;
        if (c.opt.batch_b > 0) Bb = c.opt.batch_b;
        Bf = std::max(1, std::min(std::min(Bf, 32), group_size));
        if (c.if_res) Bf = obs_->max_group(gather_bytes, Bf);
        Bb = std::max(1, std::min(Bb, Bf));
        if (!c.opt.pair_fwd) Bf = Bb = 1;
        run_batched(c, Bf, Bb);
    } else {
        run_streams(c);
    }
    write_outputs(c, misfit, grad_Lambda, grad_Mu, grad_Den, grad_stf);
    if (!async) {
        HIP_OK(hipStreamSynchronize(c.st));
    } else if (!ext_stream) {  // later work on the default stream sees this call's outputs
        HIP_OK(hipEventRecord(ev_order_, c.st));
        HIP_OK(hipStreamWaitEvent(nullptr, ev_order_, 0));
    }
    if (quiet_last_) {  // how much of the grid the last shot's forward field reached (sepfwi_stats)
        std::vector<unsigned int> bits((size_t)g_.qn);
        HIP_OK(hipMemcpyAsync(bits.data(), quiet_last_ + g_.qn, bits.size() * sizeof(unsigned int), hipMemcpyDeviceToHost, c.st));
        HIP_OK(hipStreamSynchronize(c.st));
        for (unsigned int w : bits) quiet_active_ += __builtin_popcount(w);
        quiet_total_ = (long long)(g_.nzc - 4) * ((g_.nx + 63) / 64);
    }
    total_ms_ = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count();
    last_shots_ = group_size;
    last_calc_ = calc_id;
}

}  // namespace sepfwi
