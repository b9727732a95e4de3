// session.cpp -- persistent per-(parameter file, GPU) propagation session and the cufd driver.
//
// Replaces the per-call host driver of the reference, Src/libCUFD.cu:32-820 (set-up :39-165, shot loop
// :170-708, gradient read-back :710-724, seismogram files :755-769) and the classes it instantiates on
// every call: Model (Src/Model.cu), Cpml (Src/Cpml.cu), Bnd (Src/Boundary.cu), Src_Rec (Src/Src_Rec.cu).
// Differences by design: device state is allocated once and kept; observed data are cached
// in HBM (time-major) instead of being re-read from four files per shot per call; only the axial-strain
// (ett) residual -- the only one that enters misfit and adjoint source (libCUFD.cu:427,607) -- is formed.
#include "session.hpp"

#include <sys/stat.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

#include "device_alloc.hpp"
#include "hip_check.hpp"
#include "host_checks.hpp"
#include "kernels.hpp"

namespace sepfwi {

template <class T>
T *Session::dalloc(size_t n) {
    void *p = nullptr;
    HIP_OK(dev_malloc(&p, n * sizeof(T)));
    allocs_.push_back(p);
    device_bytes_ += (long long)(n * sizeof(T));
    return (T *)p;
}

// grid geometry of the session from the parameter file (Parameter.cpp:41-178, Boundary.cu:17-27)
void Session::init_grid() {
    const Params &par = par_;
    Grid &g = g_;
    g.nz = par.nz;
    g.nx = par.nx;
    g.nzc = par.nz - par.nPad;
    g.pitch = ((par.nx + 63) / 64) * 64;
    g.fiber = par.fiber;
    g.nPml = par.nPml;
    g.zmax = g.nzc - 1 - par.nPml;
    g.xmax = par.nx - 1 - par.nPml;
    g.nSteps = par.nSteps;
    g.dt = par.dt;
    g.dt_img = par.dt;
    g.dz = par.dz;
    g.dx = par.dx;
    g.rdz = 1.0f / par.dz;
    g.rdx = 1.0f / par.dx;
    g.nzBnd = g.nzc - 2 * par.nPml + 4;  // Boundary.cu:17-18
    g.nxBnd = par.nx - 2 * par.nPml + 4;
    g.frame_len = 10 * g.nxBnd + 10 * (g.nzBnd - 10);
    g.qzw = ((g.nzc + 4) >> 5) + 2;          // quiet-segment maps (Fields::q): bits z + 2 of rows -2 ... nzc + 1, one spare word for the 64-bit window
    g.qn = (g.pitch / 64 + 2) * g.qzw;        // segment columns -1 ... pitch / 64

}

// [5 fields | 8 C-PML memories | 5 adjoint fields] contiguous (one memset clears a group; the fused kernels take them as bundles),
// media, accumulators, staging for foreign-memory inputs / outputs (libCUFD.cu:127-146)
void Session::alloc_arrays() {
    const Params &par = par_;
    const Grid &g = g_;
    const size_t n = (size_t)(g.nzc + 4) * (size_t)g.pitch;  // 4 spare rows
    cells_ = n;
    // [5 fields | 8 pml memories | 5 adjoint fields] contiguous so one memset clears a group
    state_ = dalloc<float>(18 * n);
    float *s = state_;
    fld_ = Fields{s, s + n, s + 2 * n, s + 3 * n, s + 4 * n};
    mem_ = PmlMem{s + 5 * n, s + 6 * n, s + 7 * n, s + 8 * n, s + 9 * n, s + 10 * n, s + 11 * n, s + 12 * n};
    adj_ = Fields{s + 13 * n, s + 14 * n, s + 15 * n, s + 16 * n, s + 17 * n};
    media_ = dalloc<float>(6 * n);
    HIP_OK(hipMemset(media_, 0, 6 * n * sizeof(float)));
    HIP_OK(hipDeviceSynchronize());  // the fill runs on the null stream and does not block the host; a caller's non-blocking stream would not wait for it
    md_ = Media{media_, media_ + n, media_ + 2 * n, media_ + 3 * n, media_ + 4 * n, media_ + 5 * n};
    acc_buf_ = dalloc<float>(5 * n);
    quiet_pool_ = dalloc<unsigned int>((size_t)kQuietSlots * 4 * (size_t)g.qn);
    acc_ = ImgAcc{acc_buf_, acc_buf_ + n, acc_buf_ + 2 * n, acc_buf_ + 3 * n, acc_buf_ + 4 * n};
    const size_t dense = (size_t)par.nz * (size_t)par.nx;
    in_stage_ = dalloc<float>(3 * dense);
    grad_stage_ = dalloc<float>(3 * dense);
    scal_ = dalloc<double>(4);
    cp2_bits_ = dalloc<unsigned int>(4);

}

// C-PML profiles (host) -> device, with 1/K precomputed (Cpml.cu:7-117, utilities.cu:243-359)
void Session::upload_profiles() {
    const Params &par = par_;
    const Grid &g = g_;
    {
        const int nzc = g.nzc, nx = g.nx;
        std::vector<float> K(std::max(nzc, nx)), a(K.size()), b(K.size()), Kh(K.size()), ah(K.size()), bh(K.size());
        std::vector<float> hz(6 * (size_t)nzc), hx(6 * (size_t)nx);
        cpml_profiles(K.data(), a.data(), b.data(), Kh.data(), ah.data(), bh.data(), nzc, par.nPml, par.dz, par.f0, par.dt);
        for (int i = 0; i < nzc; i++) {
            hz[i] = a[i]; hz[nzc + i] = b[i]; hz[2 * nzc + i] = 1.0f / K[i];
            hz[3 * nzc + i] = ah[i]; hz[4 * nzc + i] = bh[i]; hz[5 * nzc + i] = 1.0f / Kh[i];
        }
        cpml_profiles(K.data(), a.data(), b.data(), Kh.data(), ah.data(), bh.data(), nx, par.nPml, par.dx, par.f0, par.dt);
        for (int i = 0; i < nx; i++) {
            hx[i] = a[i]; hx[nx + i] = b[i]; hx[2 * nx + i] = 1.0f / K[i];
            hx[3 * nx + i] = ah[i]; hx[4 * nx + i] = bh[i]; hx[5 * nx + i] = 1.0f / Kh[i];
        }
        float *dz_ = dalloc<float>(hz.size() + hx.size()), *dx_ = dz_ + hz.size();  // contiguous: kernels may address x profiles as z base + 6*nzc
        HIP_OK(hipMemcpy(dz_, hz.data(), hz.size() * sizeof(float), hipMemcpyHostToDevice));
        HIP_OK(hipMemcpy(dx_, hx.data(), hx.size() * sizeof(float), hipMemcpyHostToDevice));
        pc_ = PmlCoef{dz_, dz_ + nzc, dz_ + 2 * nzc, dz_ + 3 * nzc, dz_ + 4 * nzc, dz_ + 5 * nzc,
                      dx_, dx_ + nx,  dx_ + 2 * nx,  dx_ + 3 * nx,  dx_ + 4 * nx,  dx_ + 5 * nx};
    }

}

// receivers: flat cell index per shot (validated against the grid: host_checks.cpp), directional sensitivities, per-channel windows
void Session::upload_survey() {
    const Grid &g = g_;
    {
        const int ns = (int)survey_.shots.size();
        std::vector<int> idx;
        receiver_cells(par_, survey_, g.nzc, g.nx, g.pitch, &rec_off_, &idx);
        rec_idx_ = dalloc<int>(idx.size());
        HIP_OK(hipMemcpy(rec_idx_, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice));
        bool any_sens = false;
        for (int i = 0; i < ns; i++) any_sens = any_sens || (survey_.shots[i].present && !survey_.shots[i].sens.empty());
        if (any_sens) {  // (s_xx, s_zz, s_xz) per channel, same offsets as rec_idx_
            std::vector<float> sv(3 * idx.size(), 0.0f);
            for (int i = 0; i < ns; i++) {
                const Shot &sh = survey_.shots[i];
                if (sh.present && !sh.sens.empty()) std::copy(sh.sens.begin(), sh.sens.end(), sv.begin() + 3 * (size_t)rec_off_[i]);
            }
            sens_ = dalloc<float>(sv.size());
            HIP_OK(hipMemcpy(sens_, sv.data(), sv.size() * sizeof(float), hipMemcpyHostToDevice));
        }
    }
    if (cond_on_) {  // [start | end | weight] per channel; without if_win only the weights matter (cross-correlation misfit)
        const int ns = (int)survey_.shots.size();
        const size_t tot = (size_t)rec_off_[ns] + 1;
        std::vector<float> w(3 * tot, 0.0f);
        for (int i = 0; i < ns; i++) {
            const Shot &sh = survey_.shots[i];
            if (!sh.present) continue;
            for (int r = 0; r < sh.nrec; r++) {
                const size_t k = (size_t)rec_off_[i] + r;
                w[k] = sh.win_start.empty() ? 0.0f : sh.win_start[r];
                w[tot + k] = sh.win_end.empty() ? 0.0f : sh.win_end[r];
                w[2 * tot + k] = sh.weights.empty() ? 1.0f : sh.weights[r];
            }
        }
        win_ = dalloc<float>(w.size());
        HIP_OK(hipMemcpy(win_, w.data(), w.size() * sizeof(float), hipMemcpyHostToDevice));
    }
}

Session::Session(const std::string &para_fname, int gpu_id, const std::string &para_text,
                 const std::string &survey_text, const Params &par, const Survey &survey)
    : para_fname_(para_fname), gpu_id_(gpu_id), para_text_(para_text), survey_text_(survey_text), par_(par),
      survey_(survey) {
    // The data-conditioning keys are dormant in the reference (every call site is commented out, libCUFD.cu:353-457; the one live
    // line, source_update_adj at :430-433, acts on the pressure residual that is never injected).  Here a key switches its
    // stage on for the axial-strain gathers (conditioning.hip).  One combination has no defined meaning there either: the
    // commented lines take the trace norms of the cross-correlation misfit BEFORE the source update and use them after it.
    if (par.if_src_update && par.if_cross_misfit)
        throw std::invalid_argument("parameter file: if_src_update together with if_cross_misfit is not supported");
    cond_on_ = par.if_win || par.has_filter || par.if_cross_misfit || par.if_src_update;
    HIP_OK(hipSetDevice(gpu_id_));
    HIP_OK(hipStreamCreateWithFlags(&own_stream_, hipStreamNonBlocking));
    HIP_OK(hipEventCreateWithFlags(&ev_order_, hipEventDisableTiming));
    for (auto &e : ev_) HIP_OK(hipEventCreate(&e));
    for (auto &e : probe_ev_) HIP_OK(hipEventCreate(&e));

    init_grid();
    alloc_arrays();
    upload_profiles();
    upload_survey();
    const size_t dlen = (size_t)std::max(1, survey_.max_nrec) * (size_t)par.nSteps;
    data_len_ = dlen;
    syn_ = dalloc<float>(4 * dlen);  // time-major pr, vx, vz, ett
    res_ = dalloc<float>(dlen);
    xpose_ = dalloc<float>(dlen);
    if (cond_on_) {
        xpose2_ = dalloc<float>(dlen);
        cond_.reset(new Conditioner(par.nSteps, std::max(1, survey_.max_nrec)));
        if (par.if_src_update) cond_->ensure_source_buffers(own_stream_);  // now, so that sepfwi_stats.device_bytes counts them
        device_bytes_ += cond_->device_bytes();
    }
    HIP_OK(hipHostMalloc((void **)&h_io_, dlen * sizeof(float), hipHostMallocDefault));
    {
        ObservedStore::Host h;
        h.gpu_id = gpu_id_;
        h.par = &par_;
        h.survey = &survey_;
        h.xpose = xpose_;
        h.h_io = h_io_;
        h.cond_on = cond_on_;
        h.condition = [this](hipStream_t st, float *gather, int shot_id, int nrec) { condition_gather(st, gather, shot_id, nrec); };
        obs_.reset(new ObservedStore(h));
    }
    // everything the constructor put on the null stream (fills, profile / receiver tables copied from pageable host memory) is
    // complete before any stream of a later call -- the session's own non-blocking ones or a caller's -- can touch it
    HIP_OK(hipDeviceSynchronize());
}

Session::~Session() {
    (void)hipSetDevice(gpu_id_);
    (void)hipDeviceSynchronize();
    for (float *p : {ba_.state, ba_.syn, ba_.res, ba_.frame, ba_.bwd})
        if (p) (void)hipFree(p);
    if (d_shots_) (void)hipFree(d_shots_);
    if (d_stf_) (void)hipFree(d_stf_);
    for (XLane &L : xl_) {
        if (L.state) (void)hipFree(L.state);
        if (L.frame) (void)hipFree(L.frame);
        if (L.syn) (void)hipFree(L.syn);
        if (L.res) (void)hipFree(L.res);
        if (L.stream) (void)hipStreamDestroy(L.stream);
        if (L.join) (void)hipEventDestroy(L.join);
    }
    obs_.reset();
    for (Persist *k : {&pk_, &pk_ms_}) {
        if (k->d_seg) (void)hipFree(k->d_seg);
        if (k->d_hdr) (void)hipFree(k->d_hdr);
        if (k->d_sync) (void)hipFree(k->d_sync);
        if (k->d_qnbr) (void)hipFree(k->d_qnbr);
        if (k->d_stf) (void)hipFree(k->d_stf);
        if (k->h_err) (void)hipHostFree(k->h_err);
    }
    for (auto &kv : inj_) {
        InjDev &d = kv.second;
        (void)hipFree(d.lookup);
        (void)hipFree(d.segs);
        (void)hipFree(d.tgt_start);
        (void)hipFree(d.ent_rec);
        (void)hipFree(d.ent_w);
        if (d.tile_has) (void)hipFree(d.tile_has);
        if (d.d_args) (void)hipFree(d.d_args);
    }
    if (inj_val_) (void)hipFree(inj_val_);
    for (void *p : allocs_) (void)hipFree(p);
    if (frame_) (void)hipFree(frame_);
    if (stf_grad_) (void)hipFree(stf_grad_);
    if (h_io_) (void)hipHostFree(h_io_);
    for (auto &e : ev_) (void)hipEventDestroy(e);
    for (auto &e : probe_ev_) (void)hipEventDestroy(e);
    if (ev_order_) (void)hipEventDestroy(ev_order_);
    if (own_stream_) (void)hipStreamDestroy(own_stream_);
}

// Extra lanes of forward state (fields, memory variables, boundary frames, seismograms, residual) and their streams.
void Session::ensure_lanes(int n_lanes, bool with_frames) {
    const size_t n = cells_;
    for (int k = 1; k < n_lanes && k < kMaxLanes; k++) {
        XLane &L = xl_[k];
        if (!L.stream) {
            HIP_OK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&L.join, hipEventDisableTiming));
        }
        if (!L.state) {
            HIP_OK(dev_malloc((void **)&L.state, 13 * n * sizeof(float)));
            HIP_OK(dev_malloc((void **)&L.syn, 4 * data_len_ * sizeof(float)));
            HIP_OK(dev_malloc((void **)&L.res, data_len_ * sizeof(float)));
            device_bytes_ += (long long)((13 * n + 5 * data_len_) * sizeof(float));
        }
        if (with_frames && !L.frame) {
            const size_t fb = (size_t)par_.nSteps * 5 * (size_t)g_.frame_len * sizeof(float);
            HIP_OK(dev_malloc((void **)&L.frame, fb));
            device_bytes_ += (long long)fb;
        }
    }
}

// Batched mode: n_fwd lanes of forward state, the first n_bwd of them with backward state too; shot table and source rows
// for n_shots shots.
// Lanes of the batched schedule.  The lanes of one kind lie in ONE arena at a constant stride (the multi-shot persistent loop
// addresses shot k of a launch as base + k stride, fwi_types.hpp MultiShot); an arena that is too small is replaced -- lanes carry
// nothing from one call to the next.
void Session::ensure_batch(int n_fwd, int n_bwd, bool with_frames, int n_shots) {
    const size_t n = cells_;
    auto regrow = [&](float *&arena, int &cap, int want, size_t per_lane) {
        if (want <= cap) return;
        if (arena) {
            (void)hipFree(arena);
            device_bytes_ -= (long long)((size_t)cap * per_lane * sizeof(float));
        }
        arena = nullptr;
        cap = 0;
        HIP_OK(dev_malloc((void **)&arena, (size_t)want * per_lane * sizeof(float)));
        cap = want;
        device_bytes_ += (long long)((size_t)want * per_lane * sizeof(float));
    };
    const size_t frame_lane = (size_t)par_.nSteps * 5 * (size_t)g_.frame_len;
    regrow(ba_.state, ba_.n_state, n_fwd, 13 * n);
    regrow(ba_.syn, ba_.n_syn, n_fwd, 4 * data_len_);
    regrow(ba_.res, ba_.n_res, n_fwd, data_len_);
    if (with_frames) regrow(ba_.frame, ba_.n_frame, n_fwd, frame_lane);
    regrow(ba_.bwd, ba_.n_bwd, n_bwd, 18 * n);
    bl_.assign((size_t)std::max(ba_.n_state, 1), BLane{});
    for (int k = 0; k < ba_.n_state; k++) {
        BLane &L = bl_[k];
        L.state = ba_.state + (size_t)k * 13 * n;
        L.syn = ba_.syn + (size_t)k * 4 * data_len_;
        L.res = ba_.res + (size_t)k * data_len_;
        L.frame = k < ba_.n_frame ? ba_.frame + (size_t)k * frame_lane : nullptr;
        L.bwd = k < ba_.n_bwd ? ba_.bwd + (size_t)k * 18 * n : nullptr;
    }
    if (n_shots > shots_cap_) {
        if (d_shots_) (void)hipFree(d_shots_);
        d_shots_ = nullptr;
        HIP_OK(dev_malloc((void **)&d_shots_, (size_t)n_shots * sizeof(ShotDev)));
        shots_cap_ = n_shots;
    }
    const size_t need = (size_t)n_shots * par_.nSteps;
    if (need > d_stf_len_) {
        if (d_stf_) (void)hipFree(d_stf_);
        d_stf_ = nullptr;
        HIP_OK(dev_malloc((void **)&d_stf_, need * sizeof(float)));
        d_stf_len_ = need;
    }
}

void Session::drop_observed() {
    std::lock_guard<std::mutex> lock(mu_);
    (void)hipSetDevice(gpu_id_);
    obs_->clear();
}

// Observed axial-strain gather of one shot handed over from memory ([nrec][nSteps], host or device pointer).
void Session::set_observed(int shot_id, const float *ett, int nrec, int nSteps) {
    std::lock_guard<std::mutex> lock(mu_);
    HIP_OK(hipSetDevice(gpu_id_));
    if (shot_id < 0 || shot_id >= (int)survey_.shots.size() || !survey_.shots[shot_id].present)
        throw std::invalid_argument("set_observed: unknown shot id " + std::to_string(shot_id));
    if (!ett || nrec != survey_.shots[shot_id].nrec || nSteps != par_.nSteps)
        throw std::invalid_argument("set_observed: data must be [nrec][nSteps] of the survey / parameter file");
    if (nrec > 0) order_after_null_stream(own_stream_);  // a HIP `ett` was produced on the caller's (default) stream
    obs_->release_all();
    obs_->put(shot_id, ett, nrec, own_stream_);
}

// Window and band-pass one [rec][it] gather in place, as the commented driver lines apply them to observed and synthetic
// data alike (libCUFD.cu:353-374): per-channel windows with weights when if_win, else the plain end taper; then the filter.
void Session::condition_gather(hipStream_t st, float *gather, int shot_id, int nrec) {
    const size_t tot = (size_t)rec_off_.back() + 1, off = (size_t)rec_off_[shot_id];
    const Shot &sh = survey_.shots[shot_id];
    if (par_.if_win)
        cond_->window(st, gather, nrec, par_.dt, win_ + off, win_ + tot + off, win_ + 2 * tot + off, sh.src_weight, 0.005f);
    else
        cond_->window(st, gather, nrec, par_.dt, nullptr, nullptr, nullptr, 1.0f, 0.005f);
    if (par_.has_filter) cond_->bandpass(st, gather, nrec, par_.dt, par_.filter);
}

// Test hook (sepfwi_debug_field): one wavefield of one forward lane as the last call left it, dense (nzc, nx).
void Session::copy_field(int lane, int which, float *out) {
    std::lock_guard<std::mutex> lock(mu_);
    HIP_OK(hipSetDevice(gpu_id_));
    if (!out || which < 0 || which > 9) throw std::invalid_argument("debug_field: which must be 0..9");
    const float *base = nullptr;
    if (which >= 5) {  // adjoint fields: one set per session (stream mode) or per backward lane (batched mode)
        if (last_batched_) {
            if (lane < 0 || lane >= (int)bl_.size() || !bl_[lane].bwd) throw std::invalid_argument("debug_field: no such backward lane");
            base = bl_[lane].bwd + (8 + (which - 5)) * cells_;
        } else {
            base = adj_.vz + (size_t)(which - 5) * cells_;
        }
    } else if (last_batched_) {
        if (lane < 0 || lane >= (int)bl_.size() || !bl_[lane].state) throw std::invalid_argument("debug_field: no such lane");
        base = bl_[lane].state + (size_t)which * cells_;
    } else {
        if (lane < 0 || lane >= kMaxLanes || (lane > 0 && !xl_[lane].state)) throw std::invalid_argument("debug_field: no such lane");
        base = (lane ? xl_[lane].state : state_) + (size_t)which * cells_;
    }
    HIP_OK(hipMemcpy2D(out, (size_t)g_.nx * sizeof(float), base, (size_t)g_.pitch * sizeof(float), (size_t)g_.nx * sizeof(float),
                       (size_t)g_.nzc, hipMemcpyDefault));
}

void Session::stats(sepfwi_stats *out) const {
    out->fwd_ms = fwd_ms_;
    out->bwd_ms = bwd_ms_;
    out->total_ms = total_ms_;
    out->n_c = g_.nzc * g_.nx;
    out->fwd_steps = fwd_steps_;
    out->bwd_steps = bwd_steps_;
    out->launches = launches_;
    out->device_bytes = device_bytes_ + obs_->device_bytes();
    out->obs_device_bytes = obs_->device_bytes();
    out->obs_host_bytes = obs_->host_bytes();
    out->obs_evictions = obs_->evictions();
    out->persist_steps = persist_steps_;
    out->quiet_active = quiet_active_;
    out->quiet_total = quiet_total_;
    out->probe_kernel_us = probe_calls_ ? probe_us_ / (double)probe_calls_ : 0.0;
    out->probe_calls = probe_calls_;
    // one forward pass = N_c*(nSteps-1); fwd+adj = 3x (forward, reconstruction, adjoint)
    out->cell_updates = (double)out->n_c * ((double)fwd_steps_ + 2.0 * (double)bwd_steps_);
}

// ------------------------------------------------------------------------------------------------
// registry
// ------------------------------------------------------------------------------------------------
static std::mutex g_reg_mu;
static std::map<std::pair<std::string, int>, std::shared_ptr<Session>> g_sessions;

std::shared_ptr<Session> get_session(const std::string &para_fname, int gpu_id) {
    const std::string ptext = read_first_line(para_fname);
    Params par = parse_params(ptext);
    const std::string stext = read_first_line(par.survey_fname);
    std::lock_guard<std::mutex> lock(g_reg_mu);
    auto key = std::make_pair(para_fname, gpu_id);
    auto it = g_sessions.find(key);
    if (it != g_sessions.end() && it->second->matches(ptext, stext)) return it->second;
    if (it != g_sessions.end()) g_sessions.erase(it);  // a thread still inside run() keeps its own reference
    Survey sv = parse_survey(stext, par.nPml, par.if_win_key);
    int ndev = 0;
    HIP_OK(hipGetDeviceCount(&ndev));
    if (gpu_id < 0 || gpu_id >= ndev)
        throw HipError("gpu_id " + std::to_string(gpu_id) + " out of range: " + std::to_string(ndev) + " HIP device(s) visible");
    auto sp = std::make_shared<Session>(para_fname, gpu_id, ptext, stext, par, sv);
    g_sessions[key] = sp;
    return sp;
}

std::shared_ptr<Session> find_session(const std::string &para_fname, int gpu_id) {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    auto it = g_sessions.find(std::make_pair(para_fname, gpu_id));
    return it == g_sessions.end() ? nullptr : it->second;
}

void release_all_sessions() {
    std::lock_guard<std::mutex> lock(g_reg_mu);
    g_sessions.clear();
}

void invalidate_observed_all() {
    std::vector<std::shared_ptr<Session>> all;
    {
        std::lock_guard<std::mutex> lock(g_reg_mu);
        for (auto &kv : g_sessions) all.push_back(kv.second);
    }
    for (auto &sp : all) sp->drop_observed();
}

}  // namespace sepfwi
