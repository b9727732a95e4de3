// session_batched.cpp -- the batched schedule of a call's shots: every launch advances a whole batch of shots by a
// half step (grids that are not the headline's are launch-bound: the reference issues 24 launches per shot and time step,
// Src/libCUFD.cu:268-332,545-631).  Bf shots share a forward launch, Bb <= Bf a backward launch; per-shot pointers and scalars in a
// device table (ShotDev).
//   batch_ctx / batch_table     a shot in its batch lane; the device table of the call's shots
//   batch_streams               a batch as sub-batches on streams of their own (launches of different queues overlap fill and drain)
//   batched_forward             forward time loop of one batch (+ residuals)
//   batched_backward            backward time loop of one sub-batch
//   run_batched                 the schedule
#include <algorithm>

#include "device_alloc.hpp"
#include "hip_check.hpp"
#include "kernels.hpp"
#include "session.hpp"

namespace sepfwi {

// shot `is` of the call in its batch lane
Session::ShotCtx Session::batch_ctx(const Call &c, int is, int Bf, bool with_obs) {
    ShotCtx x = make_ctx(c, is, 0, c.st, with_obs);
    const BLane &L = bl_[is % Bf];
    if (x.quiet) x.quiet = quiet_slot(kMaxLanes + is % Bf);
    use_state(x, L.state);
    x.frame = L.frame;
    x.syn = L.syn;
    x.res = L.res;
    return x;
}

// the device table of the call's shots (uploaded; the host copy tells the schedule which shots have a fused line of channels)
std::vector<ShotDev> Session::batch_table(const Call &c, int Bf, int Bb) {
    const int nSteps = par_.nSteps;
    const size_t n = cells_;
    const bool lf = c.opt.line_fuse != 0;
    std::vector<ShotDev> tab(c.group_size);
    for (int is = 0; is < c.group_size; is++) {
        const ShotCtx x = batch_ctx(c, is, Bf, false);
        const BLane &LB = bl_[(is % Bf) % Bb];  // backward lane of this shot inside its sub-batch
        ShotDev &d = tab[is];
        d.fields = x.state;
        d.mem = x.state + 5 * n;
        d.frame = x.frame;
        d.syn = x.syn;
        d.stf = d_stf_ + (size_t)is * nSteps;
        d.bmem = c.with_adj ? LB.bwd : nullptr;
        d.adj = c.with_adj ? LB.bwd + 8 * n : nullptr;
        d.acc = c.with_adj ? LB.bwd + 13 * n : nullptr;
        d.res = x.res;
        d.stf_grad = c.with_adj ? stf_grad_ + (size_t)is * nSteps : nullptr;
        d.z_src = x.sh->z_src;
        d.x_src = x.sh->x_src;
        d.lr_z = x.line.z;
        d.lr_x0 = x.line.x0;
        d.lr_n = lf ? x.line.n : 0;
        d.comps = x.comps | ((lf && x.line.n > 0 && !(x.comps & 1)) ? 16 : 0);  // bit 16: sample the line inside k_stress
        d.nrec = x.nrec;
        d.src_rxz = (float)x.sh->src_rxz;
        d.quiet = x.quiet;
        d.rec = x.rec;
        d.sens = x.sens;
    }
    HIP_OK(hipMemcpyAsync(d_shots_, tab.data(), tab.size() * sizeof(ShotDev), hipMemcpyHostToDevice, c.st));
    HIP_OK(hipStreamSynchronize(c.st));  // `tab` and `stf_rows` are pageable host memory
    return tab;
}

// sub[0] = the call's stream, sub[1 .. ns-1] = the extra lanes' streams, which start after everything queued on the call's so far
void Session::batch_streams(hipStream_t st, int ns, hipStream_t *sub) {
    sub[0] = st;
    if (ns <= 1) return;
    for (int q = 1; q < ns; q++) {
        XLane &L = xl_[q];
        if (!L.stream) {
            HIP_OK(hipStreamCreateWithFlags(&L.stream, hipStreamNonBlocking));
            HIP_OK(hipEventCreateWithFlags(&L.join, hipEventDisableTiming));
        }
        sub[q] = L.stream;
    }
    HIP_OK(hipEventRecord(ev_order_, st));
    for (int q = 1; q < ns; q++) HIP_OK(hipStreamWaitEvent(sub[q], ev_order_, 0));
}

void Session::batch_join(hipStream_t st, int ns) {  // the call's stream continues when the extra lanes are done
    for (int q = 1; q < ns; q++) {
        HIP_OK(hipEventRecord(xl_[q].join, xl_[q].stream));
        HIP_OK(hipStreamWaitEvent(st, xl_[q].join, 0));
    }
}

// forward time loop of the batch tab[is0 .. is0 + nb), libCUFD.cu:268-332, and its residuals
void Session::batched_forward(Call &c, const std::vector<ShotDev> &tab, int is0, int nb, const std::vector<ShotCtx> &cx) {
    hipStream_t st = c.st;
    const KernelOptions &opt = c.opt;
    const int nSteps = par_.nSteps;
    HIP_OK(hipEventRecord(ev_[0], st));
    for (int k = 0; k < nb; k++) forward_init(cx[k]);
    // the batch as up to three sub-batches on streams of their own (option batch_split)
    const int ns = std::max(1, std::min(std::min(opt.batch_split, (int)kMaxLanes - 1), nb));
    auto general = [&](int a0, int a1) {  // a shot in [a0, a1) whose channels are not sampled inside k_stress?
        for (int k = a0; k < a1; k++)
            if (!(tab[is0 + k].comps & 16) && tab[is0 + k].nrec > 0) return true;
        return false;
    };
    hipStream_t sub[kMaxLanes] = {};
    batch_streams(st, ns, sub);
    for (int it = 0; it <= nSteps - 2; it++)
        for (int q = 0; q < ns; q++) {
            const int a0 = (int)((long long)nb * q / ns), a1 = (int)((long long)nb * (q + 1) / ns);
            launch_stress_fwd_batch(sub[q], g_, opt, d_shots_ + is0 + a0, a1 - a0, md_, pc_, cells_, data_len_, it, c.src_scale, c.with_adj);
            launch_velocity_fwd_batch(sub[q], g_, opt, d_shots_ + is0 + a0, a1 - a0, md_, pc_, cells_);
            launches_ += 2;
            if (general(a0, a1)) {  // general receivers: ONE launch samples the new state of the sub-batch's shots into column it + 1
                launch_record_batch(sub[q], g_, d_shots_ + is0 + a0, a1 - a0, survey_.max_nrec, cells_, data_len_, it + 1);
                launches_++;
            }
        }
    batch_join(st, ns);
    for (int k = 0; k < nb; k++)
        if (tab[is0 + k].comps & 16) record_column(cx[k], nSteps - 1);
    if (c.if_res)
        for (int k = 0; k < nb; k++) cond_on_ ? residual_conditioned(c, cx[k]) : residual(cx[k]);
    HIP_OK(hipEventRecord(ev_[1], st));
    fwd_steps_ += (long long)nb * (nSteps - 1);
    HIP_OK(hipStreamSynchronize(st));
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev_[0], ev_[1]));
    fwd_ms_ += ms;
}

// backward time loop of the sub-batch tab[first .. first + nbb) in backward lanes 0 .. nbb-1, libCUFD.cu:500-675
void Session::batched_backward(Call &c, const std::vector<ShotDev> &tab, int first, int nbb, const ShotCtx *cx) {
    hipStream_t st = c.st;
    const Grid &g = g_;
    const KernelOptions &opt = c.opt;
    const int nSteps = par_.nSteps;
    const size_t n = cells_;
    HIP_OK(hipEventRecord(ev_[2], st));
    for (int k = 0; k < nbb; k++) HIP_OK(hipMemsetAsync(bl_[k].bwd, 0, 13 * n * sizeof(float), st));  // memories + adjoint fields
    for (int k = 0; k < nbb; k++)
        if (cx[k].quiet) HIP_OK(hipMemsetAsync(cx[k].quiet + 2 * (size_t)g.qn, 0, 2 * (size_t)g.qn * sizeof(unsigned int), st));
    int nsb = std::max(1, std::min(std::min(opt.batch_split, (int)kMaxLanes - 1), nbb));  // sub-batches on streams of their own, as in the forward loop
    auto general = [&](int a0, int a1) {  // a shot in [a0, a1) whose residual is not injected inside k_bwd_b?
        for (int k = a0; k < a1; k++)
            if (tab[first + k].lr_n == 0 && tab[first + k].nrec > 0) return true;
        return false;
    };
    // An experiment that lost, kept in the -DSEPFWI_PROBES build (option pk_ms; profiles/EXPERIMENTS.md #48): the whole sub-batch as ONE
    // persistent launch (the multi-shot loop, session_persist.cpp) where every shot's channels are a fused line (or absent).  On every
    // grid that takes the batched schedule the per-step launches below are faster, also against the loop without any synchronisation.
    bool lines = opt.pk_ms != 0 && opt.line_fuse != 0;
    for (int k = 0; k < nbb; k++) lines = lines && (tab[first + k].nrec == 0 || tab[first + k].lr_n > 0);
    const bool looped = lines && persist_prepare(pk_ms_, opt, nbb) && batched_backward_persistent(c, tab, first, nbb);
    hipStream_t sub[kMaxLanes] = {};
    if (looped) nsb = 1;
    batch_streams(st, nsb, sub);
    for (int it = nSteps - 2; it >= 0 && !looped; it--) {
        hipEvent_t *ev = probe_pair(c, it);
        Grid gs = g;
        if (opt.img_every > 1) gs.dt_img = (it % opt.img_every == 0) ? (float)opt.img_every * g.dt : 0.0f;
        for (int q = 0; q < nsb; q++) {
            const int a0 = (int)((long long)nbb * q / nsb), a1 = (int)((long long)nbb * (q + 1) / nsb);
            launch_bwd_a_batch(sub[q], gs, opt, d_shots_ + first + a0, a1 - a0, md_, pc_, n, it);
            launch_bwd_b_batch(sub[q], gs, opt, d_shots_ + first + a0, a1 - a0, md_, pc_, n, it, c.src_scale, (ev && q == 0) ? ev[0] : nullptr,
                               (ev && q == 0) ? ev[1] : nullptr);
            launches_ += 2;
            if (general(a0, a1)) {  // res_injection_exx / _ezz for the sub-batch's shots whose channels are not a fused line: ONE launch
                launch_inject_batch(sub[q], g, d_shots_ + first + a0, a1 - a0, survey_.max_nrec, n, it);
                launches_++;
            }
        }
    }
    batch_join(st, nsb);
    HIP_OK(hipEventRecord(ev_[3], st));
    bwd_steps_ += (long long)nbb * (nSteps - 1);
    HIP_OK(hipStreamSynchronize(st));
    collect_probes(c);
    float ms = 0.f;
    HIP_OK(hipEventElapsedTime(&ms, ev_[2], ev_[3]));
    bwd_ms_ += ms;
    if (looped) persist_check_pass(pk_ms_);
}

void Session::run_batched(Call &c, int Bf, int Bb) {
    hipStream_t st = c.st;
    const int nSteps = par_.nSteps, group_size = c.group_size;
    const size_t n = cells_;
    ensure_batch(Bf, c.with_adj ? Bb : 0, c.with_adj, group_size);
    HIP_OK(hipMemcpyAsync(d_stf_, c.stf_rows.data(), (size_t)group_size * nSteps * sizeof(float), hipMemcpyHostToDevice, st));
    const std::vector<ShotDev> tab = batch_table(c, Bf, Bb);
    if (c.with_adj)
        for (int k = 0; k < Bb; k++) HIP_OK(hipMemsetAsync(bl_[k].bwd + 13 * n, 0, 5 * n * sizeof(float), st));
    for (int is0 = 0; is0 < group_size; is0 += Bf) {
        const int nb = std::min(Bf, group_size - is0);
        std::vector<ShotCtx> cx;
        for (int k = 0; k < nb; k++) cx.push_back(batch_ctx(c, is0 + k, Bf, true));
        batched_forward(c, tab, is0, nb, cx);
        obs_->release_all();
        for (int k = 0; k < nb; k++) after_forward(c, cx[k]);
        for (int kb = 0; c.with_adj && kb < nb; kb += Bb) batched_backward(c, tab, is0 + kb, std::min(Bb, nb - kb), cx.data() + kb);
    }
    if (c.with_adj)  // the batch lanes' accumulators -> the session's (zeroed in prepare_buffers), summed in lane order
        for (int k = 0; k < Bb; k++) {
            launch_add_inplace(st, acc_.lam, bl_[k].bwd + 13 * n, 5 * n);
            launches_++;
        }
}

}  // namespace sepfwi
