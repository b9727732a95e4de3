// session_persist.cpp -- host side of the persistent backward time loop (k_bwd_persist, kernels_persist.hpp):
//   persist_ready         can this call's configuration run the loop?  (tiling, LDS budget, residency -- once per configuration)
//   backward_persistent   one shot's backward pass as ONE launch; false when the loop did not start (nothing touched)
//   persist_check_pass    after the pass: did a tile time out in flight?
// The loop replaces the per-step launches of Session::backward_step (Src/libCUFD.cu:545-631, same order on every array).
#include <cstdio>
#include <cstring>

#include "device_alloc.hpp"
#include "hip_check.hpp"
#include "inject_plan.hpp"
#include "kernels.hpp"
#include "session.hpp"

namespace sepfwi {

// What the loop's tiling assumes of the device: gfx950's eight XCDs (one band of rows per XCD: blockIdx % 8 shares an L2) and
// the LDS of its CUs, both asked of the device, not assumed.  Empty string, or why the loop is not for this device.
static std::string persist_device(int gpu_id, int *ncu, size_t *lds_cu, int *nband) {
    hipDeviceProp_t prop;
    HIP_OK(hipGetDeviceProperties(&prop, gpu_id));
    *ncu = prop.multiProcessorCount;
    *lds_cu = (size_t)prop.maxSharedMemoryPerMultiProcessor;
    *nband = 8;
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) return std::string("device is ") + prop.gcnArchName + ", the loop's band-per-XCD tiling is gfx950's";
    if (*ncu < *nband || *ncu % *nband) return "compute units are not a multiple of the 8 XCDs (a partitioned device)";
    return "";
}

// Can this call's configuration run k_bwd_persist?  Decided once per (workgroups per CU, waves, strip width, cost weights, LDS
// mask): the tiling is built and uploaded, the LDS-resident accumulators are chosen to fit, and the occupancy of that very
// configuration is asked of the runtime -- a grid that cannot be resident at once is never marked ready.  Whether the grid really
// is resident at once (the GPU may be shared) and every band sits on one XCD is decided by the start rendezvous of each pass
// (backward_persistent): a pass that does not start leaves everything untouched and runs as per-step launches.
bool Session::persist_ready(const Call &c, const ShotCtx &x) {
    (void)x;  // every receiver geometry takes the loop: a fused line of channels inside the bodies, anything else through persist_inject
    return persist_prepare(pk_, c.opt, 1);
}

// nshots = 1: the loop over ONE shot's grid (stream schedule).  nshots > 1: the multi-shot loop of the batched schedule -- the tiles
// cut the nshots grids stacked on each other (persist_plan.hpp make_persist_plan_multishot).
bool Session::persist_prepare(Persist &k, const KernelOptions &opt, int nshots) {
    // quiet_skip: the two-launch step skips quiet row segments faster than the loop can (fixed tiles: the tiles the wave front is in set
    // the pace of all, profiles/EXPERIMENTS.md #49); the loop's own quiet variant exists in the -DSEPFWI_PROBES build (option pk_quiet)
    if (opt.bwd_fuse != 4 || (opt.quiet_skip != 0 && opt.pk_quiet == 0)) return false;
    if (k.state >= 0 && k.nshots == nshots && k.wpc == opt.pk_wpc && k.strip_w == opt.pk_px && k.threads == 64 * opt.pk_waves && k.order == opt.pk_order &&
        k.wx == opt.pk_wx && k.wxp == opt.pk_wxp && k.wz == opt.pk_wz && k.lmask_req == opt.pk_lmask && k.snake == opt.pk_snake) {
        if (k.state == 0 && k.retry_in > 0 && --k.retry_in == 0) k.state = 1;  // a pass did not start because the GPU was busy: try again now
        return k.state == 1;
    }
    k.state = 0;
    k.nshots = nshots;
    k.wpc = opt.pk_wpc;
    k.strip_w = opt.pk_px;
    k.lmask_req = opt.pk_lmask;
    k.order = opt.pk_order;
    k.wx = opt.pk_wx;
    k.wxp = opt.pk_wxp;
    k.wz = opt.pk_wz;
    k.snake = opt.pk_snake;
    k.threads = 64 * opt.pk_waves;
    int ncu = 0, nband = 0;
    size_t lds_cu = 0;
    k.why = persist_device(gpu_id_, &ncu, &lds_cu, &nband);
    if (!k.why.empty()) return false;
    const int nseg = (g_.nx + 63) / 64;
    k.nwg = (ncu / nband) * nband * opt.pk_wpc;
    const bool multi = &k == &pk_ms_;  // the batched schedule's loop: the multi-shot kernel instance, also for a sub-batch of one
    // One shot per launch: tiles of a handful of row segments lose to the batched per-step launches (which that schedule would have
    // chosen).  The batched schedule itself: any sub-batch that gives every tile work (remainders of one or two small shots included).
    if (k.nwg <= 0 || (long long)g_.nzc * nshots * nseg < (multi ? 3LL * k.nwg / 2 : 4LL * k.nwg)) {
        k.why = "grid too small for " + std::to_string(k.nwg) + " tiles";
        return false;
    }
    PlanCost cost;
    cost.nx = g_.nx;
    cost.npml = g_.nPml;
    cost.w_xpml = opt.pk_wx;
    cost.w_xpure = opt.pk_wxp;
    cost.w_zpml = opt.pk_wz;
    cost.snake = opt.pk_snake != 0;
    // order of a tile's row segments inside a phase: edge segments first (the phase flag goes out early) or the strip order of the walk
    // (better locality, the flag goes out at the end); by default by tile size (KernelOptions::pk_order)
    const bool edge_first = opt.pk_order == 1 || (opt.pk_order == 2 && (long long)g_.nzc * nshots * nseg < 56LL * k.nwg);
    k.why = multi ? make_persist_plan_multishot(g_.nzc, nshots, nseg, k.nwg, nband, opt.pk_px, &k.plan, edge_first, cost)
                       : make_persist_plan(g_.nzc, nseg, k.nwg, nband, opt.pk_px, &k.plan, edge_first, cost);
    if (!k.why.empty()) return false;
    // accumulators in LDS: as many as fit beside the other workgroups of the CU (lam, mu, xz, a, b in that order)
    const size_t per_wg = lds_cu / (size_t)opt.pk_wpc - 256;
    const int masks[6] = {31, 15, 7, 3, 1, 0};  // (all five fit where a tile has at most 63 row segments: grids below the headline's size)
    k.lmask = -1;
    for (int mk : masks) {
        if (opt.pk_lmask != 16 && mk != opt.pk_lmask) continue;
        // (+ one word per row segment: the quiet-segment states of k_bwd_persist<.., QS>, behind the accumulators)
        const size_t need = (size_t)__builtin_popcount(mk) * (size_t)k.plan.cap * 64 * sizeof(float) + (size_t)k.plan.cap * sizeof(unsigned int);
        if (need <= per_wg) {
            k.lmask = mk;
            k.lds_bytes = need;
            break;
        }
    }
    if (k.lmask < 0) {
        k.why = "LDS accumulators do not fit";
        return false;
    }
    const int rc = persist_config_check(k.nwg, k.threads, k.lmask, k.lds_bytes + 64, multi);
    if (rc != 0) {
        static const char *const kWhy[] = {"", "no kernel instance for this LDS mask", "the LDS request is refused", "the occupancy query failed",
                                           "fewer workgroups fit the device than the grid has"};
        k.why = std::string("configuration cannot be resident at once: ") + kWhy[rc >= -4 && rc < 0 ? -rc : 0] + " (code " + std::to_string(rc) + ")";
        return false;
    }
    auto refree = [](auto *&p) {
        if (p) (void)hipFree(p);
        p = nullptr;
    };
    refree(k.d_seg);
    refree(k.d_hdr);
    refree(k.d_sync);
    refree(k.d_qnbr);
    if (!multi) {  // quiet row segments inside the loop: where a segment's stencil neighbours sit in its tile
        const std::vector<unsigned long long> nb = make_quiet_neighbours(k.plan);
        if (!nb.empty()) {
            HIP_OK(dev_malloc((void **)&k.d_qnbr, nb.size() * sizeof(unsigned long long)));
            HIP_OK(hipMemcpy(k.d_qnbr, nb.data(), nb.size() * sizeof(unsigned long long), hipMemcpyHostToDevice));
        }
    }
    const size_t sync_words = (size_t)k.nwg * 32 + 16;
    HIP_OK(dev_malloc((void **)&k.d_seg, k.plan.seg.size() * sizeof(uint32_t)));
    HIP_OK(dev_malloc((void **)&k.d_hdr, k.plan.hdr.size() * sizeof(TileHdr)));
    HIP_OK(dev_malloc((void **)&k.d_sync, sync_words * sizeof(unsigned int)));
    if (!k.d_stf) HIP_OK(dev_malloc((void **)&k.d_stf, (size_t)par_.nSteps * sizeof(float)));
    if (!k.h_err) HIP_OK(hipHostMalloc((void **)&k.h_err, 4 * sizeof(int), hipHostMallocDefault));
    HIP_OK(hipMemcpy(k.d_seg, k.plan.seg.data(), k.plan.seg.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(k.d_hdr, k.plan.hdr.data(), k.plan.hdr.size() * sizeof(TileHdr), hipMemcpyHostToDevice));
    k.state = 1;
    k.plan_gen++;
    k.why.clear();
    return true;
}

// Adjoint source of a shot whose receivers are not a fused horizontal line (strided or scattered channels, a vertical fibre,
// directional sensitivities): the injection plan of the shot (built once per session and shot, inject_plan.hpp) and the residual of
// THIS pass folded per target cell and time step (one launch).  Fills a.inj; leaves it empty for a fused line or no receivers.
const InjArgs *Session::persist_inject(const Call &c, const ShotCtx &x, hipStream_t st) {
    if (x.nrec == 0 || (x.line.n > 0 && c.opt.line_fuse != 0)) return nullptr;
    auto it = inj_.find(x.id);
    if (it == inj_.end()) {
        const Shot &sh = *x.sh;
        const InjectPlan p = make_inject_plan(sh.nrec, sh.z_rec.data(), sh.x_rec.data(), sh.sens.empty() ? nullptr : sh.sens.data(), par_.fiber != 0,
                                              g_.dx * g_.rdz, g_.nzc, g_.nx);
        InjDev d;
        d.ntgt = p.ntgt;
        auto up = [&](auto **dst, const auto &v) {
            HIP_OK(dev_malloc((void **)dst, std::max<size_t>(1, v.size()) * sizeof(v[0])));
            if (!v.empty()) HIP_OK(hipMemcpy(*dst, v.data(), v.size() * sizeof(v[0]), hipMemcpyHostToDevice));
            device_bytes_ += (long long)(v.size() * sizeof(v[0]));
        };
        up(&d.lookup, p.lookup);
        up(&d.segs, p.segs);
        up(&d.tgt_start, p.tgt_start);
        up(&d.ent_rec, p.ent_rec);
        up(&d.ent_w, p.ent_w);
        for (size_t sidx = 0; sidx < p.lookup.size(); sidx++)
            if (p.lookup[sidx] >= 0) d.target_segs.push_back((int)sidx);  // (which tiles own them depends on the tiling: below)
        it = inj_.emplace(x.id, d).first;
    }
    const InjDev &d = it->second;
    const size_t need = (size_t)par_.nSteps * (size_t)std::max(1, d.ntgt);
    if (need > inj_val_len_) {
        if (inj_val_) (void)hipFree(inj_val_);
        inj_val_ = nullptr;
        HIP_OK(dev_malloc((void **)&inj_val_, need * sizeof(float)));
        device_bytes_ += (long long)((need - inj_val_len_) * sizeof(float));
        inj_val_len_ = need;
    }
    launch_inject_values(st, x.res, x.nrec, par_.nSteps, d.tgt_start, d.ent_rec, d.ent_w, d.ntgt, inj_val_);
    launches_++;
    if (it->second.tile_gen != pk_.plan_gen) {  // which tiles of the CURRENT tiling own target cells (a new tiling: rebuilt)
        InjDev &dd = it->second;
        std::vector<unsigned char> has((size_t)pk_.nwg, 0);
        for (int sidx : dd.target_segs) has[(size_t)pk_.plan.owner[(size_t)sidx]] = 1;
        if (dd.tile_has) (void)hipFree(dd.tile_has);
        dd.tile_has = nullptr;
        HIP_OK(dev_malloc((void **)&dd.tile_has, has.size()));
        HIP_OK(hipMemcpy(dd.tile_has, has.data(), has.size(), hipMemcpyHostToDevice));
        dd.tile_gen = pk_.plan_gen;
    }
    InjDev &dd = it->second;  // the kernel reaches the tables through ONE pointer: this block, in device memory
    dd.h_args.tile_has = d.tile_has;
    dd.h_args.lookup = d.lookup;
    dd.h_args.segs = d.segs;
    dd.h_args.val = inj_val_;
    dd.h_args.ntgt = d.ntgt;
    dd.h_args.nseg = (g_.nx + 63) / 64;
    if (!dd.d_args) HIP_OK(dev_malloc((void **)&dd.d_args, sizeof(InjArgs)));
    HIP_OK(hipMemcpyAsync(dd.d_args, &dd.h_args, sizeof(InjArgs), hipMemcpyHostToDevice, st));  // (h_args lives in the session's map: valid until the copy has run)
    return dd.d_args;
}

std::string Session::loop_status() {
    std::lock_guard<std::mutex> lock(mu_);
    const Persist &k = last_batched_ ? pk_ms_ : pk_;  // the schedule of the last call: multi-shot loop (batched) or one loop per shot (streams)
    if (k.state < 0) return "not considered yet (no gradient call, bwd_fuse != 4, or shots whose channels are not fused lines in a batched call)";
    return k.state == 1 ? std::string() : k.why;
}

// The loop is not used for this pass (and, unless `retry_in` says otherwise, for the rest of the session): say so once.
void Session::persist_demote(Persist &k, const std::string &why, int retry_in) {
    k.state = 0;
    k.why = why;
    k.retry_in = retry_in;
    if (k.aborts++ == 0) fprintf(stderr, "sepfwi: persistent backward loop not started (%s); this pass runs as per-step launches\n", why.c_str());
}

// One shot's backward pass as one launch.  Returns false when the loop did not run -- the launch was refused, or the start
// rendezvous found the grid not resident at once / a band spread over several XCDs: in both cases NOTHING has been touched, and the
// caller runs the per-step launches.  Synchronises the stream (the verdict of the rendezvous is read on the host).
bool Session::backward_persistent(Call &c, const ShotCtx &x, const BwdLane &L) {
    Persist &k = pk_;
    const int nSteps = par_.nSteps;
    hipStream_t st = L.s;
    HIP_OK(hipMemcpyAsync(k.d_stf, x.stf_s, (size_t)nSteps * sizeof(float), hipMemcpyHostToDevice, st));
    PersistArgs a{};
    ShotDev &d = a.s;
    d.fields = x.state;
    d.frame = x.frame;
    d.stf = k.d_stf;
    d.bmem = L.bm.dvz_dz;
    d.adj = L.adj.vz;
    d.acc = L.acc.lam;
    d.res = x.res;
    d.stf_grad = stf_grad_ + (size_t)x.is * nSteps;
    d.z_src = x.sh->z_src;
    d.x_src = x.sh->x_src;
    d.lr_z = x.line.z;
    d.lr_x0 = x.line.x0;
    d.lr_n = x.line.n;
    d.nrec = x.nrec;
    d.src_rxz = (float)x.sh->src_rxz;
    a.injp = persist_inject(c, x, st);
    if (x.quiet && k.d_qnbr && !a.injp) {  // option quiet_skip (fused line of channels, or none): the quiet variant of the loop
        a.q.maps = x.quiet;
        a.q.nbr = k.d_qnbr;
    }
    if (!persist_launch(k, c, a, st)) return false;
    persist_steps_ += (long long)(nSteps - 1);
    return true;
}

// Several shots' backward passes as one launch (batched schedule; shots tab[first .. first + nbb) in backward lanes 0 .. nbb-1).
// Same contract as backward_persistent.  The shots' arrays must lie at constant strides (ensure_batch's arenas) -- checked here,
// not assumed -- and their channels must be fused lines (or absent).
bool Session::batched_backward_persistent(Call &c, const std::vector<ShotDev> &tab, int first, int nbb) {
    Persist &k = pk_ms_;
    const int nSteps = par_.nSteps;
    PersistArgs a{};
    a.s = tab[first];
    MultiShot &m = a.ms;
    m.shots = d_shots_ + first;
    m.nshot = nbb;
    if (nbb > 1) {
        m.state_stride = (size_t)(tab[first + 1].fields - tab[first].fields);
        m.bwd_stride = (size_t)(tab[first + 1].bmem - tab[first].bmem);
        m.frame_stride = (size_t)(tab[first + 1].frame - tab[first].frame);
        m.res_stride = (size_t)(tab[first + 1].res - tab[first].res);
    }
    for (int q = 0; q < nbb; q++) {
        const ShotDev &d = tab[first + q], &d0 = tab[first];
        const bool ok = d.fields == d0.fields + q * m.state_stride && d.bmem == d0.bmem + q * m.bwd_stride && d.adj == d0.adj + q * m.bwd_stride &&
                        d.acc == d0.acc + q * m.bwd_stride && d.frame == d0.frame + q * m.frame_stride && d.res == d0.res + q * m.res_stride &&
                        d.stf == d0.stf + (size_t)q * nSteps && d.stf_grad == d0.stf_grad + (size_t)q * nSteps;
        if (!ok) throw std::logic_error("multi-shot loop: the batch lanes do not lie at constant strides");
    }
    if (!persist_launch(k, c, a, c.st)) return false;
    persist_steps_ += (long long)nbb * (nSteps - 1);
    return true;
}

// The launch of a prepared configuration and its start verdict (shared by the single- and the multi-shot loop).
bool Session::persist_launch(Persist &k, Call &c, PersistArgs &a, hipStream_t st) {
    const int nSteps = par_.nSteps;
    unsigned int *band_xcc = k.d_sync + (size_t)k.nwg * 32;
    int *err = (int *)(band_xcc + 10);
    HIP_OK(hipMemsetAsync(k.d_sync, 0, ((size_t)k.nwg * 32 + 16) * sizeof(unsigned int), st));
    HIP_OK(hipMemsetAsync(band_xcc, 0xff, 8 * sizeof(unsigned int), st));
    a.media = md_.lam;
    a.cz = pc_.a_z;
    a.n = cells_;
    a.it_hi = nSteps - 2;
    a.it_lo = 0;
    a.src_scale = c.src_scale;
    a.img_every = c.opt.img_every;
    a.nband = k.plan.nband;
    a.per_band = k.plan.per_band;
    a.cap = k.plan.cap;
    a.seg = k.d_seg;
    a.hdr = k.d_hdr;
    a.flags = k.d_sync;
    a.band_xcc = band_xcc;
    a.err = err;
    a.phase0 = 0;
    a.nosync = c.opt.pk_nosync;
    a.lock = c.opt.pk_lock;
    a.prio = c.opt.pk_prio;
    const int rc = launch_bwd_persist(st, g_, c.opt, a, k.nwg, k.threads, k.lmask, k.lds_bytes + 64);
    if (rc != 0 || hipPeekAtLastError() != hipSuccess) {  // refused before anything ran: the per-step launches from now on
        const hipError_t e = hipGetLastError();
        persist_demote(k, "the launch was refused (code " + std::to_string(rc) + (e != hipSuccess ? std::string(", ") + hipGetErrorString(e) : std::string()) + ")", 0);
        return false;
    }
    launches_++;
    HIP_OK(hipMemcpyAsync(k.h_err, err, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipMemcpyAsync(k.h_err + 1, band_xcc + 9, sizeof(int), hipMemcpyDeviceToHost, st));
    HIP_OK(hipStreamSynchronize(st));
    if (k.h_err[1] != (int)kPersistGo) {  // the loop did not start: this pass, and the session from now on, as per-step launches
        const bool busy = k.h_err[1] != (int)kPersistAbortPlacement;
        // transient contention: another try after 16 passes, three times at most
        persist_demote(k, busy ? "the grid was not resident at once (GPU busy, or the configuration does not fit)" : "workgroups of one band run on several XCDs",
                       busy && k.aborts < 3 ? 16 : 0);
        return false;
    }
    return true;
}

// After a pass that ran in the loop: a wait inside the pass timed out (a tile's neighbour never published)?  The results are
// discarded, the session goes back to the two-launch step and the call fails with where the tiles stood (the reference: exit(1),
// Src/utilities.h:28-36).
void Session::persist_check_pass(Persist &k) {
    if (k.h_err[0] == 0) return;
    std::vector<unsigned int> fl((size_t)k.nwg * 32);  // flags[tile] = phases whose edge part is complete
    HIP_OK(hipMemcpy(fl.data(), k.d_sync, fl.size() * sizeof(unsigned int), hipMemcpyDeviceToHost));
    unsigned int lo = ~0u, hi = 0;
    int t_lo = 0, never = 0;
    for (int t = 0; t < k.nwg; t++) {
        const unsigned int v = fl[(size_t)t * 32] & 0x0fffffffu;  // (the top four bits: quiet-segment summary)
        if (v < lo) {
            lo = v;
            t_lo = t;
        }
        hi = std::max(hi, v);
        never += v == 0;
    }
    k.state = 0;
    k.why = "a pass failed";
    throw HipError(std::string("persistent backward loop: a tile waited for its neighbour beyond the time limit") + " (results discarded; tiles reached phases " +
                   std::to_string(lo) + " ... " + std::to_string(hi) + " of " + std::to_string(2 * (par_.nSteps - 1)) + ", slowest tile " + std::to_string(t_lo) + ", " +
                   std::to_string(never) + " of " + std::to_string(k.nwg) + " never published)");
}

}  // namespace sepfwi
