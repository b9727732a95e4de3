// session.hpp -- persistent propagation session (see session.cpp).
#pragma once
#include <hip/hip_runtime.h>

#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sepfwi.h"
#include "conditioning.hpp"
#include "config.hpp"
#include "errors.hpp"
#include "fwi_types.hpp"
#include "kernels.hpp"
#include "obs_store.hpp"
#include "persist_plan.hpp"

namespace sepfwi {

class Session {
  public:
    Session(const std::string &para_fname, int gpu_id, const std::string &para_text, const std::string &survey_text,
            const Params &par, const Survey &survey);
    ~Session();
    Session(const Session &) = delete;
    Session &operator=(const Session &) = delete;

    bool matches(const std::string &ptext, const std::string &stext) const { return ptext == para_text_ && stext == survey_text_; }
    void run(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf, const float *Lambda,
             const float *Mu, const float *Den, const float *stf, int calc_id, int group_size, const int *shot_ids,
             hipStream_t ext_stream, bool async);
    void stats(sepfwi_stats *out) const;
    std::string loop_status();  // "" while the persistent backward loop is in use, else why not (sepfwi_loop_status)
    void drop_observed();
    // observed axial strain of one shot from memory ([nrec][nSteps] like the files; host or device pointer)
    void set_observed(int shot_id, const float *ett, int nrec, int nSteps);
    // test hook: wavefield `which` (0..4 vz, vx, szz, sxx, sxz; 5..9 their adjoint twins) of forward lane `lane` as left
    // by the last call, dense (nz - nPad, nx) row-major, host or device pointer
    void copy_field(int lane, int which, float *out);
    const Params &params() const { return par_; }

  private:
    template <class T> T *dalloc(size_t n);
    void init_grid();        // constructor steps
    void alloc_arrays();
    void upload_profiles();
    void upload_survey();
    void ensure_lanes(int n_lanes, bool with_frames);
    void ensure_batch(int n_fwd, int n_bwd, bool with_frames, int n_shots);
    void order_after_null_stream(hipStream_t st);
    void condition_gather(hipStream_t st, float *gather_rec_major, int shot_id, int nrec);

    // ---- one cufd call (run): its state and its passes -------------------------------------------------------------------
    struct Call {  // what every pass of one call shares; ONE snapshot of the kernel options
        KernelOptions opt;
        hipStream_t st = nullptr;
        bool if_res = false, with_adj = false, to_store = false;
        int group_size = 0;
        const int *shot_ids = nullptr;
        std::vector<float> stf_rows;  // tapered source traces of the call's shots (host)
        float src_scale = 0.0f;
        int n_probe = 0;              // HIP-event pairs handed out in the running backward pass
    };
    struct ShotCtx {  // one shot of the call in the lane it runs in
        int is, id, nrec, comps;
        const Shot *sh;
        const int *rec;
        const float *stf_s, *d_obs;
        const float *sens;  // directional sensitivities of this shot's channels (device) or null
        bool scratch;
        LineRec line;
        float *state;  // [5 fields | 8 memory variables] of this lane
        unsigned int *quiet;  // option quiet_skip: the lane's four quiet-segment maps (forward v, forward s, adjoint v, adjoint s), or null
        Fields fld;
        PmlMem mem;
        float *frame, *syn, *res;
        hipStream_t st;
    };
    struct BwdLane {  // stream + backward-pass memory variables + adjoint fields + imaging accumulators
        hipStream_t s;
        PmlMem bm;
        Fields adj;
        ImgAcc acc;
    };
    void prepare_media(Call &c, const float *Lambda, const float *Mu, const float *Den);
    void prepare_buffers(Call &c, const float *stf);
    ShotCtx make_ctx(const Call &c, int is, int lane, hipStream_t lane_st, bool with_obs = true);
    void use_state(ShotCtx &x, float *state) const;
    static constexpr int kQuietSlots = 4 + 64;  // one per stream lane (kMaxLanes) and batch lane (option batch_f <= 64)
    unsigned int *quiet_slot(int slot) const { return quiet_pool_ + (size_t)slot * 4 * (size_t)g_.qn; }
    bool quiet_wanted(const Call &c, const ShotCtx &x) const { return c.opt.quiet_skip != 0 && (x.nrec == 0 || (x.line.n > 0 && c.opt.line_fuse != 0)); }
    float *syn_of(const ShotCtx &x, int comp) const { return x.syn + (size_t)comp * data_len_; }
    bool forward_inline(const Call &c, const ShotCtx &x) const { return x.line.n > 0 && !(x.comps & 1) && c.opt.line_fuse != 0; }
    // forward pass of one shot, stream form (libCUFD.cu:268-332)
    void forward_init(const ShotCtx &x);
    void forward_step(const Call &c, const ShotCtx &x, int it, bool inl);
    void record_column(const ShotCtx &x, int column);
    void residual(const ShotCtx &x);
    void residual_conditioned(const Call &c, const ShotCtx &x);
    // what a forward pass leaves behind, by kind of call
    void after_forward(Call &c, const ShotCtx &x);
    void export_gathers(const Call &c, const ShotCtx &x);
    void scratch_dumps(const Call &c, const ShotCtx &x);
    // backward pass of one shot, stream form (libCUFD.cu:500-675)
    void backward_init(const BwdLane &L);
    void backward_step(Call &c, const ShotCtx &x, const BwdLane &L, int it);
    void backward(Call &c, const ShotCtx &x);
    // the same pass as ONE persistent launch (option bwd_fuse = 4; kernels.hip k_bwd_persist)
    struct Persist;
    bool persist_ready(const Call &c, const ShotCtx &x);
    bool persist_prepare(Persist &k, const KernelOptions &opt, int nshots);
    bool backward_persistent(Call &c, const ShotCtx &x, const BwdLane &L);
    bool batched_backward_persistent(Call &c, const std::vector<ShotDev> &tab, int first, int nbb);
    bool persist_launch(Persist &k, Call &c, PersistArgs &a, hipStream_t st);
    const InjArgs *persist_inject(const Call &c, const ShotCtx &x, hipStream_t st);
    void persist_demote(Persist &k, const std::string &why, int retry_in);
    void persist_check_pass(Persist &k);
    hipEvent_t *probe_pair(Call &c, int it);
    void collect_probes(Call &c);
    // the two schedules of a call's shots
    void run_streams(Call &c);
    // batched schedule (session_batched.cpp)
    void run_batched(Call &c, int Bf, int Bb);
    ShotCtx batch_ctx(const Call &c, int is, int Bf, bool with_obs);
    std::vector<ShotDev> batch_table(const Call &c, int Bf, int Bb);
    void batch_streams(hipStream_t st, int ns, hipStream_t *sub);
    void batch_join(hipStream_t st, int ns);
    void batched_forward(Call &c, const std::vector<ShotDev> &tab, int is0, int nb, const std::vector<ShotCtx> &cx);
    void batched_backward(Call &c, const std::vector<ShotDev> &tab, int first, int nbb, const ShotCtx *cx);
    void write_outputs(Call &c, float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf);

    std::string para_fname_;
    int gpu_id_;
    std::string para_text_, survey_text_;
    Params par_;
    Survey survey_;
    Grid g_{};
    std::mutex mu_;
    hipStream_t own_stream_ = nullptr;
    hipEvent_t ev_[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ev_order_ = nullptr;
    static constexpr int kProbePairs = 64;
    hipEvent_t probe_ev_[2 * kProbePairs] = {};
    double probe_us_ = 0.0;
    long long probe_calls_ = 0;
    std::vector<void *> allocs_;
    long long device_bytes_ = 0;

    size_t cells_ = 0, data_len_ = 0;
    // extra forward lanes (lane 0 = state_/frame_/syn_/res_ on the call's stream): fields + memories, frames, seismograms,
    // residual, stream, join event
    static constexpr int kMaxLanes = 4;
    struct XLane {
        float *state = nullptr, *frame = nullptr, *syn = nullptr, *res = nullptr;
        hipStream_t stream = nullptr;
        hipEvent_t join = nullptr;
    };
    XLane xl_[kMaxLanes];
    size_t stf_grad_len_ = 0;
    // batched mode: lanes of per-shot state (forward: fields + memories, frames, seismograms, residual; backward: memories,
    // adjoint fields, accumulators) and the device table of the call's shots
    struct BLane {
        float *state = nullptr, *bwd = nullptr, *frame = nullptr, *syn = nullptr, *res = nullptr;
    };
    std::vector<BLane> bl_;
    struct BatchArenas {  // the lanes of one kind at a constant stride (ensure_batch)
        float *state = nullptr, *syn = nullptr, *res = nullptr, *frame = nullptr, *bwd = nullptr;
        int n_state = 0, n_syn = 0, n_res = 0, n_frame = 0, n_bwd = 0;
    } ba_;
    ShotDev *d_shots_ = nullptr;
    int shots_cap_ = 0;
    float *d_stf_ = nullptr;
    size_t d_stf_len_ = 0;
    bool last_batched_ = false;
    float *state_ = nullptr, *media_ = nullptr, *acc_buf_ = nullptr, *in_stage_ = nullptr, *grad_stage_ = nullptr;
    float *frame_ = nullptr, *syn_ = nullptr, *res_ = nullptr, *xpose_ = nullptr, *stf_grad_ = nullptr, *h_io_ = nullptr;
    double *scal_ = nullptr;
    unsigned int *cp2_bits_ = nullptr;
    int *rec_idx_ = nullptr;
    float *sens_ = nullptr;  // directional DAS sensitivities (3 per channel) or null
    // data conditioning (parameter keys if_win / filter / if_cross_misfit): per-channel windows and weights (3 per channel:
    // start, end, weight; same offsets as rec_idx_), a second [rec][it] scratch gather, the hipFFT work space
    bool cond_on_ = false;
    float *win_ = nullptr, *xpose2_ = nullptr;
    std::unique_ptr<Conditioner> cond_;
    std::vector<int> rec_off_;
    Fields fld_{}, adj_{};
    PmlMem mem_{};
    Media md_{};
    PmlCoef pc_{};
    ImgAcc acc_{};
    std::unique_ptr<ObservedStore> obs_;
    unsigned int *quiet_pool_ = nullptr;  // kQuietSlots x 4 maps of Grid::qn words (Fields::q)
    // persistent backward time loop: the tiling in use, its device copy, synchronisation words, what the census of the grid said
    struct Persist {
        PersistPlan plan;
        uint32_t *d_seg = nullptr;
        TileHdr *d_hdr = nullptr;
        unsigned int *d_sync = nullptr;  // [nwg x 32 flag words | 8 band XCC ids | arrived | err]
        unsigned long long *d_qnbr = nullptr;  // quiet variant: stencil neighbours of every row segment inside its tile (persist_plan.hpp)
        float *d_stf = nullptr;
        int *h_err = nullptr;            // pinned
        int nwg = 0, threads = 0, lmask = 0, lmask_req = -1, wpc = 0, strip_w = 0, order = -1, wx = -1, wxp = -1, wz = -1, snake = -1, nshots = 0;
        size_t lds_bytes = 0;
        int state = -1;                  // -1 not examined for this configuration, 0 the two-launch step is used, 1 ready
        std::string why;                 // when state == 0
        int plan_gen = 0;                // counts the tilings built (what depends on one is rebuilt when it changes)
        int retry_in = 0, aborts = 0;    // passes until the loop is tried again after a start rendezvous that failed; how often it did
    } pk_, pk_ms_;  // one shot per launch (stream schedule) / the shots of a backward sub-batch in one launch (batched schedule)
    // adjoint-source injection inside the loop for shots whose receivers are not a fused line: the plan of each such shot (device
    // copies, built on first use) and the pass's residual folded per target cell [nSteps][ntgt]
    struct InjDev {
        int *lookup = nullptr, *tgt_start = nullptr, *ent_rec = nullptr;
        InjSeg *segs = nullptr;
        float *ent_w = nullptr;
        std::vector<int> target_segs;       // row segments (z * nseg + xs) that hold target cells
        unsigned char *tile_has = nullptr;  // per tile of the tiling numbered tile_gen: owns target cells?
        InjArgs h_args{}, *d_args = nullptr;  // what the kernel reads through PersistArgs::injp
        int ntgt = 0, tile_gen = -1;
    };
    std::map<int, InjDev> inj_;
    float *inj_val_ = nullptr;
    size_t inj_val_len_ = 0;
    long long persist_steps_ = 0;
    long long quiet_active_ = 0, quiet_total_ = 0;
    unsigned int *quiet_last_ = nullptr;  // maps of the shot whose forward pass started last in this call

    double fwd_ms_ = 0, bwd_ms_ = 0, total_ms_ = 0;
    long long fwd_steps_ = 0, bwd_steps_ = 0, launches_ = 0;
    int last_shots_ = 0, last_calc_ = -1;
};

// The registry hands out shared ownership: a session that another thread is still running survives being replaced
// (parameter file rewritten) or released.
std::shared_ptr<Session> get_session(const std::string &para_fname, int gpu_id);
std::shared_ptr<Session> find_session(const std::string &para_fname, int gpu_id);
void release_all_sessions();
void invalidate_observed_all();

}  // namespace sepfwi
