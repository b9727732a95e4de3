// obs_store.hpp -- the session's store of observed axial-strain gathers.
//
// Replaces the reference's per-call fread of four files per shot (Src/libCUFD.cu:216-223): a gather is read once -- from the
// shot's Shot_ett{id}.bin, from the survey's packed file, or handed over from memory -- transposed to the device layout
// (time-major [it][rec]; with data conditioning: conditioned, [rec][it]) and kept.  Two tiers:
//   HBM          every gather, as long as the budget allows (default: no budget, everything stays in HBM);
//   pinned host  with a budget ("obs_cache_mb"), the least recently used gathers that are not in use move to page-locked host
//                memory (hipHostMalloc) and come back by one asynchronous copy on the call's stream when a shot needs them.
// The bytes that travel are the device-layout bytes, so a gather that went to the host tier and back is bit-identical.
#pragma once
#include <hip/hip_runtime.h>

#include <functional>
#include <map>
#include <string>

#include "config.hpp"
#include "host_checks.hpp"

namespace sepfwi {

class ObservedStore {
  public:
    struct Host {  // what the store borrows from its session
        int gpu_id = 0;
        const Params *par = nullptr;
        const Survey *survey = nullptr;
        float *xpose = nullptr;  // device scratch, one gather
        float *h_io = nullptr;   // pinned host staging, one gather
        bool cond_on = false;
        std::function<void(hipStream_t, float *, int, int)> condition;  // window + band-pass one [rec][it] gather in place (shot id, nrec)
    };
    explicit ObservedStore(const Host &h) : h_(h) {}
    ~ObservedStore() { clear(); }
    ObservedStore(const ObservedStore &) = delete;
    ObservedStore &operator=(const ObservedStore &) = delete;

    void set_budget_bytes(long long b) { budget_ = b > 0 ? b : 0; }  // 0: unlimited
    long long budget_bytes() const { return budget_; }
    // How many gathers of `bytes` each a call may hold at once (a group of concurrent forward passes) out of `want`.
    int max_group(size_t bytes, int want) const;

    // Device pointer of the shot's gather, resident and protected from eviction until release_all().  nrec <= 0: nullptr.
    const float *acquire(int shot_id, int nrec, hipStream_t st);
    void release_all();
    // [nrec][nSteps] from memory (host or device pointer): sepfwi_set_observed
    void put(int shot_id, const float *ett, int nrec, hipStream_t st);
    // the session's own modelled gather, time-major on the device: calc_id SEPFWI_CALC_OBSERVE_TO_STORE
    void put_device_gather(int shot_id, const float *syn_time_major, int nrec, hipStream_t st);
    void forget(int shot_id);  // its file was just rewritten
    void clear();

    long long device_bytes() const { return dev_bytes_; }
    long long host_bytes() const { return host_bytes_; }
    long long evictions() const { return evictions_; }
    long long uploads() const { return uploads_; }

  private:
    struct Entry {
        float *d = nullptr;  // HBM copy (device layout) or null
        float *h = nullptr;  // pinned host copy of the same bytes or null
        size_t bytes = 0;
        long long size = 0, mtime_ns = 0;  // stamp of the file behind it
        bool from_memory = false;          // no file behind it
        bool held = false;                 // in use by the running call
        long long tick = 0;                // last use
    };
    size_t want_bytes(int nrec) const { return (size_t)nrec * (size_t)h_.par->nSteps * sizeof(float); }
    void make_room(size_t bytes, hipStream_t st);
    void to_host_tier(Entry &e, hipStream_t st);
    void materialise(Entry &e, hipStream_t st);
    void reset(Entry &e, size_t bytes);  // fresh device buffer of `bytes`, host copy dropped
    void free_entry(Entry &e);
    void fill_from_xpose(Entry &e, int shot_id, int nrec, hipStream_t st);  // xpose ([rec][it]) -> device layout
    long long pack_offset(int shot_id, int nrec);

    Host h_;
    std::map<int, Entry> obs_;
    PackIndex pack_;
    long long pack_mtime_ns_ = -1, pack_size_ = -1;
    long long budget_ = 0, dev_bytes_ = 0, host_bytes_ = 0, evictions_ = 0, uploads_ = 0, clock_ = 0;
};

}  // namespace sepfwi
