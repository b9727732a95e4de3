// obs_store.cpp -- see obs_store.hpp.
#include "obs_store.hpp"

#include <sys/stat.h>

#include <algorithm>
#include <cstdio>
#include <stdexcept>

#include "device_alloc.hpp"
#include "hip_check.hpp"
#include "kernels.hpp"

namespace sepfwi {

static const char *kEttFile = "/Shot_ett";  // libCUFD.cu:216-223,755-769

int ObservedStore::max_group(size_t bytes, int want) const {
    if (budget_ <= 0 || bytes == 0) return want;
    const long long fit = budget_ / (long long)bytes;
    return (int)std::max(1LL, std::min<long long>(fit, want));
}

void ObservedStore::free_entry(Entry &e) {
    if (e.d) {
        (void)hipFree(e.d);
        dev_bytes_ -= (long long)e.bytes;
        e.d = nullptr;
    }
    if (e.h) {
        (void)hipHostFree(e.h);
        host_bytes_ -= (long long)e.bytes;
        e.h = nullptr;
    }
}

void ObservedStore::clear() {
    (void)hipSetDevice(h_.gpu_id);
    for (auto &kv : obs_) free_entry(kv.second);
    obs_.clear();
}

void ObservedStore::forget(int shot_id) {
    auto it = obs_.find(shot_id);
    if (it == obs_.end()) return;
    free_entry(it->second);
    obs_.erase(it);
}

void ObservedStore::release_all() {
    for (auto &kv : obs_) kv.second.held = false;
}

// HBM copy -> pinned host copy (made once: the device bytes never change afterwards), HBM copy freed
void ObservedStore::to_host_tier(Entry &e, hipStream_t st) {
    if (!e.d) return;
    if (!e.h) {
        HIP_OK(hipHostMalloc((void **)&e.h, e.bytes, hipHostMallocDefault));
        host_bytes_ += (long long)e.bytes;
        HIP_OK(hipMemcpyAsync(e.h, e.d, e.bytes, hipMemcpyDeviceToHost, st));
        HIP_OK(hipStreamSynchronize(st));
    }
    (void)hipFree(e.d);
    e.d = nullptr;
    dev_bytes_ -= (long long)e.bytes;
    evictions_++;
}

// room for `bytes` more in HBM under the budget: the least recently used gathers that no running shot needs go to the host tier
void ObservedStore::make_room(size_t bytes, hipStream_t st) {
    if (budget_ <= 0) return;
    while (dev_bytes_ + (long long)bytes > budget_) {
        Entry *lru = nullptr;
        for (auto &kv : obs_) {
            Entry &e = kv.second;
            if (e.d && !e.held && (!lru || e.tick < lru->tick)) lru = &e;
        }
        if (!lru) {
            if (dev_bytes_ == 0) return;  // a single gather larger than the budget: it has to be resident to be used at all
            throw std::invalid_argument("observed-data budget (obs_cache_mb) is smaller than the gathers one group of shots needs at once");
        }
        to_host_tier(*lru, st);
    }
}

void ObservedStore::materialise(Entry &e, hipStream_t st) {
    if (e.d || e.bytes == 0) return;
    make_room(e.bytes, st);
    HIP_OK(dev_malloc((void **)&e.d, e.bytes));
    dev_bytes_ += (long long)e.bytes;
    HIP_OK(hipMemcpyAsync(e.d, e.h, e.bytes, hipMemcpyHostToDevice, st));  // pinned source: asynchronous, ordered on the call's stream
    uploads_++;
}

void ObservedStore::reset(Entry &e, size_t bytes) {
    if (e.h) {  // bytes are about to change: the host copy is stale
        (void)hipHostFree(e.h);
        host_bytes_ -= (long long)e.bytes;
        e.h = nullptr;
    }
    if (e.d && e.bytes != bytes) {
        (void)hipFree(e.d);
        dev_bytes_ -= (long long)e.bytes;
        e.d = nullptr;
    }
    e.bytes = bytes;
}

// gather in xpose ([rec][it]) -> the entry's device layout
void ObservedStore::fill_from_xpose(Entry &e, int shot_id, int nrec, hipStream_t st) {
    if (h_.cond_on) {  // kept conditioned and trace-major
        h_.condition(st, h_.xpose, shot_id, nrec);
        HIP_OK(hipMemcpyAsync(e.d, h_.xpose, e.bytes, hipMemcpyDeviceToDevice, st));
    } else {
        launch_transpose(st, h_.xpose, e.d, nrec, h_.par->nSteps);  // [rec][it] -> [it][rec]
    }
    HIP_OK(hipStreamSynchronize(st));
}

void ObservedStore::put(int shot_id, const float *ett, int nrec, hipStream_t st) {
    const size_t want = want_bytes(nrec);
    Entry &e = obs_[shot_id];
    const bool was_held = e.held;
    reset(e, want);
    e.from_memory = true;
    e.tick = ++clock_;
    if (nrec <= 0) return;
    e.held = true;  // not a candidate while room is made for it
    if (!e.d) {
        make_room(want, st);
        HIP_OK(dev_malloc((void **)&e.d, want));
        dev_bytes_ += (long long)want;
    }
    HIP_OK(hipMemcpyAsync(h_.xpose, ett, want, hipMemcpyDefault, st));
    fill_from_xpose(e, shot_id, nrec, st);
    e.held = was_held;
}

void ObservedStore::put_device_gather(int shot_id, const float *syn_time_major, int nrec, hipStream_t st) {
    const size_t want = want_bytes(nrec);
    Entry &e = obs_[shot_id];
    const bool was_held = e.held;
    reset(e, want);
    e.from_memory = true;
    e.tick = ++clock_;
    if (nrec <= 0) return;
    e.held = true;
    if (!e.d) {
        make_room(want, st);
        HIP_OK(dev_malloc((void **)&e.d, want));
        dev_bytes_ += (long long)want;
    }
    if (h_.cond_on) {  // the device gather is time-major; the conditioned store is trace-major
        launch_transpose(st, syn_time_major, h_.xpose, h_.par->nSteps, nrec);  // [it][rec] -> [rec][it]
        h_.condition(st, h_.xpose, shot_id, nrec);
        HIP_OK(hipMemcpyAsync(e.d, h_.xpose, want, hipMemcpyDeviceToDevice, st));
    } else {
        HIP_OK(hipMemcpyAsync(e.d, syn_time_major, want, hipMemcpyDeviceToDevice, st));
    }
    HIP_OK(hipStreamSynchronize(st));
    e.held = was_held;
}

// Byte offset of a shot's gather in the packed observed-data file, or -1 when the pack does not hold the shot.  The index is
// re-read when the file changes.
long long ObservedStore::pack_offset(int shot_id, int nrec) {
    const std::string &fn = h_.par->obs_pack_fname;
    struct stat sb;
    if (stat(fn.c_str(), &sb) != 0) throw IoError("cannot read packed observed data '" + fn + "'");
    const long long stamp = (long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec;
    if (stamp != pack_mtime_ns_ || (long long)sb.st_size != pack_size_) {
        pack_mtime_ns_ = pack_size_ = -1;
        read_pack_index(fn, h_.par->nSteps, (long long)sb.st_size, &pack_);
        pack_mtime_ns_ = stamp;
        pack_size_ = (long long)sb.st_size;
    }
    auto it = pack_.entries.find(shot_id);
    if (it == pack_.entries.end()) return -1;
    if (it->second.second != nrec) throw IoError("packed observed data: shot " + std::to_string(shot_id) + " has another channel count than the survey");
    return it->second.first;
}

const float *ObservedStore::acquire(int shot_id, int nrec, hipStream_t st) {
    if (nrec <= 0) return nullptr;  // nothing to compare against
    const size_t want = want_bytes(nrec);
    {
        auto im = obs_.find(shot_id);
        if (im != obs_.end() && im->second.from_memory && im->second.bytes == want) {  // handed over from memory
            Entry &e = im->second;
            e.held = true;
            e.tick = ++clock_;
            materialise(e, st);
            return e.d;
        }
    }
    // where the gather lives: the survey's packed file when the parameter file names one and it holds this shot, else the
    // shot's own Shot_ett{id}.bin (libCUFD.cu:216-223)
    std::string fn = h_.par->data_dir_name + kEttFile + std::to_string(shot_id) + ".bin";
    long long file_off = 0;
    if (!h_.par->obs_pack_fname.empty()) {
        const long long off = pack_offset(shot_id, nrec);
        if (off >= 0) {
            fn = h_.par->obs_pack_fname;
            file_off = off;
        }
    }
    struct stat sb;
    if (stat(fn.c_str(), &sb) != 0) {
        // a frequent cause since gathers can be observed straight into a session's store: that session is another one
        throw IoError("cannot read observed data '" + fn + "' (if shot " + std::to_string(shot_id) + " was observed into a session's memory store -- "
                      "calc_id 3 / obscalc(to_store=True) -- it lives in the store of the rank or device that modelled it; this call gave the shot to another one)");
    }
    if ((long long)sb.st_size < file_off + (long long)want) throw IoError("observed data '" + fn + "' is shorter than nrec*nSteps floats");
    const long long stamp = (long long)sb.st_mtim.tv_sec * 1000000000LL + sb.st_mtim.tv_nsec;
    auto it = obs_.find(shot_id);
    if (it != obs_.end() && it->second.mtime_ns == stamp && it->second.size == (long long)sb.st_size && it->second.bytes == want &&
        !it->second.from_memory) {
        Entry &e = it->second;
        e.held = true;
        e.tick = ++clock_;
        materialise(e, st);
        return e.d;
    }
    FILE *fp = fopen(fn.c_str(), "rb");
    if (!fp) throw IoError("cannot read observed data '" + fn + "'");
    HIP_OK(hipStreamSynchronize(st));  // h_io / xpose may still be in use
    size_t got = 0;
    if (fseeko(fp, (off_t)file_off, SEEK_SET) == 0) got = fread(h_.h_io, 1, want, fp);
    fclose(fp);
    if (got != want) throw IoError("short read on '" + fn + "'");
    Entry &e = obs_[shot_id];
    reset(e, want);
    e.held = true;
    e.tick = ++clock_;
    e.from_memory = false;
    e.size = (long long)sb.st_size;
    e.mtime_ns = stamp;
    if (!e.d) {
        make_room(want, st);
        HIP_OK(dev_malloc((void **)&e.d, want));
        dev_bytes_ += (long long)want;
    }
    HIP_OK(hipMemcpyAsync(h_.xpose, h_.h_io, want, hipMemcpyHostToDevice, st));
    fill_from_xpose(e, shot_id, nrec, st);
    return e.d;
}

}  // namespace sepfwi
