// capi.cpp -- the extern "C" surface declared in include/sepfwi.h.  Exceptions never cross it: every
// failure becomes an error code plus a thread-local message (the reference printf()s and exit(1)s,
// Src/utilities.h:28-36, which would take the Python interpreter down).
#include <cstdio>
#include <cstring>
#include <initializer_list>
#include <string>

#include "kernels.hpp"
#include "param_maps.hpp"
#include "session.hpp"

using namespace sepfwi;

static thread_local std::string t_last_error = "";

static int fail(int code, const std::string &msg) {
    t_last_error = msg;
    return code;
}

template <class Fn>
static int guarded(Fn &&fn) {
    try {
        fn();
        return SEPFWI_OK;
    } catch (const CourantError &e) {
        return fail(SEPFWI_ECOURANT, e.what());
    } catch (const IoError &e) {
        return fail(SEPFWI_EIO, e.what());
    } catch (const HipError &e) {
        return fail(SEPFWI_EHIP, e.what());
    } catch (const std::invalid_argument &e) {
        return fail(SEPFWI_EINVAL, e.what());
    } catch (const std::runtime_error &e) {
        const char *w = e.what();
        if (std::strncmp(w, "EIO:", 4) == 0) return fail(SEPFWI_EIO, w + 5);
        if (std::strncmp(w, "JSON", 4) == 0 || std::strstr(w, "JSON")) return fail(SEPFWI_EJSON, w);
        return fail(SEPFWI_EINVAL, w);
    } catch (const std::exception &e) {
        return fail(SEPFWI_EINVAL, e.what());
    } catch (...) {
        return fail(SEPFWI_EINVAL, "unknown failure");
    }
}

extern "C" {

const char *sepfwi_last_error(void) { return t_last_error.c_str(); }

int sepfwi_version(void) { return 100; }

int sepfwi_device_count(void) {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(SEPFWI_EHIP, std::string("hipGetDeviceCount: ") + hipGetErrorString(e));
    return n;
}

int sepfwi_cufd_stream(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf,
                       const float *Lambda, const float *Mu, const float *Den, const float *stf, int calc_id,
                       int gpu_id, int group_size, const int *shot_ids, const char *para_fname, void *hip_stream,
                       int async) {
    return guarded([&] {
        if (calc_id < 0 || calc_id > 3) throw std::invalid_argument("Invalid calc_id " + std::to_string(calc_id));  // libCUFD.cu:43-46
        if (!para_fname) throw std::invalid_argument("para_fname is NULL");
        if (!Lambda || !Mu || !Den || !stf) throw std::invalid_argument("Lambda, Mu, Den and stf must not be NULL");
        if (group_size < 0 || (group_size > 0 && !shot_ids)) throw std::invalid_argument("bad shot list");
        if (calc_id == 1 && (!grad_Lambda || !grad_Mu || !grad_Den)) throw std::invalid_argument("gradient outputs must not be NULL for calc_id 1");
        std::shared_ptr<Session> s = get_session(para_fname, gpu_id);
        s->run(misfit, grad_Lambda, grad_Mu, grad_Den, grad_stf, Lambda, Mu, Den, stf, calc_id, group_size, shot_ids,
              (hipStream_t)hip_stream, async != 0);
    });
}

int sepfwi_cufd(float *misfit, float *grad_Lambda, float *grad_Mu, float *grad_Den, float *grad_stf,
                const float *Lambda, const float *Mu, const float *Den, const float *stf, int calc_id, int gpu_id,
                int group_size, const int *shot_ids, const char *para_fname) {
    return sepfwi_cufd_stream(misfit, grad_Lambda, grad_Mu, grad_Den, grad_stf, Lambda, Mu, Den, stf, calc_id, gpu_id,
                              group_size, shot_ids, para_fname, nullptr, 0);
}

void sepfwi_release_all(void) {
    try { release_all_sessions(); } catch (...) {}
}

int sepfwi_set_observed(const char *para_fname, int gpu_id, int shot_id, const float *ett, int nrec, int nSteps) {
    return guarded([&] {
        if (!para_fname) throw std::invalid_argument("para_fname is NULL");
        get_session(para_fname, gpu_id)->set_observed(shot_id, ett, nrec, nSteps);
    });
}

void sepfwi_invalidate_observed(void) {
    try { invalidate_observed_all(); } catch (...) {}
}

int sepfwi_cpml_profiles(float *K, float *a, float *b, float *K_half, float *a_half, float *b_half, int N, int nPml,
                         float dh, float f0, float dt) {
    return guarded([&] {
        if (!K || !a || !b || !K_half || !a_half || !b_half || N <= 0 || nPml <= 0) throw std::invalid_argument("bad arguments");
        cpml_profiles(K, a, b, K_half, a_half, b_half, N, nPml, dh, f0, dt);
    });
}

int sepfwi_stf_taper(float *trace, int nt, float dt, float ratio) {
    return guarded([&] {
        if (!trace || nt <= 0) throw std::invalid_argument("bad arguments");
        if (!stf_taper(trace, nt, dt, ratio)) throw std::invalid_argument("Window error 2: taper longer than half the trace");
    });
}

int sepfwi_shot_split(int group_size, int ngpu, int *starts) {
    return guarded([&] {
        if (!starts || ngpu <= 0 || group_size < 0) throw std::invalid_argument("bad arguments");
        if (ngpu > group_size) throw std::invalid_argument("The number of GPUs should be smaller than the number of shots!");  // Torch_Fwi.cpp:49-52
        shot_split(group_size, ngpu, starts);
    });
}

int sepfwi_get_stats(const char *para_fname, int gpu_id, sepfwi_stats *out) {
    return guarded([&] {
        if (!para_fname || !out) throw std::invalid_argument("bad arguments");
        std::shared_ptr<Session> s = find_session(para_fname, gpu_id);
        if (!s) throw std::invalid_argument("no session for this parameter file / gpu");
        s->stats(out);
    });
}

int sepfwi_loop_status(const char *para_fname, int gpu_id, char *why, int len) {
    return guarded([&] {
        if (!para_fname || !why || len < 1) throw std::invalid_argument("bad arguments");
        std::shared_ptr<Session> s = find_session(para_fname, gpu_id);
        if (!s) throw std::invalid_argument("no session for this parameter file / gpu");
        const std::string w = s->loop_status();
        std::snprintf(why, (size_t)len, "%s", w.c_str());
    });
}

int sepfwi_debug_field(const char *para_fname, int gpu_id, int lane, int which, float *out) {
    return guarded([&] {
        if (!para_fname || !out) throw std::invalid_argument("bad arguments");
        std::shared_ptr<Session> s = find_session(para_fname, gpu_id);
        if (!s) throw std::invalid_argument("no session for this parameter file / gpu");
        s->copy_field(lane, which, out);
    });
}

// The calling thread's current HIP device, restored on scope exit: the parameterisation maps run on the autograd thread, whose
// current device torch's own guards read back with hipGetDevice.
struct DeviceRestore {
    int prev = -1;
    DeviceRestore() { if (hipGetDevice(&prev) != hipSuccess) { (void)hipGetLastError(); prev = -1; } }
    ~DeviceRestore() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// every pointer of the fused parameterisation maps must be device memory of ONE device (the maps run where the tensors live)
static int common_device(std::initializer_list<const void *> ptrs) {
    int dev = -1;
    for (const void *p : ptrs) {
        hipPointerAttribute_t attr;
        if (!p || hipPointerGetAttributes(&attr, p) != hipSuccess || (attr.type != hipMemoryTypeDevice && attr.type != hipMemoryTypeManaged)) {
            (void)hipGetLastError();
            throw std::invalid_argument("param maps: every array must be device memory (the host chain stays in torch)");
        }
        if (dev >= 0 && attr.device != dev) throw std::invalid_argument("param maps: arrays live on different devices");
        dev = attr.device;
    }
    return dev;
}

static void check_param_dims(int kind, int nz, int nx, int nPml, int nPad) {
    if (kind < 0 || kind >= PARAM_KINDS) throw std::invalid_argument("param maps: unknown parameterisation " + std::to_string(kind));
    if (nz < 1 || nx < 1 || nPml < 0 || nPad < 0) throw std::invalid_argument("param maps: bad sizes");
}

int sepfwi_param_forward(int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                         const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, float *Lambda, float *Mu,
                         float *Den, void *hip_stream) {
    return guarded([&] {
        check_param_dims(kind, nz, nx, nPml, nPad);
        const int dev = common_device({A, B, C, A_ref, B_ref, C_ref, Mask, Lambda, Mu, Den});
        DeviceRestore restore;
        if (hipSetDevice(dev) != hipSuccess) throw HipError("hipSetDevice failed");
        launch_param_fwd((hipStream_t)hip_stream, kind, nz, nx, nPml, nPad, A, B, C, A_ref, B_ref, C_ref, Mask, Lambda, Mu, Den);
        if (hipGetLastError() != hipSuccess) throw HipError("param map launch failed");
    });
}

int sepfwi_param_backward(int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                          const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, const float *gLambda,
                          const float *gMu, const float *gDen, float *gA, float *gB, float *gC, void *hip_stream) {
    return guarded([&] {
        check_param_dims(kind, nz, nx, nPml, nPad);
        const int dev = common_device({A, B, C, A_ref, B_ref, C_ref, Mask, gLambda, gMu, gDen, gA, gB, gC});
        DeviceRestore restore;
        if (hipSetDevice(dev) != hipSuccess) throw HipError("hipSetDevice failed");
        launch_param_bwd((hipStream_t)hip_stream, kind, nz, nx, nPml, nPad, A, B, C, A_ref, B_ref, C_ref, Mask, gLambda, gMu, gDen, gA,
                         gB, gC);
        if (hipGetLastError() != hipSuccess) throw HipError("param map launch failed");
    });
}

int sepfwi_get_option(const char *name) { return get_kernel_option(name); }

int sepfwi_set_option(const char *name, int value) {
    if (set_kernel_option(name, value) == 0) return SEPFWI_OK;
    return fail(SEPFWI_EINVAL, std::string("unknown option or bad value: '") + (name ? name : "") + "'");
}

}  // extern "C"
