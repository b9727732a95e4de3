// conditioning.hpp -- data-conditioning chain of the misfit (conditioning.hip): windows, band-pass (hipFFT), normalised
// cross-correlation misfit.  All gathers are [rec][nt] float32 device arrays.
#pragma once
#include <hip/hip_runtime.h>

#include <map>

namespace sepfwi {

class Conditioner {
  public:
    Conditioner(int nt, int max_nrec);
    ~Conditioner();
    Conditioner(const Conditioner &) = delete;
    Conditioner &operator=(const Conditioner &) = delete;

    // cuda_window (utilities.cu:787-884): win_start == nullptr -> one end taper for all traces; else per-trace windows
    // [win_start, win_end] in seconds and amplitude weights[r] * src_weight.  ratio: libCUFD.cu:63 (0.005).
    void window(hipStream_t st, float *data, int nrec, float dt, const float *win_start, const float *win_end, const float *weights,
                float src_weight, float ratio);
    // bp_filter1d (utilities.cu:1115-1166): zero-phase sin^2 / cos^2 band-pass with corners filt[0..3] Hz, in place
    void bandpass(hipStream_t st, float *data, int nrec, float dt, const float filt[4]);
    // gpuMinus + cuda_cal_objective (utilities.cu:154-205): res = obs - syn (first sample zeroed), *acc += sum res^2
    void l2_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, double *acc);
    // cuda_find_normfact x3 + cuda_normal_misfit + cuda_normal_adjoint_source (utilities.cu:1010-1111):
    // *acc += -2 sum_r <obs,syn>_r / (|obs|_r |syn|_r) w_r ; res = the adjoint source.  weights may be null (all ones).
    void cross_residual(hipStream_t st, const float *obs, const float *syn, float *res, int nrec, const float *weights, float src_weight,
                        double *acc);
    // source_update (utilities.cu:1170-1281, cuda_spectrum_update :905-977): the source-signature update as a matching filter.
    // Both gathers zero-padded to 2 nt and end-tapered over the padded length (ratio 0.01); per frequency one coefficient
    // coef(f) = sum_r conj(C_r) O_r / (sum_r |C_r|^2 + 1e-6), kept for the adjoint step; the synthetic spectra are multiplied by
    // it, transformed back, cropped.  `syn` is updated in place, `obs` is only read.
    void source_update(hipStream_t st, const float *obs, float *syn, int nrec, float dt);
    // transpose of the map syn -> updated syn at the coefficients of the last source_update: pad, FFT, conj(coef), inverse FFT,
    // end taper of the padded length, crop.  (Conscious fix of source_update_adj, utilities.cu:1283-1325: oracle/oracle.py.)
    void source_update_adj(hipStream_t st, float *res, int nrec, float dt);
    // the second padded gather / spectrum and the coefficients of the source update, allocated (and the coefficients zeroed on `st`)
    // once; the session calls it up-front when the parameter file sets if_src_update, so that device_bytes() is complete and no
    // shot pays an allocation inside its time loop
    void ensure_source_buffers(hipStream_t st);
    long long device_bytes() const;

  private:
    struct Plans {
        void *fwd, *inv;  // hipfftHandle
    };
    Plans &plans_for(int nrec, hipStream_t st);
    int nt_, cap_;
    float *pad_ = nullptr, *norm_ = nullptr, *pad2_ = nullptr;
    void *spec_ = nullptr;  // hipfftComplex [cap][nt + 1]
    void *spec2_ = nullptr, *coef_ = nullptr;  // second padded gather / spectrum and the nt + 1 matching-filter coefficients (source update)
    std::map<int, Plans> plans_;  // by number of traces
};

}  // namespace sepfwi
