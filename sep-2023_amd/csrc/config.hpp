// config.hpp -- parameter / survey description of one propagation setup.
// Same JSON schema as the reference (writer: fwi_utils.py:46-124; readers it replaces:
// Src/Parameter.cpp:17-178 and Src/Src_Rec.cu:20-281).
#pragma once
#include <string>
#include <vector>

namespace sepfwi {

struct Params {
    int nz = 0, nx = 0;  // padded grid sizes
    float dz = 0, dx = 0, dt = 0, f0 = 0;
    int nSteps = 0, nPml = 0, nPad = 0;
    std::string survey_fname, data_dir_name, scratch_dir_name;
    // optional key "obs_pack_fname" (SURVEY.md 8f-2): ONE packed file of the survey's observed axial-strain gathers (layout:
    // sepfwi/utils.py pack_observed) instead of the reference's four files per shot (libCUFD.cu:216-223)
    std::string obs_pack_fname;
    // optional key "obs_cache_mb": HBM budget [MB] of the session's observed-data store; the least recently used gathers beyond it
    // wait in pinned host memory (obs_store.hpp).  0 / absent: no budget (option "obs_cache_mb" of sepfwi_set_option applies).
    int obs_cache_mb = 0;
    // data-conditioning keys (dormant in the reference's driver, libCUFD.cu:353-457; live here, csrc/conditioning.hip)
    bool if_win = false, if_src_update = false, if_cross_misfit = false, has_filter = false;
    // optional key "conditioning": "live" (default) -- the four keys above switch their stage on; "reference" -- they are parsed
    // and validated like the reference does and then IGNORED like its driver does (every call site commented out,
    // libCUFD.cu:353-457): results are those of a file without them.  if_win_key keeps what the file said, because the survey
    // file must carry win_start / win_end whenever if_win is set, used or not (Src_Rec.cu:144-174).
    bool conditioning_reference = false, if_win_key = false;
    float filter[4] = {0, 0, 0, 0};  // band-pass corner frequencies [Hz]   (Parameter.cpp:147-159)
    // optional key "das_fiber": "horizontal" (default; recording_exx / res_injection_exx, libCUFD.cu:325,607) or
    // "vertical" (recording_ezz / res_injection_ezz, utilities.cu:620-641 -- in the reference a source edit)
    int fiber = 0;
};

struct Shot {
    int z_src = 0, x_src = 0;  // already shifted by +nPml (Src_Rec.cu:87-92)
    int nrec = 0;
    std::vector<int> z_rec, x_rec;  // shifted by +nPml (Src_Rec.cu:107-115)
    double src_rxz = 1.0;           // RSXXZZ default, Src_Rec.cu:259-264
    // optional key "das_sensitivity" (nrec x 6, the Numba solver's layout: column 0 weighs exx, 3 ezz, 1 exz,
    // MOD/elasticSolver.py:152-153,276): per-channel directional sensitivities, stored as (s_xx, s_zz, s_xz) triples.
    // Empty: straight fibre along x or z (para key "das_fiber").
    std::vector<float> sens;
    // with if_win: per-channel time windows [s] and trace weights, shot weight (Src_Rec.cu:144-200); weights default to 1
    std::vector<float> win_start, win_end, weights;
    float src_weight = 1.0f;
    bool present = false;
};

struct Survey {
    int nShots = 0;
    std::vector<Shot> shots;  // indexed by shot id ("shot%d" keys)
    int max_nrec = 0;
};

// Read the first line of a file (the reference only reads one line: Parameter.cpp:29, Src_Rec.cu:32).
// Throws std::runtime_error("EIO: ...") if the file cannot be opened.
std::string read_first_line(const std::string &fname);

// Parse; throw std::runtime_error on malformed input.
Params parse_params(const std::string &json_text);
Survey parse_survey(const std::string &json_text, int nPml, bool if_win = false);

// C-PML 1-D profiles (replaces cpmlInit, Src/utilities.cu:243-359).
void cpml_profiles(float *K, float *a, float *b, float *K_half, float *a_half, float *b_half, int N, int nPml,
                   float dh, float f0, float dt);

// sin^2 / cos^2 end taper of a trace (replaces the 5-argument cuda_window, Src/utilities.cu:844-884).
// Returns false (trace untouched) in the reference's "Window error 2" case.
bool stf_taper(float *trace, int nt, float dt, float ratio);

// Contiguous split of `group_size` shots over `ngpu` devices (Src/Torch_Fwi.cpp:59-60,78-80):
// starts[i] = (int) float32 linspace(0, group_size, ngpu+1)[i].
void shot_split(int group_size, int ngpu, int *starts);

}  // namespace sepfwi
