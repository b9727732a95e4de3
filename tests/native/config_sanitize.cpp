// CPU-only sanitizer harness for the host-side parsers and set-up math of libsepfwi (csrc/config.cpp,
// csrc/json_min.hpp): built with -fsanitize=address,undefined by tests/test_sanitizers.py (GPU AddressSanitizer is not
// available on the target pool, so the device-free host code is the part that gets sanitized).
//   argv[1] = seed, argv[2] = number of mutated documents
#include <cstdio>
#include <cstdlib>
#include <random>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../sep-2023_amd/csrc/config.hpp"

using namespace sepfwi;

static const char *PARA =
    "{\"nz\": 96, \"nx\": 80, \"dz\": 10.0, \"dx\": 10.0, \"nSteps\": 100, \"dt\": 0.001, \"f0\": 10.0, \"nPoints_pml\": 10, "
    "\"nPad\": 12, \"survey_fname\": \"s.json\", \"data_dir_name\": \"D\", \"das_fiber\": \"vertical\", \"if_win\": true, "
    "\"filter\": [0.0, 0.0, 2.0, 2.5], \"scratch_dir_name\": \"S\", \"if_src_update\": true, \"obs_pack_fname\": \"D/pack.bin\", \"obs_cache_mb\": 64, "
    "\"if_cross_misfit\": false}";
static const char *SURVEY =
    "{\"nShots\": 2, \"shot0\": {\"z_src\": 2, \"x_src\": 5, \"nrec\": 3, \"z_rec\": [4, 4, 4], \"x_rec\": [1, 2, 3], \"src_rxz\": 1.0, "
    "\"win_start\": [0.0, 0.01, 0.02], \"win_end\": [0.05, 0.06, 0.07], \"weights\": [1.0, 0.5, 2.0], \"src_weight\": 1.5, "
    "\"das_sensitivity\": [[1, 0, 0, 0, 0, 0], [0.5, 0.1, 0, 0.5, 0, 0], [0, 0, 0, 1, 0, 0]]}, "
    "\"shot1\": {\"z_src\": 2, \"x_src\": 9, \"nrec\": 2, \"z_rec\": [4, 4], \"x_rec\": [7, 8], \"win_start\": [0.0, 0.0], "
    "\"win_end\": [0.1, 0.1]}}";

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n = argc > 2 ? atoi(argv[2]) : 2000;
    // 1. the well-formed documents parse to the expected values
    Params p = parse_params(PARA);
    Survey s = parse_survey(SURVEY, p.nPml, p.if_win);
    if (!p.if_src_update || p.obs_pack_fname != "D/pack.bin" || p.obs_cache_mb != 64 || p.if_cross_misfit || p.nz != 96 || p.nx != 80 || p.fiber != 1 || !p.if_win || !p.has_filter || s.nShots != 2 || s.shots[1].nrec != 2 ||
        s.shots[0].x_rec[2] != 3 + p.nPml || p.filter[3] != 2.5f || s.shots[0].win_end[2] != 0.07f || s.shots[0].weights[1] != 0.5f ||
        s.shots[0].src_weight != 1.5f || s.shots[0].sens[3 * 1 + 1] != 0.5f || s.shots[0].sens[3 * 1 + 2] != 0.1f || !s.shots[1].weights.empty()) {
        printf("FAIL: reference documents mis-parsed\n");
        return 1;
    }
    {   // "conditioning": "reference": the four keys are parsed (and validated) and then ignored like the reference's driver does; what the
        // file said about if_win survives for the survey reader, which requires the windows whenever the key is set
        std::string pr(PARA);
        pr.insert(pr.rfind('}'), ", \"conditioning\": \"reference\"");
        Params q = parse_params(pr);
        if (!q.conditioning_reference || q.if_win || q.has_filter || q.if_src_update || q.if_cross_misfit || !q.if_win_key || q.nz != 96) {
            printf("FAIL: conditioning = reference mis-parsed\n");
            return 1;
        }
        std::string bad(PARA);
        bad.insert(bad.rfind('}'), ", \"conditioning\": \"sometimes\"");
        bool threw = false;
        try { parse_params(bad); } catch (const std::exception &) { threw = true; }
        if (!threw) { printf("FAIL: unknown conditioning value accepted\n"); return 1; }
    }
    // keys that merely start with "shot" are not shots; ids beyond nShots are ignored like the reference's reader does
    // (Src_Rec.cu:78 loops i < group_size over "shot" + to_string(id))
    {
        Survey q = parse_survey("{\"nShots\": 1, \"shots_meta\": {\"z_src\": 1}, \"shot-3\": 5, \"shot7\": {\"z_src\": 1}, "
                                "\"shot0\": {\"z_src\": 2, \"x_src\": 5, \"nrec\": 0, \"z_rec\": [], \"x_rec\": []}}", 10);
        if (q.shots.size() != 1 || !q.shots[0].present || q.shots[0].z_src != 12) {
            printf("FAIL: shot key validation\n");
            return 1;
        }
    }
    // 2. mutated documents either parse or throw std::exception -- never crash, overflow or leak
    std::mt19937 rng(seed);
    const std::string docs[2] = {PARA, SURVEY};
    const char alphabet[] = "{}[]:,\"\\ -+.eE0123456789tfnul\n\t\0x";
    int ok = 0, thrown = 0;
    for (int k = 0; k < n; k++) {
        std::string d = docs[k & 1];
        const int edits = 1 + (int)(rng() % 4);
        for (int e = 0; e < edits; e++) {
            const size_t pos = rng() % (d.size() + 1);
            switch (rng() % 4) {
                case 0: if (pos < d.size()) d.erase(pos, 1 + rng() % 8); break;
                case 1: d.insert(pos, 1, alphabet[rng() % (sizeof(alphabet) - 1)]); break;
                case 2: if (pos < d.size()) d[pos] = alphabet[rng() % (sizeof(alphabet) - 1)]; break;
                default: d.resize(pos); break;
            }
        }
        try {
            if (k & 1) {
                Survey sv = parse_survey(d, 10, (k & 2) != 0);
                (void)sv;
            } else {
                Params pp = parse_params(d);
                (void)pp;
            }
            ok++;
        } catch (const std::exception &) {
            thrown++;
        }
    }
    // 3. set-up math on awkward sizes
    for (int N : {12, 13, 64, 65, 2064}) {
        std::vector<float> K(N), a(N), b(N), Kh(N), ah(N), bh(N);
        cpml_profiles(K.data(), a.data(), b.data(), Kh.data(), ah.data(), bh.data(), N, N / 4 > 2 ? N / 4 : 2, 10.0f, 10.0f, 1e-3f);
    }
    for (int nt : {2, 3, 10, 1000, 4000}) {
        std::vector<float> tr(nt, 1.0f);
        stf_taper(tr.data(), nt, 1e-3f, 0.001f);
    }
    for (int ng = 1; ng <= 16; ng++) {
        std::vector<int> st(ng + 1);
        shot_split(37, ng, st.data());
        if (st[0] != 0 || st[ng] != 37) {
            printf("FAIL: shot_split\n");
            return 1;
        }
    }
    printf("OK parsed %d rejected %d\n", ok, thrown);
    return 0;
}
