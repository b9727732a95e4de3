// config.cpp -- host-side setup math: JSON -> Params/Survey, C-PML profiles, source taper, shot split.
#include "config.hpp"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <stdexcept>

#include "json_min.hpp"

namespace sepfwi {

std::string read_first_line(const std::string &fname) {
    std::ifstream in(fname);
    if (!in.is_open()) throw std::runtime_error("EIO: cannot open '" + fname + "'");
    std::string line;
    std::getline(in, line);
    return line;
}

Params parse_params(const std::string &text) {
    JsonValue j = JsonReader(text).parse();
    if (j.kind != JsonValue::Object) throw std::runtime_error("parameter JSON is not an object");
    Params p;
    p.nz = j.at("nz").as_int("nz");
    p.nx = j.at("nx").as_int("nx");
    p.dz = (float)j.at("dz").as_number("dz");
    p.dx = (float)j.at("dx").as_number("dx");
    p.nSteps = j.at("nSteps").as_int("nSteps");
    p.nPml = j.at("nPoints_pml").as_int("nPoints_pml");
    p.nPad = j.at("nPad").as_int("nPad");
    p.dt = (float)j.at("dt").as_number("dt");
    p.f0 = (float)j.at("f0").as_number("f0");
    p.survey_fname = j.at("survey_fname").as_string("survey_fname");
    p.data_dir_name = j.at("data_dir_name").as_string("data_dir_name");
    if (j.has("scratch_dir_name")) p.scratch_dir_name = j.at("scratch_dir_name").as_string("scratch_dir_name");
    if (j.has("obs_pack_fname")) p.obs_pack_fname = j.at("obs_pack_fname").as_string("obs_pack_fname");
    if (j.has("obs_cache_mb")) {
        p.obs_cache_mb = j.at("obs_cache_mb").as_int("obs_cache_mb");
        if (p.obs_cache_mb < 0) throw std::runtime_error("parameter JSON: obs_cache_mb must be >= 0");
    }
    if (j.has("if_win")) p.if_win = j.at("if_win").as_bool("if_win");
    if (j.has("if_src_update")) p.if_src_update = j.at("if_src_update").as_bool("if_src_update");
    if (j.has("if_cross_misfit")) p.if_cross_misfit = j.at("if_cross_misfit").as_bool("if_cross_misfit");
    p.has_filter = j.has("filter");
    if (p.has_filter) {
        const JsonValue &f = j.at("filter");
        if (f.kind != JsonValue::Array || f.arr.size() != 4) throw std::runtime_error("parameter JSON: filter must be [f0, f1, f2, f3]");
        for (int k = 0; k < 4; k++) p.filter[k] = (float)f.arr[k].as_number("filter[]");
        if (!(p.filter[0] <= p.filter[1] && p.filter[1] <= p.filter[2] && p.filter[2] <= p.filter[3]))
            throw std::runtime_error("parameter JSON: filter corners must be ascending");
    }
    p.if_win_key = p.if_win;
    if (j.has("conditioning")) {
        const std::string c = j.at("conditioning").as_string("conditioning");
        if (c == "reference") {
            p.conditioning_reference = true;
            p.if_win = p.if_src_update = p.if_cross_misfit = p.has_filter = false;
        } else if (c != "live") {
            throw std::runtime_error("parameter JSON: conditioning must be \"live\" or \"reference\"");
        }
    }
    if (j.has("das_fiber")) {
        const std::string f = j.at("das_fiber").as_string("das_fiber");
        if (f == "vertical")
            p.fiber = 1;
        else if (f != "horizontal")
            throw std::runtime_error("parameter JSON: das_fiber must be \"horizontal\" or \"vertical\"");
    }
    if (p.nz <= 0 || p.nx <= 0 || p.nSteps < 2 || p.nPml < 2 || p.nPad < 0)
        throw std::runtime_error("parameter JSON: need nz,nx > 0, nSteps >= 2, nPoints_pml >= 2, nPad >= 0");
    if (p.nz - p.nPad - 2 * p.nPml < 6 || p.nx - 2 * p.nPml < 6)
        throw std::runtime_error("parameter JSON: physical grid (nz-nPad-2*nPml, nx-2*nPml) must be at least 6x6");
    if (p.nz > 32767 || p.nx > 32767) throw std::runtime_error("parameter JSON: nz and nx must be below 32768");
    if (!(p.dz > 0 && p.dx > 0 && p.dt > 0)) throw std::runtime_error("parameter JSON: dz, dx, dt must be positive");
    return p;
}

Survey parse_survey(const std::string &text, int nPml, bool if_win) {
    JsonValue j = JsonReader(text).parse();
    if (j.kind != JsonValue::Object) throw std::runtime_error("survey JSON is not an object");
    Survey s;
    s.nShots = j.at("nShots").as_int("nShots");
    for (const auto &kv : j.obj) {
        // "shot" followed by digits only ("shots_meta" or "shot-1" are not shots), id inside the declared range
        if (kv.first.size() < 5 || kv.first.size() > 12 || kv.first.compare(0, 4, "shot") != 0) continue;
        if (kv.first.find_first_not_of("0123456789", 4) != std::string::npos) continue;
        const long id_l = std::atol(kv.first.c_str() + 4);
        if (id_l >= (long)std::max(s.nShots, 0)) continue;  // the reference reads shot0 .. shot{nShots-1} only (Src_Rec.cu:74-77)
        const int id = (int)id_l;
        if ((int)s.shots.size() <= id) s.shots.resize(id + 1);
        const JsonValue &js = kv.second;
        Shot sh;
        sh.present = true;
        sh.z_src = js.at("z_src").as_int("z_src") + nPml;
        sh.x_src = js.at("x_src").as_int("x_src") + nPml;
        sh.nrec = js.at("nrec").as_int("nrec");
        const JsonValue &zr = js.at("z_rec"), &xr = js.at("x_rec");
        if (zr.kind != JsonValue::Array || xr.kind != JsonValue::Array || (int)zr.arr.size() < sh.nrec ||
            (int)xr.arr.size() < sh.nrec)
            throw std::runtime_error("survey JSON: z_rec/x_rec shorter than nrec for " + kv.first);
        sh.z_rec.resize(sh.nrec);
        sh.x_rec.resize(sh.nrec);
        for (int r = 0; r < sh.nrec; r++) {
            sh.z_rec[r] = zr.arr[r].as_int("z_rec[]") + nPml;
            sh.x_rec[r] = xr.arr[r].as_int("x_rec[]") + nPml;
        }
        if (js.has("src_rxz")) sh.src_rxz = js.at("src_rxz").as_number("src_rxz");
        auto floats = [&](const char *key, std::vector<float> &out) {
            const JsonValue &a = js.at(key);
            if (a.kind != JsonValue::Array || (int)a.arr.size() < sh.nrec)
                throw std::runtime_error(std::string("survey JSON: ") + key + " shorter than nrec for " + kv.first);
            out.resize(sh.nrec);
            for (int r = 0; r < sh.nrec; r++) out[r] = (float)a.arr[r].as_number(key);
        };
        if (if_win) {  // Src_Rec.cu:144-174: both arrays are mandatory with if_win
            floats("win_start", sh.win_start);
            floats("win_end", sh.win_end);
        }
        if (js.has("weights")) floats("weights", sh.weights);                                    // Src_Rec.cu:176-192
        if (js.has("src_weight")) sh.src_weight = (float)js.at("src_weight").as_number("src_weight");  // :195-200
        if (js.has("das_sensitivity")) {
            const JsonValue &ds = js.at("das_sensitivity");
            if (ds.kind != JsonValue::Array || (int)ds.arr.size() < sh.nrec)
                throw std::runtime_error("survey JSON: das_sensitivity must hold nrec rows of 6 numbers for " + kv.first);
            sh.sens.resize(3 * (size_t)sh.nrec);
            for (int r = 0; r < sh.nrec; r++) {
                const JsonValue &row = ds.arr[r];
                if (row.kind != JsonValue::Array || row.arr.size() != 6)
                    throw std::runtime_error("survey JSON: das_sensitivity rows need 6 numbers (exx, exz, -, ezz, -, -) for " + kv.first);
                sh.sens[3 * r + 0] = (float)row.arr[0].as_number("das_sensitivity[][0]");  // exx
                sh.sens[3 * r + 1] = (float)row.arr[3].as_number("das_sensitivity[][3]");  // ezz
                sh.sens[3 * r + 2] = (float)row.arr[1].as_number("das_sensitivity[][1]");  // exz
            }
        }
        if (sh.nrec > s.max_nrec) s.max_nrec = sh.nrec;
        s.shots[id] = std::move(sh);
    }
    return s;
}

// One side of the absorbing layer at distance `depth` (>= 0) into it.  Polynomial damping profile
// 0.25 d + 0.75 d^8, K grading to 2, alpha grading from pi*f0 to 0 (utilities.cu:248-260,281-286).
// Single-precision variables with double-precision intermediates, as the reference evaluates them.
namespace {
struct LayerPoint {
    float damp, K, alpha;
};
inline LayerPoint layer_point(float depth, float thickness, float d0, float alpha_max, bool k_with_powf) {
    const float dn = depth / thickness;
    const double dn8 = std::pow((double)dn, 8.0);
    LayerPoint r;
    r.damp = (float)((double)d0 * ((double)(0.25f * dn) + 0.75 * dn8 + 0.0 * std::pow((double)dn, 16.0)));
    r.K = k_with_powf ? (float)(1.0 + (2.0 - 1.0) * (double)powf(dn, 8.0f)) : (float)(1.0 + (2.0 - 1.0) * dn8);
    r.alpha = (float)((double)alpha_max * (1.0 - (double)dn));
    return r;
}
}  // namespace

void cpml_profiles(float *K, float *a, float *b, float *K_half, float *a_half, float *b_half, int N, int nPml,
                   float dh, float f0, float dt) {
    const float thickness = (float)nPml * dh;
    const float alpha_max = (float)(2.0 * 3.141592653589793238462643383279502884197169 * ((double)f0 / 2.0));
    // theoretical reflection coefficient 8e-4 at a fixed reference velocity of 3000 m/s (utilities.cu:248,260)
    const float d0 = (float)(-(double)(8.0f + 1.0f) * 3000.0 * std::log((double)0.0008f) / (2.0 * (double)thickness));
    for (int i = 0; i < N; i++) {
        float damp = 0.f, damp_h = 0.f, alpha = 0.f, alpha_h = 0.f, Ki = 1.f, Kh = 1.f;
        // near edge (index 0 side): integer points at nPml - i, half points at nPml - i - 1/2
        float depth = (float)(nPml - i) * dh;
        if (depth >= 0.0f) { LayerPoint p = layer_point(depth, thickness, d0, alpha_max, false); damp = p.damp; Ki = p.K; alpha = p.alpha; }
        depth = (float)(((double)(nPml - i) - 0.5) * (double)dh);
        if (depth >= 0.0f) { LayerPoint p = layer_point(depth, thickness, d0, alpha_max, false); damp_h = p.damp; Kh = p.K; alpha_h = p.alpha; }
        // far edge (index N-1 side)
        depth = (float)(nPml - N + i) * dh;
        if (depth >= 0.0f) { LayerPoint p = layer_point(depth, thickness, d0, alpha_max, false); damp = p.damp; Ki = p.K; alpha = p.alpha; }
        depth = (float)(((double)(nPml - N + i) + 0.5) * (double)dh);
        if (depth >= 0.0f) { LayerPoint p = layer_point(depth, thickness, d0, alpha_max, true); damp_h = p.damp; Kh = p.K; alpha_h = p.alpha; }
        if (alpha < 0.0f) alpha = 0.0f;
        if (alpha_h < 0.0f) alpha_h = 0.0f;
        K[i] = Ki;
        K_half[i] = Kh;
        b[i] = expf(-(damp / Ki + alpha) * dt);
        b_half[i] = expf(-(damp_h / Kh + alpha_h) * dt);
        a[i] = 0.0f;
        a_half[i] = 0.0f;
        if (std::fabs((double)damp) > 1.0e-6) a[i] = (float)((double)damp * ((double)b[i] - 1.0) / (double)(Ki * (damp + Ki * alpha)));
        if (std::fabs((double)damp_h) > 1.0e-6)
            a_half[i] = (float)((double)damp_h * ((double)b_half[i] - 1.0) / (double)(Kh * (damp_h + Kh * alpha_h)));
    }
}

bool stf_taper(float *trace, int nt, float dt, float ratio) {
    const float t_end = (float)nt * dt;
    const float ramp = (float)nt * dt * ratio;
    if (2.0 * (double)ramp >= (double)t_end) return false;
    const float t1 = ramp, t2 = t_end - ramp;
    const double half_pi = 3.141592653589793238462643383279502884197169 / 2.0;
    for (int k = 0; k < nt; k++) {
        const float t = (float)k * dt;
        float w;
        if (t >= 0.0f && t < t1) w = (float)std::sin(half_pi * (double)t / (double)t1);
        else if (t >= t1 && t < t2) w = 1.0f;
        else if (t >= t2 && t < t_end) w = (float)std::cos(half_pi * (double)(t - t2) / (double)(t_end - t2));
        else w = 0.0f;
        trace[k] *= w * w;
    }
    return true;
}

void shot_split(int group_size, int ngpu, int *starts) {
    // float32 torch::linspace(0, group_size, ngpu+1) then truncation by .item<int>()
    const int steps = ngpu + 1;
    const float start = 0.0f, end = (float)group_size;
    const float step = (steps > 1) ? (end - start) / (float)(steps - 1) : 0.0f;
    const int halfway = steps / 2;
    for (int i = 0; i < steps; i++) {
        // ATen's CPU kernel evaluates both branches with fused multiply-adds
        float v = (i < halfway) ? std::fmaf(step, (float)i, start) : std::fmaf(-step, (float)(steps - i - 1), end);
        starts[i] = (int)v;
    }
}

}  // namespace sepfwi
