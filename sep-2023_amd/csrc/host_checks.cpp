// host_checks.cpp -- see host_checks.hpp.
#include "host_checks.hpp"

#include <cstdint>
#include <cstdio>
#include <cstring>

namespace sepfwi {

void read_pack_index(const std::string &fname, int nSteps, long long file_size, PackIndex *out) {
    out->entries.clear();
    FILE *fp = fopen(fname.c_str(), "rb");
    if (!fp) throw IoError("cannot read packed observed data '" + fname + "'");
    char magic[8];
    int32_t head[2] = {0, 0};
    bool ok = fread(magic, 1, 8, fp) == 8 && std::memcmp(magic, "SEPFWIP1", 8) == 0 && fread(head, 4, 2, fp) == 2 && head[0] >= 0;
    if (ok && head[1] != nSteps) {
        fclose(fp);
        throw IoError("packed observed data '" + fname + "' was written for another nSteps");
    }
    const long long head_bytes = 16 + 16LL * head[0];  // magic + (count, nSteps) + count x (id, nrec, offset)
    std::string bad;
    for (int k = 0; ok && k < head[0]; k++) {
        int32_t e[2];
        int64_t off;
        ok = fread(e, 4, 2, fp) == 2 && fread(&off, 8, 1, fp) == 1;
        if (!ok) break;
        const long long want = (long long)e[1] * (long long)nSteps * (long long)sizeof(float);
        if (e[1] < 0 || off < head_bytes || off > file_size || want > file_size - off)
            bad = "entry " + std::to_string(k) + " (shot " + std::to_string(e[0]) + ") points outside the file";
        else if (out->entries.count(e[0]))
            bad = "shot " + std::to_string(e[0]) + " is listed twice";
        if (!bad.empty()) break;
        out->entries[e[0]] = std::make_pair((long long)off, (int)e[1]);
    }
    fclose(fp);
    if (!bad.empty()) {
        out->entries.clear();
        throw IoError("packed observed data '" + fname + "': " + bad);
    }
    if (!ok) {
        out->entries.clear();
        throw IoError("'" + fname + "' is not a packed observed-data file");
    }
}

void receiver_cells(const Params &par, const Survey &survey, int nzc, int nx, int pitch, std::vector<int> *rec_off, std::vector<int> *idx) {
    const int ns = (int)survey.shots.size();
    rec_off->assign(ns + 1, 0);
    for (int i = 0; i < ns; i++) (*rec_off)[i + 1] = (*rec_off)[i] + (survey.shots[i].present ? survey.shots[i].nrec : 0);
    idx->assign((size_t)(*rec_off)[ns] + 1, 0);
    for (int i = 0; i < ns; i++) {
        const Shot &sh = survey.shots[i];
        if (!sh.present) continue;
        if ((int)sh.z_rec.size() < sh.nrec || (int)sh.x_rec.size() < sh.nrec)
            throw std::runtime_error("survey: shot " + std::to_string(i) + " lists fewer receiver coordinates than nrec");
        if (sh.z_src < 2 || sh.z_src > nzc - 3 || sh.x_src < 2 || sh.x_src > nx - 3)
            throw std::runtime_error("survey: source of shot " + std::to_string(i) + " lies outside the computed grid");
        const bool dir = !sh.sens.empty();  // directional channels reach one cell in every direction
        if (dir && (long long)sh.sens.size() < 3LL * sh.nrec)
            throw std::runtime_error("survey: shot " + std::to_string(i) + " lists fewer sensitivities than channels");
        for (int r = 0; r < sh.nrec; r++) {
            // the axial-strain difference reaches one cell to the left (horizontal fibre) or up (vertical fibre)
            if (sh.z_rec[r] < ((par.fiber || dir) ? 1 : 0) || sh.z_rec[r] >= nzc - (dir ? 1 : 0) || sh.x_rec[r] < ((par.fiber && !dir) ? 0 : 1) ||
                sh.x_rec[r] >= nx - (dir ? 1 : 0))
                throw std::runtime_error("survey: receiver " + std::to_string(r) + " of shot " + std::to_string(i) + " lies outside the grid");
            (*idx)[(size_t)(*rec_off)[i] + r] = sh.z_rec[r] * pitch + sh.x_rec[r];
        }
    }
}

}  // namespace sepfwi
