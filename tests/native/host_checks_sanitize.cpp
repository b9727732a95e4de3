// CPU-only sanitizer harness for the device-free validation code of the session (csrc/host_checks.cpp): the index reader of the
// packed observed-data file and the survey-geometry check that produces the receivers' flat cell indices.  Built with
// -fsanitize=address,undefined by tests/test_sanitizers.py.   argv[1] = seed, argv[2] = number of random cases, argv[3] = scratch dir
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>
#include <string>
#include <vector>

#include "../../sep-2023_amd/csrc/host_checks.hpp"

using namespace sepfwi;

static std::vector<unsigned char> make_pack(int nshots, int nSteps, const std::vector<int> &nrec) {
    std::vector<unsigned char> b;
    auto put = [&](const void *p, size_t n) { b.insert(b.end(), (const unsigned char *)p, (const unsigned char *)p + n); };
    put("SEPFWIP1", 8);
    int32_t head[2] = {nshots, nSteps};
    put(head, 8);
    int64_t off = 16 + 16LL * nshots;
    for (int k = 0; k < nshots; k++) {
        int32_t e[2] = {k * 3, nrec[k]};
        put(e, 8);
        put(&off, 8);
        off += (int64_t)nrec[k] * nSteps * 4;
    }
    b.resize((size_t)off, 0);
    return b;
}

static void write_file(const std::string &fn, const std::vector<unsigned char> &b) {
    FILE *fp = fopen(fn.c_str(), "wb");
    if (!fp) { printf("FAIL: cannot write %s\n", fn.c_str()); exit(1); }
    fwrite(b.data(), 1, b.size(), fp);
    fclose(fp);
}

int main(int argc, char **argv) {
    const unsigned seed = argc > 1 ? (unsigned)atoi(argv[1]) : 1u;
    const int n = argc > 2 ? atoi(argv[2]) : 500;
    const std::string dir = argc > 3 ? argv[3] : "/tmp";
    std::mt19937 rng(seed);
    const std::string fn = dir + "/pack_" + std::to_string(seed) + ".bin";
    int ok_packs = 0, bad_packs = 0, ok_surveys = 0, bad_surveys = 0;

    // ---- 1. a well-formed pack reads back exactly
    {
        const std::vector<int> nrec = {4, 0, 7};
        const auto b = make_pack(3, 11, nrec);
        write_file(fn, b);
        PackIndex ix;
        read_pack_index(fn, 11, (long long)b.size(), &ix);
        if (ix.entries.size() != 3 || ix.entries[6].second != 7 || ix.entries[0].first != 16 + 48 || ix.entries[6].first != 16 + 48 + 4 * 11 * 4) {
            printf("FAIL: well-formed pack mis-read\n");
            return 1;
        }
        bool threw = false;
        try { read_pack_index(fn, 12, (long long)b.size(), &ix); } catch (const IoError &) { threw = true; }
        if (!threw || !ix.entries.empty()) { printf("FAIL: wrong nSteps accepted\n"); return 1; }
    }
    // ---- 2. damaged packs: flipped bytes in magic / header / index, truncated files, lying file sizes
    for (int it = 0; it < n; it++) {
        const int ns = 1 + (int)(rng() % 6), nSteps = 1 + (int)(rng() % 40);
        std::vector<int> nrec(ns);
        for (int &r : nrec) r = (int)(rng() % 9);
        auto b = make_pack(ns, nSteps, nrec);
        const size_t head = 16 + 16 * (size_t)ns;
        const int nflip = (int)(rng() % 4);
        for (int k = 0; k < nflip; k++) b[rng() % head] ^= (unsigned char)(1u << (rng() % 8));
        if (rng() % 5 == 0) b.resize(rng() % (b.size() + 1));
        write_file(fn, b);
        long long claimed = (long long)b.size();
        if (rng() % 7 == 0) claimed = (long long)(rng() % (2 * b.size() + 2));
        PackIndex ix;
        try {
            read_pack_index(fn, nSteps, claimed, &ix);
            for (const auto &kv : ix.entries) {  // whatever was accepted lies inside the claimed file
                const long long want = (long long)kv.second.second * nSteps * 4;
                if (kv.second.second < 0 || kv.second.first < 0 || kv.second.first + want > claimed) { printf("FAIL: accepted entry outside the file\n"); return 1; }
            }
            ok_packs++;
        } catch (const IoError &) {
            if (!ix.entries.empty()) { printf("FAIL: index not cleared after an error\n"); return 1; }
            bad_packs++;
        }
    }
    remove(fn.c_str());
    {   // a missing file
        PackIndex ix;
        bool threw = false;
        try { read_pack_index(dir + "/no_such_pack.bin", 5, 100, &ix); } catch (const IoError &) { threw = true; }
        if (!threw) { printf("FAIL: missing pack accepted\n"); return 1; }
    }
    // ---- 3. survey geometry: sources and receivers on, at and beyond the edges of the stored grid
    for (int it = 0; it < n; it++) {
        Params par;
        par.fiber = (int)(rng() % 2);
        const int nzc = 6 + (int)(rng() % 30), nx = 6 + (int)(rng() % 90), pitch = ((nx + 63) / 64) * 64;
        Survey sv;
        const int ns = 1 + (int)(rng() % 4);
        sv.shots.resize(ns);
        auto coord = [&](int hi) { return (int)(rng() % (unsigned)(hi + 4)) - 2; };  // -2 .. hi + 1
        for (Shot &sh : sv.shots) {
            sh.present = rng() % 8 != 0;
            sh.z_src = (rng() % 3) ? 2 + (int)(rng() % (unsigned)(nzc - 4)) : coord(nzc);
            sh.x_src = (rng() % 3) ? 2 + (int)(rng() % (unsigned)(nx - 4)) : coord(nx);
            sh.nrec = (int)(rng() % 6);
            const bool tame = rng() % 2;
            for (int r = 0; r < sh.nrec; r++) {
                sh.z_rec.push_back(tame ? 1 + (int)(rng() % (unsigned)(nzc - 2)) : coord(nzc));
                sh.x_rec.push_back(tame ? 1 + (int)(rng() % (unsigned)(nx - 2)) : coord(nx));
            }
            if (rng() % 10 == 0 && sh.nrec > 0) sh.z_rec.pop_back();          // shorter list than nrec
            if (rng() % 4 == 0) sh.sens.assign((size_t)3 * sh.nrec - (rng() % 9 == 0 && sh.nrec > 0 ? 1 : 0), 0.5f);
        }
        std::vector<int> off, idx;
        try {
            receiver_cells(par, sv, nzc, nx, pitch, &off, &idx);
            if ((int)off.size() != ns + 1 || idx.size() != (size_t)off[ns] + 1) { printf("FAIL: table sizes\n"); return 1; }
            for (int i = 0; i < ns; i++)
                for (int k = off[i]; k < off[i + 1]; k++) {
                    const int z = idx[k] / pitch, x = idx[k] % pitch;
                    if (idx[k] < 0 || z >= nzc || x >= nx) { printf("FAIL: accepted receiver outside the grid\n"); return 1; }
                }
            ok_surveys++;
        } catch (const std::runtime_error &) {
            bad_surveys++;
        }
    }
    printf("OK packs %d accepted / %d refused, surveys %d accepted / %d refused\n", ok_packs, bad_packs, ok_surveys, bad_surveys);
    return 0;
}
