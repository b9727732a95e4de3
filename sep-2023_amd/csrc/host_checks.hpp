// host_checks.hpp -- device-free validation code of the session: the index of a packed observed-data file and the survey's
// geometry against the computed grid.  Pure host C++ (no HIP), so that it runs under AddressSanitizer / UBSan on the CPU
// (tests/native/host_checks_sanitize.cpp); session.cpp / obs_store.cpp call it and turn its exceptions into error codes.
#pragma once
#include <map>
#include <stdexcept>
#include <string>
#include <utility>
#include <vector>

#include "config.hpp"
#include "errors.hpp"

namespace sepfwi {

// Index of a packed observed-data file (writer: sepfwi/utils.py pack_observed):
//   "SEPFWIP1" | int32 count | int32 nSteps | count x (int32 shot id, int32 nrec, int64 byte offset) | float32 gathers [nrec][nSteps]
// Every entry is checked where it is read: a corrupt index must not look like "shot not in the pack" (which silently falls back
// to Shot_ett{id}.bin) or surface later as a short read.  Throws IoError.
struct PackIndex {
    std::map<int, std::pair<long long, int>> entries;  // shot id -> (byte offset, nrec)
};
void read_pack_index(const std::string &fname, int nSteps, long long file_size, PackIndex *out);

// Flat cell index (z * pitch + x) of every receiver of every present shot, `rec_off[i]` = first entry of shot i (size nShots + 1;
// idx has one spare entry at the end).  Validates what the kernels assume: sources inside the updated region [2, n - 3]^2, every
// channel's stencil (one cell left for a horizontal fibre, one up for a vertical one, one in every direction for directional
// channels) inside the stored rows / columns.  Throws std::runtime_error naming the shot and receiver.
void receiver_cells(const Params &par, const Survey &survey, int nzc, int nx, int pitch, std::vector<int> *rec_off, std::vector<int> *idx);

}  // namespace sepfwi
