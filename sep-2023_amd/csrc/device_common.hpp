// device_common.hpp -- device-side helpers shared by the gfx950 kernel files.
#pragma once
#include <hip/hip_runtime.h>

#include "fwi_types.hpp"

namespace sepfwi {
namespace dev {

constexpr float C1 = 9.0f / 8.0f;   // el_stress.cu:42
constexpr float C2 = 1.0f / 24.0f;  // el_stress.cu:43
// backward-staggered first derivative D-:  (c1 (f0 - fm1) - c2 (fp1 - fm2)) / h
__device__ __forceinline__ float dminus(float fm2, float fm1, float f0, float fp1, float rh) {
    return (C1 * (f0 - fm1) - C2 * (fp1 - fm2)) * rh;
}
// forward-staggered first derivative D+:   (c1 (fp1 - f0) - c2 (fp2 - fm1)) / h
__device__ __forceinline__ float dplus(float fm1, float f0, float fp1, float fp2, float rh) {
    return (C1 * (fp1 - f0) - C2 * (fp2 - fm1)) * rh;
}

// Slot of cell (z,x) in the packed boundary frame, or -1.  The frame is the 5-cell-thick ring
// rows/cols [nPml-2, nPml+2] U [n-nPml-3, n-nPml+1] of the reference (utilities.cu:362-392) without
// its duplicated corners:  [top 5 rows][bottom 5 rows][middle rows: 5 left + 5 right cells].
__device__ __forceinline__ int frame_slot(const Grid &g, int z, int x) {
    const int zf = z - (g.nPml - 2), xf = x - (g.nPml - 2);
    if (zf < 0 || zf >= g.nzBnd || xf < 0 || xf >= g.nxBnd) return -1;
    if (zf < 5) return zf * g.nxBnd + xf;
    if (zf >= g.nzBnd - 5) return (5 + zf - (g.nzBnd - 5)) * g.nxBnd + xf;
    const int base = 10 * g.nxBnd + (zf - 5) * 10;
    if (xf < 5) return base + xf;
    if (xf >= g.nxBnd - 5) return base + 5 + (xf - (g.nxBnd - 5));
    return -1;
}

__device__ __forceinline__ bool in_pml_z(const Grid &g, int z) { return z < g.nPml || z > g.nzc - g.nPml - 1; }


}  // namespace dev
}  // namespace sepfwi
