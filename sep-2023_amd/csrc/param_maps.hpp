// param_maps.hpp -- fused parameterisation maps (param_maps.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace sepfwi {

// which triple of user parameters (A, B, C) the module inverts for; numbering is part of the C ABI (include/sepfwi.h)
enum ParamKind {
    PARAM_VP_VS_DEN = 0,   // FWI                FWI_ops.py:66-127
    PARAM_LAM_MU_DEN = 1,  // FWI_Lame_Den       FWI_ops.py:145-204
    PARAM_IP_IS_DEN = 2,   // FWI_IP_IS_Den      FWI_ops.py:208-266
    PARAM_VP_VS_IP = 3,    // FWI_Vp_Vs_IP       FWI_ops.py:270-330
    PARAM_VP_VS_IS = 4,    // FWI_Vp_Vs_IS       FWI_ops.py:333-393
    PARAM_ROCK_VRH = 5,       // FWI_Rock_Physics_VRH       FWI_ops.py:401-497   (A, B, C) = (porosity, clay content, water saturation)
    PARAM_ROCK_GASSMANN = 6,  // FWI_Rock_Physics_gassmann  FWI_ops.py:504-619
    PARAM_KINDS = 7
};

void launch_param_fwd(hipStream_t st, int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                      const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, float *Lam, float *Mu,
                      float *Den);
void launch_param_bwd(hipStream_t st, int kind, int nz, int nx, int nPml, int nPad, const float *A, const float *B, const float *C,
                      const float *A_ref, const float *B_ref, const float *C_ref, const float *Mask, const float *gLam,
                      const float *gMu, const float *gDen, float *gA, float *gB, float *gC);

}  // namespace sepfwi
