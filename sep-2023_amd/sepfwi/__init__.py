"""sepfwi -- MI355X-native drop-in for the elastic FWI operator of seisfwi/SEP-2023 (TorchFWI-DAS).

Host side (Python, mirrors DAS_Waveform_Inversion/Ops/FWI of the reference):

    ops.fwi_ops            module object with forward / backward / obscalc   (Src/Torch_Fwi.cpp:138-142)
    ops.FWIFunction        the torch.autograd.Function of FWI_ops.py:46-63
    modules.FWI ...        parameterisation modules (FWI_ops.py:66-619): callers of the operator, fused HIP maps on GPU tensors
    utils                  paraGen / surveyGen / sourceGene / padding          (fwi_utils.py:11-140)
    obj_wrapper            SciPy L-BFGS-B glue                                 (obj_wrapper.py:10-97)
    propagator             Model / Survey / ElasticPropagator, the non-autograd caller (propagator.py:8-226, survey.py:3-38)
    dist                   one-process-per-GPU shot sharding + RCCL all-reduce

Device side: libsepfwi.so (csrc/, C ABI in include/sepfwi.h), hand-written HIP for gfx950.
"""
from . import _native  # noqa: F401
from .ops import FWIFunction, fwi_ops  # noqa: F401
from .modules import (FWI, FWI_obscalc, FWI_Lame_Den, FWI_IP_IS_Den, FWI_Vp_Vs_IP, FWI_Vp_Vs_IS,  # noqa: F401
                      FWI_Rock_Physics_VRH, FWI_Rock_Physics_gassmann)
