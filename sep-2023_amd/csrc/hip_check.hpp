// hip_check.hpp -- HIP_OK(call): a failed HIP runtime call becomes a HipError naming the call and its place.
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "errors.hpp"

#define HIP_OK(call)                                                                                          \
    do {                                                                                                      \
        hipError_t e_ = (call);                                                                               \
        if (e_ != hipSuccess)                                                                                 \
            throw ::sepfwi::HipError(std::string("HIP error '") + hipGetErrorString(e_) + "' at " + __FILE__ + ":" + \
                                     std::to_string(__LINE__) + " in " #call);                               \
    } while (0)
