// errors.hpp -- exception types of the library; capi.cpp maps them to the SEPFWI_E* codes of include/sepfwi.h.
#pragma once
#include <stdexcept>

namespace sepfwi {

struct HipError : std::runtime_error { using std::runtime_error::runtime_error; };
struct IoError : std::runtime_error { using std::runtime_error::runtime_error; };
struct CourantError : std::runtime_error { using std::runtime_error::runtime_error; };

}  // namespace sepfwi
