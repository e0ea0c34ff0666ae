// mv_tuning.h -- environment knobs for the A/B tools under tools/ and for the tests that force an alternative kernel.
// Included by cpu-vision_amd/csrc/mv_common.h ONLY in -DMV_TUNING builds (libmi355vision_tuning.so and the named
// variants); the product library contains no environment lookup and decides from shapes alone.
#pragma once
#include <cstdlib>

namespace mv {
inline const char* tune_env(const char* name) { return std::getenv(name); }
}  // namespace mv
