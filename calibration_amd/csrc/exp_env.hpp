// exp_env.hpp - experiment knobs (see the comment at engine.hpp's include of this file): environment variables that are read only
// by a library built with -DCBA_EXPERIMENTS (make EXPERIMENTS=1), never by the shipped one.
#pragma once
#include <cstdlib>

namespace cba {

inline const char* cba_exp_env(const char* name) {
#ifdef CBA_EXPERIMENTS
    return std::getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}

}  // namespace cba
