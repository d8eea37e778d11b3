// tests/cpu_backend/seed_cpu.cpp — TEST-ONLY host run of seed_math.hpp (the per-view pose seed the GPU computes one wavefront
// per view) with the single-thread cooperative group.  Never linked into the product.
#include "../../calibration_amd/csrc/seed_math.hpp"

using namespace cba;

extern "C" void hm_planar_seed(int n, const double* X, const double* Y, const double* u, const double* v, const double* kmtx5, double* pose7) {
    SerialCoop co;
    planar_seed_view(n, X, Y, u, v, kmtx5, co, pose7);
}
