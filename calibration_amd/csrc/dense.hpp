// dense.hpp — tiny host dense helpers for the REDUCED system (<= a few hundred unknowns): the part
// of the solve that is a handful of kFLOP per LM step and stays on the host next to the accept /
// reject control flow.
#pragma once
#include <cmath>
#include <vector>

namespace cba {

// in-place lower Cholesky, row-major n x n; false if not positive definite
inline bool chol_inplace(std::vector<double>& A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[static_cast<size_t>(j) * n + j];
        for (int k = 0; k < j; ++k) d -= A[static_cast<size_t>(j) * n + k] * A[static_cast<size_t>(j) * n + k];
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[static_cast<size_t>(j) * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[static_cast<size_t>(i) * n + j];
            for (int k = 0; k < j; ++k) s -= A[static_cast<size_t>(i) * n + k] * A[static_cast<size_t>(j) * n + k];
            A[static_cast<size_t>(i) * n + j] = s / d;
        }
    }
    return true;
}
inline void chol_solve(const std::vector<double>& L, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[static_cast<size_t>(i) * n + k] * b[k];
        b[i] = s / L[static_cast<size_t>(i) * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[static_cast<size_t>(k) * n + i] * b[k];
        b[i] = s / L[static_cast<size_t>(i) * n + i];
    }
}
// inverse of an SPD matrix from its Cholesky factor
inline void chol_inverse(const std::vector<double>& L, int n, std::vector<double>& inv) {
    inv.assign(static_cast<size_t>(n) * n, 0.0);
    std::vector<double> e(n);
    for (int c = 0; c < n; ++c) {
        for (int i = 0; i < n; ++i) e[i] = 0.0;
        e[c] = 1.0;
        chol_solve(L, n, e.data());
        for (int r = 0; r < n; ++r) inv[static_cast<size_t>(r) * n + c] = e[r];
    }
}

}  // namespace cba
