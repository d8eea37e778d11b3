// dense.hpp — tiny host dense helpers for the REDUCED system (<= a few hundred unknowns): the part
// of the solve that is a handful of kFLOP per LM step and stays on the host next to the accept /
// reject control flow.
#pragma once
#include <cmath>
#include <vector>

namespace cba {

// sum_k a[k] b[k] with four independent partial sums in a FIXED order (k mod 4): a single accumulator makes the row-row dot
// products of the factorisation latency-bound (one multiply-add per 4 cycles); the 120-wide system of the 8-camera rig drops from
// 0.18 to 0.06 ms per LM step, on the critical path between two stream synchronisations.  No ISA-specific code: the result is the
// same on every host.
inline double dot4(const double* a, const double* b, int n) {
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    int k = 0;
    for (; k + 4 <= n; k += 4) {
        s0 += a[k] * b[k];
        s1 += a[k + 1] * b[k + 1];
        s2 += a[k + 2] * b[k + 2];
        s3 += a[k + 3] * b[k + 3];
    }
    for (; k < n; ++k) s0 += a[k] * b[k];
    return (s0 + s1) + (s2 + s3);
}

// in-place lower Cholesky, row-major n x n; false if not positive definite
inline bool chol_inplace(std::vector<double>& A, int n) {
    for (int j = 0; j < n; ++j) {
        const double* rj = &A[static_cast<size_t>(j) * n];
        double d = rj[j] - dot4(rj, rj, j);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        A[static_cast<size_t>(j) * n + j] = d;
        for (int i = j + 1; i < n; ++i) {
            double* ri = &A[static_cast<size_t>(i) * n];
            ri[j] = (ri[j] - dot4(ri, rj, j)) / d;
        }
    }
    return true;
}
inline void chol_solve(const std::vector<double>& L, int n, double* b) {
    for (int i = 0; i < n; ++i) b[i] = (b[i] - dot4(&L[static_cast<size_t>(i) * n], b, i)) / L[static_cast<size_t>(i) * n + i];
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
