// dense.hpp — tiny host dense helpers for the REDUCED system (<= a few hundred unknowns): the part
// of the solve that is a handful of kFLOP per LM step and stays on the host next to the accept /
// reject control flow.
#pragma once
#include <algorithm>
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

#if defined(__x86_64__) && !defined(__HIP_DEVICE_COMPILE__)
#define CBA_DENSE_AVX2 1
// The 120-wide reduced system of an 8-camera rig is factorised once per LM step, on the critical path between the step's
// stream wait and the next launches (0.6 MFLOP: 60-90 us with dot4 — a tenth of a step when the problem is split over 8 GPUs).
// Blocked form for hosts with AVX2 + FMA (checked at run time; every rank of a job runs on the same kind of host, so the ranks'
// reduced solves stay bit-identical): columns in blocks of 4, a 4 x 2 register tile of dot products per pass, 4-wide over k.
typedef double dense_v4d __attribute__((vector_size(32)));
// out[r * stride + c] = sum_{k < kn} a_r[k] * b_c[k], kn a multiple of 4
__attribute__((target("avx2,fma"))) inline void dots42(const double* a0, const double* a1, const double* a2, const double* a3,
                                                       const double* b0, const double* b1, int kn, double* out, int stride) {
    dense_v4d c00 = {0, 0, 0, 0}, c01 = c00, c10 = c00, c11 = c00, c20 = c00, c21 = c00, c30 = c00, c31 = c00;
    for (int k = 0; k + 4 <= kn; k += 4) {
        dense_v4d x0, x1, x2, x3, y0, y1;
        __builtin_memcpy(&x0, a0 + k, 32); __builtin_memcpy(&x1, a1 + k, 32); __builtin_memcpy(&x2, a2 + k, 32);
        __builtin_memcpy(&x3, a3 + k, 32); __builtin_memcpy(&y0, b0 + k, 32); __builtin_memcpy(&y1, b1 + k, 32);
        c00 += x0 * y0; c01 += x0 * y1; c10 += x1 * y0; c11 += x1 * y1;
        c20 += x2 * y0; c21 += x2 * y1; c30 += x3 * y0; c31 += x3 * y1;
    }
#define CBA_HSUM(v) (((v)[0] + (v)[1]) + ((v)[2] + (v)[3]))
    out[0] = CBA_HSUM(c00); out[1] = CBA_HSUM(c01);
    out[stride] = CBA_HSUM(c10); out[stride + 1] = CBA_HSUM(c11);
    out[2 * stride] = CBA_HSUM(c20); out[2 * stride + 1] = CBA_HSUM(c21);
    out[3 * stride] = CBA_HSUM(c30); out[3 * stride + 1] = CBA_HSUM(c31);
#undef CBA_HSUM
}
__attribute__((target("avx2,fma"))) inline bool chol_inplace_blocked(double* A, int n) {
    const int nb = n & ~3;
    for (int j = 0; j < nb; j += 4) {
        const double* c0 = A + static_cast<size_t>(j) * n;
        const double *c1 = c0 + n, *c2 = c1 + n, *c3 = c2 + n;
        double inv[4] = {0, 0, 0, 0};
        for (int i = j; i < n; i += 4) {  // rows i .. i+3 against columns j .. j+3 (a short last group re-reads its first row)
            const int ir = n - i < 4 ? n - i : 4;
            const double* r[4];
            for (int q = 0; q < 4; ++q) r[q] = A + static_cast<size_t>(i + (q < ir ? q : 0)) * n;
            double acc[16];
            dots42(r[0], r[1], r[2], r[3], c0, c1, j, acc, 4);
            dots42(r[0], r[1], r[2], r[3], c2, c3, j, acc + 2, 4);
            if (i == j) {  // the 4 x 4 diagonal block: factorise
                for (int c = 0; c < 4; ++c) {
                    double* rc = A + static_cast<size_t>(j + c) * n;
                    double d = rc[j + c] - acc[c * 4 + c];
                    for (int k = 0; k < c; ++k) d -= rc[j + k] * rc[j + k];
                    if (!(d > 0.0) || !std::isfinite(d)) return false;
                    d = std::sqrt(d);
                    rc[j + c] = d;
                    inv[c] = 1.0 / d;
                    for (int q = c + 1; q < 4; ++q) {
                        double* rr = A + static_cast<size_t>(j + q) * n;
                        double sres = rr[j + c] - acc[q * 4 + c];
                        for (int k = 0; k < c; ++k) sres -= rr[j + k] * rc[j + k];
                        rr[j + c] = sres * inv[c];
                    }
                }
            } else {  // panel rows: forward substitution against the diagonal block (one reciprocal per column, not a division per entry)
                for (int q = 0; q < ir; ++q) {
                    double* rr = A + static_cast<size_t>(i + q) * n;
                    const double l0 = (rr[j] - acc[q * 4]) * inv[0];
                    const double l1 = (rr[j + 1] - acc[q * 4 + 1] - l0 * c1[j]) * inv[1];
                    const double l2 = (rr[j + 2] - acc[q * 4 + 2] - l0 * c2[j] - l1 * c2[j + 1]) * inv[2];
                    const double l3 = (rr[j + 3] - acc[q * 4 + 3] - l0 * c3[j] - l1 * c3[j + 1] - l2 * c3[j + 2]) * inv[3];
                    rr[j] = l0; rr[j + 1] = l1; rr[j + 2] = l2; rr[j + 3] = l3;
                }
            }
        }
    }
    for (int j = nb; j < n; ++j) {  // the last n mod 4 columns
        double* rj = A + static_cast<size_t>(j) * n;
        double d = rj[j] - dot4(rj, rj, j);
        if (!(d > 0.0) || !std::isfinite(d)) return false;
        d = std::sqrt(d);
        rj[j] = d;
        for (int i = j + 1; i < n; ++i) {
            double* ri = A + static_cast<size_t>(i) * n;
            ri[j] = (ri[j] - dot4(ri, rj, j)) / d;
        }
    }
    return true;
}
inline bool dense_has_avx2_fma() {
    static const bool ok = __builtin_cpu_supports("avx2") && __builtin_cpu_supports("fma");
    return ok;
}
#endif

// in-place lower Cholesky, row-major n x n; false if not positive definite
inline bool chol_inplace(std::vector<double>& A, int n) {
#ifdef CBA_DENSE_AVX2
    if (n >= 32 && dense_has_avx2_fma()) return chol_inplace_blocked(A.data(), n);
#endif
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
    for (int i = n - 1; i >= 0; --i) {  // L^T x = y by rows of L (contiguous), not by its columns
        const double* li = &L[static_cast<size_t>(i) * n];
        const double x = b[i] / li[i];
        b[i] = x;
        for (int k = 0; k < i; ++k) b[k] -= li[k] * x;
    }
}
// Extreme eigenvalues of a symmetric n x n matrix (row-major, destroyed) by cyclic Jacobi rotations: the reduced system of a
// covariance request is a few hundred wide at most and this runs once per request, off every hot path.
inline void sym_eig_minmax(std::vector<double>& A, int n, double* lmin, double* lmax) {
    for (int sweep = 0; sweep < 60; ++sweep) {
        double off = 0.0, dia = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) (i == j ? dia : off) += A[static_cast<size_t>(i) * n + j] * A[static_cast<size_t>(i) * n + j];
        if (off <= 1e-30 * dia) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = A[static_cast<size_t>(p) * n + q];
                if (apq == 0.0) continue;
                const double theta = (A[static_cast<size_t>(q) * n + q] - A[static_cast<size_t>(p) * n + p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < n; ++k) {  // columns p, q
                    const double akp = A[static_cast<size_t>(k) * n + p], akq = A[static_cast<size_t>(k) * n + q];
                    A[static_cast<size_t>(k) * n + p] = c * akp - sn * akq;
                    A[static_cast<size_t>(k) * n + q] = sn * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {  // rows p, q
                    const double apk = A[static_cast<size_t>(p) * n + k], aqk = A[static_cast<size_t>(q) * n + k];
                    A[static_cast<size_t>(p) * n + k] = c * apk - sn * aqk;
                    A[static_cast<size_t>(q) * n + k] = sn * apk + c * aqk;
                }
            }
    }
    double lo = A[0], hi = A[0];
    for (int i = 1; i < n; ++i) {
        lo = std::min(lo, A[static_cast<size_t>(i) * n + i]);
        hi = std::max(hi, A[static_cast<size_t>(i) * n + i]);
    }
    *lmin = lo;
    *lmax = hi;
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
