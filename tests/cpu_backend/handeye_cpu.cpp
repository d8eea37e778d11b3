// tests/cpu_backend/handeye_cpu.cpp — TEST-ONLY host run of the AX = XB device math (axxb_math.hpp)
// and of the product's hand-eye LM driver (handeye_core.hpp).  Never linked into libcalibba.so.
#include <cmath>
#include <string>
#include <vector>

#include "../../calibration_amd/csrc/handeye_core.hpp"
#include "../../calibration_amd/csrc/seed_math.hpp"

using namespace cba;

namespace {
struct CpuAxxb final : AxxbEval {
    int n;
    std::vector<double> poses;
    CpuAxxb(int n_, const double* bTg, const double* cTt) : n(n_), poses(static_cast<size_t>(n_) * 24) {
        for (int k = 0; k < n; ++k) {
            double nb = 0, nc = 0, qb[4], qc[4];
            for (int a = 0; a < 4; ++a) { nb += bTg[7 * k + a] * bTg[7 * k + a]; nc += cTt[7 * k + a] * cTt[7 * k + a]; }
            for (int a = 0; a < 4; ++a) { qb[a] = bTg[7 * k + a] / std::sqrt(nb); qc[a] = cTt[7 * k + a] / std::sqrt(nc); }
            quat_to_rotmat(qb, &poses[24 * static_cast<size_t>(k)]);
            quat_to_rotmat(qc, &poses[24 * static_cast<size_t>(k) + 12]);
            for (int a = 0; a < 3; ++a) { poses[24 * static_cast<size_t>(k) + 9 + a] = bTg[7 * k + 4 + a]; poses[24 * static_cast<size_t>(k) + 21 + a] = cTt[7 * k + 4 + a]; }
        }
    }
    void eval(const double* pose7, double huber_delta, double* acc) override {
        double X[12];
        quat_to_rotmat(pose7, X);
        for (int a = 0; a < 3; ++a) X[9 + a] = pose7[4 + a];
        for (int e = 0; e < AXXB_NACC; ++e) acc[e] = 0.0;
        for (int i = 0; i + 1 < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                const double *pi = &poses[24 * static_cast<size_t>(i)], *pj = &poses[24 * static_cast<size_t>(j)];
                double RA[9], RB[9], tA[3], tB[3];
                if (!motion_pair(pi, pi + 9, pj, pj + 9, pi + 12, pi + 21, pj + 12, pj + 21, 0.5 * 3.14159265358979323846 / 180.0, 1e-3, RA, RB, tA, tB)) continue;
                double r[6], J[36];
                axxb_point(X, X + 9, RA, RB, tA, tB, r, J);
                axxb_accumulate(r, J, huber_delta, acc);
            }
    }
};
thread_local std::string g_err;
}  // namespace

extern "C" {
const char* hm_handeye_last_error(void) { return g_err.c_str(); }

void hm_axxb_eval(const double* q, const double* t, const double* RA, const double* RB, const double* tA, const double* tB,
                  double* r6, double* J66) {
    double RX[9];
    quat_to_rotmat(q, RX);
    axxb_point(RX, t, RA, RB, tA, tB, r6, J66);
}

// the product's partition of the AX = XB pairs over ranks (handeye_core.hpp)
void hm_axxb_rank_range(int n, int n_ranks, int rank, int* i0, int* i1) { axxb_rank_range(n, n_ranks, rank, i0, i1); }

// Tsai-Lenz all-pairs seed through the product's per-pair sums (axxb_math.hpp), serial; returns 0, 1 (no pairs) or 2 (singular)
int hm_handeye_dlt(int n, const double* bTg, const double* cTt, double min_angle_deg, double* pose7) {
    CpuAxxb ev(n, bTg, cTt);
    const double min_angle = min_angle_deg * 3.14159265358979323846 / 180.0;
    double w[3], RX[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, t[3];
    for (int mode = 0; mode < 2; ++mode) {
        double acc[AXXB_NACC] = {0};
        for (int i = 0; i + 1 < n; ++i)
            for (int j = i + 1; j < n; ++j) {
                const double *pi = &ev.poses[24 * static_cast<size_t>(i)], *pj = &ev.poses[24 * static_cast<size_t>(j)];
                double RA[9], RB[9], tA[3], tB[3];
                if (!motion_pair(pi, pi + 9, pj, pj + 9, pi + 12, pi + 21, pj + 12, pj + 21, min_angle, 1e-3, RA, RB, tA, tB)) continue;
                tsai_lenz_accumulate(mode, RA, RB, tA, tB, RX, acc);
            }
        if (acc[9] < 0.5) return 1;
        if (!tsai_lenz_solve(acc, 1e-12, mode == 0 ? w : t)) return 2;
        if (mode == 0) exp_so3(w, RX);
    }
    seed_rotmat_to_quat(RX, pose7);
    for (int k = 0; k < 3; ++k) pose7[4 + k] = t[k];
    return 0;
}

// number of pairs the product's filter keeps, and the pairs themselves ([RA RB tA tB] x 24) if out != NULL
int hm_build_pairs(int n, const double* bTg, const double* cTt, double* out) {
    CpuAxxb ev(n, bTg, cTt);
    int cnt = 0;
    for (int i = 0; i + 1 < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double *pi = &ev.poses[24 * static_cast<size_t>(i)], *pj = &ev.poses[24 * static_cast<size_t>(j)];
            double RA[9], RB[9], tA[3], tB[3];
            if (!motion_pair(pi, pi + 9, pj, pj + 9, pi + 12, pi + 21, pj + 12, pj + 21, 0.5 * 3.14159265358979323846 / 180.0, 1e-3, RA, RB, tA, tB)) continue;
            if (out) {
                double* o = out + 24 * static_cast<size_t>(cnt);
                for (int k = 0; k < 9; ++k) { o[k] = RA[k]; o[9 + k] = RB[k]; }
                for (int k = 0; k < 3; ++k) { o[18 + k] = tA[k]; o[21 + k] = tB[k]; }
            }
            ++cnt;
        }
    return cnt;
}

int hm_handeye_solve(int n, const double* bTg, const double* cTt, double* pose7, const cba_options* o, cba_summary* s, double* cov77) {
    try {
        if (n < 2) throw std::runtime_error("Inconsistent hand-eye input sizes");
        CpuAxxb ev(n, bTg, cTt);
        handeye_lm(ev, pose7, *o, s, cov77);
        return CBA_OK;
    } catch (const std::runtime_error& e) { g_err = e.what(); return CBA_ERR_RUNTIME; }
    catch (const std::exception& e) { g_err = e.what(); return CBA_ERR_INTERNAL; }
}
}
