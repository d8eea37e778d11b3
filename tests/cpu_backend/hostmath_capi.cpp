// tests/cpu_backend/hostmath_capi.cpp — TEST-ONLY host build of the product's device math.
//
// Compiles calibration_amd/csrc/reproj_math.hpp (the __host__ __device__ arithmetic the HIP kernels
// run per lane) with g++, so the analytic Jacobians can be checked against the oracle's dual
// numbers in the CPU-only test tier.  Not part of the product; never loaded by calibration_amd.
#include <cstddef>
#include <vector>

#include "../../calibration_amd/csrc/reproj_math.hpp"
#include "../../include/calibba.h"

using namespace cba;

template <int CHAIN, int MODEL>
static void eval_all(const cba_reproj_problem* d, double* r, double* J) {
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    for (int b = 0; b < d->n_blocks; ++b) {
        const int c = d->blk_cam ? d->blk_cam[b] : 0;
        const double* intr = d->intr + static_cast<size_t>(c) * PI;
        const double *pA, *pB = nullptr, *aux = nullptr;
        if (CHAIN == CH_INTRINSIC) pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view ? d->blk_view[b] : b);
        else if (CHAIN == CH_EXTRINSIC) { pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view[b]); pB = d->cam_pose + 7 * static_cast<size_t>(c); }
        else { pA = d->target_pose; pB = d->cam_pose + 7 * static_cast<size_t>(c); aux = d->blk_b_T_g + 12 * static_cast<size_t>(b); }
        double bc[BC_SIZE], sd[SD_SIZE] = {0};
        block_consts<CHAIN>(pA, pB, aux, bc);
        if (MODEL == CAM_SCHEIMPFLUG) scheimpflug_consts(intr, sd);
        for (int64_t i = d->blk_offset[b]; i < d->blk_offset[b + 1]; ++i) {
            double rr[2], Ju[PL], Jv[PL];
            reproj_point<CHAIN, MODEL>(bc, intr, sd, d->X[i], d->Y[i], d->u[i], d->v[i], rr, Ju, Jv);
            r[2 * i] = rr[0]; r[2 * i + 1] = rr[1];
            if (J) for (int k = 0; k < PL; ++k) { J[(2 * i) * PL + k] = Ju[k]; J[(2 * i + 1) * PL + k] = Jv[k]; }
            double r2[2];
            reproj_residual<MODEL>(bc, intr, sd, d->X[i], d->Y[i], d->u[i], d->v[i], r2);
            if (r2[0] != rr[0] || r2[1] != rr[1]) r[2 * i] = 1e300;  // residual-only path must agree bit for bit
        }
    }
}

// fp32 instantiation of the same arithmetic (BASELINE config 5 tolerance study): chain constants are
// built in fp64 and rounded once, observations are rounded to fp32, everything per observation is fp32.
template <int CHAIN, int MODEL>
static void eval_all_f32(const cba_reproj_problem* d, float* r, float* J) {
    constexpr int PI = IntrSize<MODEL>::value;
    constexpr int PL = LocalCols<CHAIN, MODEL>::value;
    for (int b = 0; b < d->n_blocks; ++b) {
        const int c = d->blk_cam ? d->blk_cam[b] : 0;
        const double* intr = d->intr + static_cast<size_t>(c) * PI;
        const double *pA, *pB = nullptr, *aux = nullptr;
        if (CHAIN == CH_INTRINSIC) pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view ? d->blk_view[b] : b);
        else if (CHAIN == CH_EXTRINSIC) { pA = d->view_pose + 7 * static_cast<size_t>(d->blk_view[b]); pB = d->cam_pose + 7 * static_cast<size_t>(c); }
        else { pA = d->target_pose; pB = d->cam_pose + 7 * static_cast<size_t>(c); aux = d->blk_b_T_g + 12 * static_cast<size_t>(b); }
        double bc[BC_SIZE], sd[SD_SIZE] = {0};
        block_consts<CHAIN>(pA, pB, aux, bc);
        if (MODEL == CAM_SCHEIMPFLUG) scheimpflug_consts(intr, sd);
        float bcf[BC_SIZE], sdf[SD_SIZE], inf[12];
        for (int k = 0; k < BC_SIZE; ++k) bcf[k] = static_cast<float>(bc[k]);
        for (int k = 0; k < SD_SIZE; ++k) sdf[k] = static_cast<float>(sd[k]);
        for (int k = 0; k < PI; ++k) inf[k] = static_cast<float>(intr[k]);
        for (int64_t i = d->blk_offset[b]; i < d->blk_offset[b + 1]; ++i) {
            float rr[2], Ju[PL], Jv[PL];
            reproj_point<CHAIN, MODEL, float>(bcf, inf, sdf, static_cast<float>(d->X[i]), static_cast<float>(d->Y[i]),
                                              static_cast<float>(d->u[i]), static_cast<float>(d->v[i]), rr, Ju, Jv);
            r[2 * i] = rr[0]; r[2 * i + 1] = rr[1];
            if (J) for (int k = 0; k < PL; ++k) { J[(2 * i) * PL + k] = Ju[k]; J[(2 * i + 1) * PL + k] = Jv[k]; }
        }
    }
}

extern "C" int hm_reproj_eval_f32(const cba_reproj_problem* d, float* r, float* J) {
    switch (d->chain * 2 + d->camera_model) {
        case 0: eval_all_f32<CH_INTRINSIC, CAM_PINHOLE_BC>(d, r, J); break;
        case 1: eval_all_f32<CH_INTRINSIC, CAM_SCHEIMPFLUG>(d, r, J); break;
        case 2: eval_all_f32<CH_EXTRINSIC, CAM_PINHOLE_BC>(d, r, J); break;
        case 3: eval_all_f32<CH_EXTRINSIC, CAM_SCHEIMPFLUG>(d, r, J); break;
        case 4: eval_all_f32<CH_BUNDLE, CAM_PINHOLE_BC>(d, r, J); break;
        case 5: eval_all_f32<CH_BUNDLE, CAM_SCHEIMPFLUG>(d, r, J); break;
        default: return 1;
    }
    return 0;
}

extern "C" int hm_reproj_eval(const cba_reproj_problem* d, double* r, double* J) {
    const int key = d->chain * 2 + d->camera_model;
    switch (key) {
        case 0: eval_all<CH_INTRINSIC, CAM_PINHOLE_BC>(d, r, J); break;
        case 1: eval_all<CH_INTRINSIC, CAM_SCHEIMPFLUG>(d, r, J); break;
        case 2: eval_all<CH_EXTRINSIC, CAM_PINHOLE_BC>(d, r, J); break;
        case 3: eval_all<CH_EXTRINSIC, CAM_SCHEIMPFLUG>(d, r, J); break;
        case 4: eval_all<CH_BUNDLE, CAM_PINHOLE_BC>(d, r, J); break;
        case 5: eval_all<CH_BUNDLE, CAM_SCHEIMPFLUG>(d, r, J); break;
        default: return 1;
    }
    return 0;
}
