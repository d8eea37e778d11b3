// structure.hpp — host-side description of one reprojection problem: sizes, CSR links, the layout
// of the SHARED (reduced) tangent space and the constant/gauge masks.  Pure host logic, shared by
// the HIP engine and the CPU test build.
//
// Mirrors how the reference builds its ceres::Problem:
//   intrinsics.cpp:63-90   one block per view: (c_q_t[v], c_t_t[v], intr); skew subset-constant
//   extrinsics.cpp:86-160  block (view,cam): (c_q_r[c], c_t_r[c], r_q_t[v], r_t_t[v], intr[c]);
//                          gauge: !optimize_intrinsics -> intr constant, else target pose 0 constant;
//                                 !optimize_extrinsics -> camera poses constant, else camera 0 constant
//   bundle.cpp:83-133      block: (b_q_t, b_t_t, g_q_c[c], g_t_c[c], intr[c]); three on/off switches
//
// Elimination structure: pose A of the INTRINSIC / EXTRINSIC chains is private to a view and is
// Schur-eliminated per view; everything else is "shared" and forms the reduced system:
//   INTRINSIC  shared = [intr(PI)]
//   EXTRINSIC  shared = per camera [c_T_r d(3) t(3) | intr(PI)]
//   BUNDLE     shared = [b_T_t d(3) t(3)] + per camera [g_T_c d(3) t(3) | intr(PI)]   (no private blocks)
#pragma once
#include <algorithm>
#include <cstdint>
#include <stdexcept>
#include <vector>

#include "../../include/calibba.h"

namespace cba {

struct Structure {
    int chain = 0, model = 0;
    int n_blocks = 0, n_cams = 0, n_views = 0;
    int64_t first_view_global = 0;
    int64_t n_obs = 0;
    int PI = 10;    // intrinsics per camera
    int PL = 16;    // local tangent columns per observation
    int NH = 136;   // PL (PL+1) / 2
    int NACC = 153; // NH + PL + 1
    int PSH = 10;   // shared-local columns of a block (PL - 6, or PL for BUNDLE)
    int PC = 10;    // shared columns per camera
    int sh_base = 0;  // first per-camera shared column (6 for BUNDLE: target pose first)
    int nsh = 10;     // size of the reduced system
    std::vector<int64_t> blk_offset;
    std::vector<int32_t> blk_cam, blk_view;
    // CSR: blocks of each private view, and view x cam -> block (-1 if absent)
    std::vector<int64_t> link_off;
    std::vector<int32_t> link_blk;
    std::vector<int32_t> view_cam_blk;
    // blocks of each camera in block order (deterministic per-camera reductions)
    std::vector<int64_t> cam_off;
    std::vector<int32_t> cam_blk;

    bool has_private() const { return chain != CBA_CHAIN_BUNDLE; }
    // packed index of (i <= j) in the upper triangle, row-major
    int hidx(int i, int j) const { return i * PL - i * (i - 1) / 2 + (j - i); }
    // global shared column of local column lc of a block with camera c (-1: private column)
    int shared_col(int c, int lc) const {
        if (chain == CBA_CHAIN_BUNDLE) return lc < 6 ? lc : sh_base + c * PC + (lc - 6);
        return lc < 6 ? -1 : c * PC + (lc - 6);
    }
};

// have_records: the observations arrive as per-block {X, Y, u, v} records (cba_reproj_create_aos), not through d.X .. d.v
inline void build_structure(const cba_reproj_problem& d, Structure& s, bool have_records = false) {
    if (d.chain < 0 || d.chain > 2) throw std::invalid_argument("unknown chain");
    if (d.camera_model < 0 || d.camera_model > 1) throw std::invalid_argument("unknown camera model");
    s.chain = d.chain; s.model = d.camera_model;
    s.PI = d.camera_model == CBA_CAMERA_SCHEIMPFLUG ? 12 : 10;
    s.PL = (d.chain == CBA_CHAIN_INTRINSIC ? 6 : 12) + s.PI;
    s.NH = s.PL * (s.PL + 1) / 2;
    s.NACC = s.NH + s.PL + 1;
    // validation mirroring the reference (SURVEY.md §8b "Errors")
    if (d.chain == CBA_CHAIN_BUNDLE) {
        if (d.n_cams <= 0) throw std::invalid_argument("No camera intrinsics provided");  // bundle.cpp:139-141
        if (d.n_blocks <= 0) throw std::invalid_argument("No observations provided");     // bundle.cpp:142-144
        if (!d.blk_b_T_g || !d.target_pose || !d.cam_pose) throw std::invalid_argument("bundle: missing pose arrays");
    }
    if (d.n_blocks < 0 || d.n_cams <= 0) throw std::invalid_argument("bad problem sizes");
    if (d.chain == CBA_CHAIN_INTRINSIC && d.n_cams != 1) throw std::invalid_argument("intrinsic chain takes exactly one camera");
    if (d.n_blocks > 0 && (!d.blk_offset || (!have_records && (!d.X || !d.Y || !d.u || !d.v)))) throw std::invalid_argument("missing observation arrays");
    if (!d.intr) throw std::invalid_argument("missing intrinsics");
    if (d.chain == CBA_CHAIN_EXTRINSIC && (!d.cam_pose || !d.blk_view || !d.blk_cam)) throw std::invalid_argument("extrinsic: missing arrays");
    if (d.chain != CBA_CHAIN_BUNDLE && d.n_views > 0 && !d.view_pose) throw std::invalid_argument("missing view poses");
    s.n_blocks = d.n_blocks; s.n_cams = d.n_cams;
    s.n_views = d.chain == CBA_CHAIN_BUNDLE ? 0 : d.n_views;
    s.first_view_global = d.first_view_global;
    s.blk_offset.assign(d.blk_offset, d.blk_offset + d.n_blocks + 1);
    s.blk_cam.resize(d.n_blocks); s.blk_view.resize(d.n_blocks);
    for (int b = 0; b < d.n_blocks; ++b) {
        // IntrinsicResidual::create etc. throw on an empty view (intrinsicresidual.h:38-40)
        if (s.blk_offset[b + 1] - s.blk_offset[b] <= 0) throw std::invalid_argument("No observations provided");
        const int c = d.blk_cam ? d.blk_cam[b] : 0;
        if (c < 0 || c >= d.n_cams) throw std::invalid_argument("camera index out of range");
        int v = -1;
        if (d.chain != CBA_CHAIN_BUNDLE) {
            v = d.blk_view ? d.blk_view[b] : b;
            if (v < 0 || v >= d.n_views) throw std::invalid_argument("view index out of range");
        }
        s.blk_cam[b] = c; s.blk_view[b] = v;
    }
    s.n_obs = d.n_blocks ? s.blk_offset[d.n_blocks] : 0;
    if (d.chain == CBA_CHAIN_INTRINSIC) { s.PSH = s.PI; s.PC = s.PI; s.sh_base = 0; s.nsh = s.PI; }
    else if (d.chain == CBA_CHAIN_EXTRINSIC) { s.PSH = 6 + s.PI; s.PC = 6 + s.PI; s.sh_base = 0; s.nsh = s.n_cams * s.PC; }
    else { s.PSH = s.PL; s.PC = 6 + s.PI; s.sh_base = 6; s.nsh = 6 + s.n_cams * s.PC; }
    // links
    s.link_off.assign(s.n_views + 1, 0);
    s.view_cam_blk.assign(static_cast<size_t>(s.n_views) * s.n_cams, -1);
    if (s.has_private()) {
        for (int b = 0; b < s.n_blocks; ++b) s.link_off[s.blk_view[b] + 1]++;
        for (int v = 0; v < s.n_views; ++v) s.link_off[v + 1] += s.link_off[v];
        s.link_blk.resize(s.n_blocks);
        std::vector<int64_t> cur(s.link_off.begin(), s.link_off.end() - 1);
        for (int b = 0; b < s.n_blocks; ++b) {
            const int v = s.blk_view[b];
            s.link_blk[cur[v]++] = b;
            int32_t& slot = s.view_cam_blk[static_cast<size_t>(v) * s.n_cams + s.blk_cam[b]];
            // two residual blocks on the same (view, camera) pair would need a merged Schur row; the
            // reference never produces that (views[v][c] is a single PlanarView)
            if (slot != -1) throw std::invalid_argument("duplicate (view, camera) residual block");
            slot = b;
        }
    }
    s.cam_off.assign(s.n_cams + 1, 0);
    for (int b = 0; b < s.n_blocks; ++b) s.cam_off[s.blk_cam[b] + 1]++;
    for (int c = 0; c < s.n_cams; ++c) s.cam_off[c + 1] += s.cam_off[c];
    s.cam_blk.resize(s.n_blocks);
    {
        std::vector<int64_t> cur(s.cam_off.begin(), s.cam_off.end() - 1);
        for (int b = 0; b < s.n_blocks; ++b) s.cam_blk[cur[s.blk_cam[b]]++] = b;
    }
}

// Mode B / R tile length for a problem of n_blocks residual blocks and n_obs observations (capi.cpp builds the tile tables with it).
// A tile pays one wave reduction and one partial row whatever its length (~880 instructions per wavefront against 440 per 128
// observations in the direct form of the one-camera pinhole chain, ~370 against 510 per 256 in the 4-wavefront forms: 8 - 11 % of a
// 2048-observation tile, and one more row per tile for the block sum), so longer tiles are cheaper - as long as the launch still fills
// the chip's workgroup slots (256 CUs x 8 wavefronts / the wavefronts of a workgroup) in whole rounds.  Picks the number of tiles per
// AVERAGE block that minimises  rounds x (passes per tile x loop + epilogue); never shorter than min_tile (2048).
inline int64_t choose_mode_b_tile(int64_t n_blocks, int64_t n_obs, bool two_wavefront_form, int64_t min_tile) {
    if (n_blocks <= 0) return min_tile;
    const int64_t np_obs = two_wavefront_form ? 128 : 256, loop = two_wavefront_form ? 440 : 510, epi = two_wavefront_form ? 880 : 370;
    const int64_t slots = 256 * (two_wavefront_form ? 4 : 2);
    const int64_t n_avg = std::max<int64_t>(1, n_obs / n_blocks);
    const int64_t kmax = std::max<int64_t>(1, (n_avg + min_tile - 1) / min_tile);
    int64_t best = -1, tile = min_tile;
    for (int64_t k = kmax; k >= 1; --k) {  // (ties go to the shorter tile)
        const int64_t len = ((n_avg + k - 1) / k + np_obs - 1) / np_obs * np_obs;
        const int64_t tiles = n_blocks * ((n_avg + len - 1) / len);
        const int64_t cost = ((tiles + slots - 1) / slots) * (len / np_obs * loop + epi);
        if (best < 0 || cost < best) { best = cost; tile = std::max<int64_t>(len, min_tile); }
    }
    return tile;
}

}  // namespace cba
