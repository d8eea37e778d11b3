// tests/cpu_backend/backend_cpu.cpp — TEST-ONLY host implementation of cba::Backend.
//
// Runs the product's __host__ __device__ bodies (reproj_math.hpp, schur_math.hpp) in plain loops so
// the host LM driver (lm_core.hpp), the Schur algebra, the masks/gauge rules, the covariance
// assembly and the multi-rank all-reduce protocol can be tested without a GPU (gloo, world_size 2).
// It lives under tests/, is built by the test suite, and is never linked into libcalibba.so.
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../calibration_amd/csrc/lm_core.hpp"

using namespace cba;

namespace {

struct CpuBackend final : Backend {
    const Structure& s;
    const cba_reproj_problem& d;
    std::vector<double> intr[2], cam[2], target[2], view[2];
    std::vector<double> bc, sd, blk_acc, blk_w, blk_Z, vL, vy, vD, vgp, vscale2, vdelta;
    std::vector<int32_t> fixed;
    SchurDims dims;

    CpuBackend(const Structure& s_, const cba_reproj_problem& d_, const std::vector<double>& view0) : s(s_), d(d_) {
        dims = SchurDims{s.PL, s.NH, s.NACC, s.PSH, s.PC, s.n_cams, s.chain};
        for (int k = 0; k < 2; ++k) {
            intr[k].assign(static_cast<size_t>(s.n_cams) * s.PI, 0.0);
            cam[k].assign(static_cast<size_t>(s.n_cams) * 7, 0.0);
            target[k].assign(7, 0.0);
            view[k] = view0;
        }
        bc.assign(static_cast<size_t>(s.n_blocks) * BC_SIZE, 0.0);
        sd.assign(static_cast<size_t>(s.n_cams) * SD_SIZE, 0.0);
        blk_acc.assign(static_cast<size_t>(s.n_blocks) * s.NACC, 0.0);
        blk_w.assign(s.n_blocks, 1.0);
        blk_Z.assign(static_cast<size_t>(s.n_blocks) * 6 * s.PSH, 0.0);
        vL.assign(static_cast<size_t>(s.n_views) * 36, 0.0);
        vy.assign(static_cast<size_t>(s.n_views) * 6, 0.0);
        vD = vy; vgp = vy; vdelta = vy;
        vscale2.assign(static_cast<size_t>(s.n_views) * 6, 1.0);
        fixed.assign(s.n_views, 0);
    }
    void set_view_fixed(const std::vector<int32_t>& f) override { fixed = f; }
    void upload_shared(int which, const double* i, const double* c, const double* t) override {
        std::memcpy(intr[which].data(), i, sizeof(double) * intr[which].size());
        if (s.chain != CBA_CHAIN_INTRINSIC) std::memcpy(cam[which].data(), c, sizeof(double) * cam[which].size());
        if (s.chain == CBA_CHAIN_BUNDLE) std::memcpy(target[which].data(), t, sizeof(double) * 7);
    }
    void consts(int which) {
        for (int b = 0; b < s.n_blocks; ++b) {
            const int c = s.blk_cam[b];
            const double *pA, *pB = nullptr, *ax = nullptr;
            if (s.chain == CBA_CHAIN_INTRINSIC) pA = &view[which][7 * static_cast<size_t>(s.blk_view[b])];
            else if (s.chain == CBA_CHAIN_EXTRINSIC) { pA = &view[which][7 * static_cast<size_t>(s.blk_view[b])]; pB = &cam[which][7 * static_cast<size_t>(c)]; }
            else { pA = target[which].data(); pB = &cam[which][7 * static_cast<size_t>(c)]; ax = d.blk_b_T_g + 12 * static_cast<size_t>(b); }
            double* o = &bc[static_cast<size_t>(b) * BC_SIZE];
            if (s.chain == CBA_CHAIN_INTRINSIC) block_consts<CH_INTRINSIC>(pA, pB, ax, o);
            else if (s.chain == CBA_CHAIN_EXTRINSIC) block_consts<CH_EXTRINSIC>(pA, pB, ax, o);
            else block_consts<CH_BUNDLE>(pA, pB, ax, o);
        }
        if (s.model == CBA_CAMERA_SCHEIMPFLUG)
            for (int c = 0; c < s.n_cams; ++c) scheimpflug_consts(&intr[which][static_cast<size_t>(c) * 12], &sd[static_cast<size_t>(c) * SD_SIZE]);
    }
    template <int CHAIN, int MODEL>
    void mode_b(int which) {
        constexpr int PL = LocalCols<CHAIN, MODEL>::value;
        for (int b = 0; b < s.n_blocks; ++b) {
            double* acc = &blk_acc[static_cast<size_t>(b) * s.NACC];
            for (int e = 0; e < s.NACC; ++e) acc[e] = 0.0;
            const int c = s.blk_cam[b];
            for (int64_t i = s.blk_offset[b]; i < s.blk_offset[b + 1]; ++i) {
                double rr[2], Ju[PL], Jv[PL];
                reproj_point<CHAIN, MODEL>(&bc[static_cast<size_t>(b) * BC_SIZE], &intr[which][static_cast<size_t>(c) * s.PI],
                                           &sd[static_cast<size_t>(c) * SD_SIZE], d.X[i], d.Y[i], d.u[i], d.v[i], rr, Ju, Jv);
                int e = 0;
                for (int a = 0; a < PL; ++a)
                    for (int bb = a; bb < PL; ++bb) acc[e++] += Ju[a] * Ju[bb] + Jv[a] * Jv[bb];
                for (int a = 0; a < PL; ++a) acc[s.NH + a] += Ju[a] * rr[0] + Jv[a] * rr[1];
                acc[s.NH + PL] += rr[0] * rr[0] + rr[1] * rr[1];
            }
        }
    }
    static int UI_diag(int PL, int i) { return i * PL - i * (i - 1) / 2; }  // packed index of H[i][i]
    // the same blocks through the moment form of the two-pose chains (reproj_math.hpp: mom_point / pose_affine_G / mom_expand_entry)
    template <int CHAIN, int MODEL>
    void mode_b_moments(int which) {
        constexpr int PI = IntrSize<MODEL>::value;
        constexpr int NMOM = MomLayout<PI>::N;
        for (int b = 0; b < s.n_blocks; ++b) {
            double mom[NMOM];
            for (int e = 0; e < NMOM; ++e) mom[e] = 0.0;
            const int c = s.blk_cam[b];
            for (int64_t i = s.blk_offset[b]; i < s.blk_offset[b + 1]; ++i)
                mom_point<MODEL, 1, 0, double>(&bc[static_cast<size_t>(b) * BC_SIZE], &intr[which][static_cast<size_t>(c) * s.PI],
                                               &sd[static_cast<size_t>(c) * SD_SIZE], d.X[i], d.Y[i], d.u[i], d.v[i], mom);
            double G[3][36];
            pose_affine_G<CHAIN>(&bc[static_cast<size_t>(b) * BC_SIZE], G);
            double* acc = &blk_acc[static_cast<size_t>(b) * s.NACC];
            // the two-stage expansion k_mom_expand runs (T = Qh Gh, then 9 products per pose-pose entry), checked entry by entry
            // against the one-stage definition
            constexpr int PL = 12 + PI;
            static constexpr UpperIndex<PL> UI{};
            double T[9 * 12];
            for (int e = 0; e < 9 * 12; ++e) T[e] = mom_expand_T<PI>(mom, G, e / 12, e % 12);
            for (int e = 0; e < s.NACC; ++e) {
                acc[e] = mom_expand_entry_T<PI>(mom, G, T, e, e < s.NH ? UI.i[e] : 0, e < s.NH ? UI.j[e] : 0);
                const double one = mom_expand_entry<PI>(mom, G, e);
                const double scale = std::sqrt(std::fabs(acc[UI_diag(PL, e < s.NH ? UI.i[e] : 0)] * acc[UI_diag(PL, e < s.NH ? UI.j[e] : 0)]));
                if (std::fabs(acc[e] - one) > 1e-12 * std::max(scale, std::fabs(one)) && e < s.NH && UI.i[e] != UI.j[e])
                    throw std::runtime_error("two-stage moment expansion differs from the one-stage definition");
            }
        }
    }
    bool use_moments = false;
    double block_s(int which, int b) {
        const int c = s.blk_cam[b];
        double ss = 0;
        for (int64_t i = s.blk_offset[b]; i < s.blk_offset[b + 1]; ++i) {
            double rr[2];
            if (s.model == CBA_CAMERA_SCHEIMPFLUG)
                reproj_residual<CAM_SCHEIMPFLUG>(&bc[static_cast<size_t>(b) * BC_SIZE], &intr[which][static_cast<size_t>(c) * s.PI], &sd[static_cast<size_t>(c) * SD_SIZE], d.X[i], d.Y[i], d.u[i], d.v[i], rr);
            else
                reproj_residual<CAM_PINHOLE_BC>(&bc[static_cast<size_t>(b) * BC_SIZE], &intr[which][static_cast<size_t>(c) * s.PI], &sd[static_cast<size_t>(c) * SD_SIZE], d.X[i], d.Y[i], d.u[i], d.v[i], rr);
            ss += rr[0] * rr[0] + rr[1] * rr[1];
        }
        return ss;
    }
    void normal_eq(double hub, std::vector<double>& cam_acc, double cost2[2]) override { normal_eq_at(0, hub, cam_acc, cost2); }
    void normal_eq_at(int w, double hub, std::vector<double>& cam_acc, double cost2[2]) {
        consts(w);
        switch (s.chain * 2 + s.model) {
            case 0: mode_b<CH_INTRINSIC, CAM_PINHOLE_BC>(w); break;
            case 1: mode_b<CH_INTRINSIC, CAM_SCHEIMPFLUG>(w); break;
            case 2: if (use_moments) mode_b_moments<CH_EXTRINSIC, CAM_PINHOLE_BC>(w); else mode_b<CH_EXTRINSIC, CAM_PINHOLE_BC>(w); break;
            case 3: if (use_moments) mode_b_moments<CH_EXTRINSIC, CAM_SCHEIMPFLUG>(w); else mode_b<CH_EXTRINSIC, CAM_SCHEIMPFLUG>(w); break;
            case 4: if (use_moments) mode_b_moments<CH_BUNDLE, CAM_PINHOLE_BC>(w); else mode_b<CH_BUNDLE, CAM_PINHOLE_BC>(w); break;
            default: if (use_moments) mode_b_moments<CH_BUNDLE, CAM_SCHEIMPFLUG>(w); else mode_b<CH_BUNDLE, CAM_SCHEIMPFLUG>(w); break;
        }
        cost2[0] = cost2[1] = 0;
        cam_acc.assign(static_cast<size_t>(s.n_cams) * s.NACC, 0.0);
        for (int b = 0; b < s.n_blocks; ++b) {
            const double ss = blk_acc[static_cast<size_t>(b) * s.NACC + s.NH + s.PL];
            double rho, w;
            huber(ss, hub, &rho, &w);
            blk_w[b] = w;
            cost2[0] += 0.5 * rho;
            cost2[1] += ss;
        }
        for (int c = 0; c < s.n_cams; ++c)
            for (int64_t k = s.cam_off[c]; k < s.cam_off[c + 1]; ++k) {
                const int b = s.cam_blk[k];
                for (int e = 0; e < s.NACC; ++e) cam_acc[static_cast<size_t>(c) * s.NACC + e] += blk_w[b] * blk_acc[static_cast<size_t>(b) * s.NACC + e];
            }
    }
    void schur(double radius, bool init_scale, bool constrained, std::vector<double>& S, std::vector<double>& g, double* gmax_priv,
               int* nfail) override {
        schur_at(0, radius, init_scale, constrained, S, g, gmax_priv, nfail);
    }
    void schur_at(int w, double radius, bool init_scale, bool constrained, std::vector<double>& S, std::vector<double>& g,
                  double* gmax_priv, int* nfail) {
        const int n = s.nsh;
        S.assign(static_cast<size_t>(n) * n, 0.0);
        g.assign(n, 0.0);
        *gmax_priv = 0;
        *nfail = 0;
        for (int v = 0; v < s.n_views; ++v) {
            const int nb = static_cast<int>(s.link_off[v + 1] - s.link_off[v]);
            const int32_t* blks = s.link_blk.data() + s.link_off[v];
            double gm = 0;
            const bool ok = schur_view_body(dims, nb, blks, blk_acc.data(), blk_w.data(), fixed[v] != 0, radius, init_scale, constrained,
                                            &view[w][7 * static_cast<size_t>(v)], &vscale2[6 * static_cast<size_t>(v)], &vL[36 * static_cast<size_t>(v)],
                                            &vy[6 * static_cast<size_t>(v)], &vD[6 * static_cast<size_t>(v)], &vgp[6 * static_cast<size_t>(v)], blk_Z.data(), &gm);
            if (!ok) { ++*nfail; continue; }
            *gmax_priv = std::max(*gmax_priv, gm);
            // S += Zv^T Zv, g += Zv^T y
            for (int k1 = 0; k1 < nb; ++k1) {
                const int b1 = blks[k1];
                const double* Z1 = &blk_Z[static_cast<size_t>(b1) * 6 * s.PSH];
                const int g1 = s.blk_cam[b1] * s.PC;
                for (int c1 = 0; c1 < s.PSH; ++c1) {
                    double sy = 0;
                    for (int k = 0; k < 6; ++k) sy += Z1[k * s.PSH + c1] * vy[6 * static_cast<size_t>(v) + k];
                    g[g1 + c1] += sy;
                    for (int k2 = 0; k2 < nb; ++k2) {
                        const int b2 = blks[k2];
                        const double* Z2 = &blk_Z[static_cast<size_t>(b2) * 6 * s.PSH];
                        const int g2 = s.blk_cam[b2] * s.PC;
                        for (int c2 = 0; c2 < s.PSH; ++c2) {
                            double sum = 0;
                            for (int k = 0; k < 6; ++k) sum += Z1[k * s.PSH + c1] * Z2[k * s.PSH + c2];
                            S[static_cast<size_t>(g1 + c1) * n + g2 + c2] += sum;
                        }
                    }
                }
            }
        }
    }
    void trial(const double* delta_sh, double hub, TrialStats* st) override {
        *st = TrialStats();
        for (int v = 0; v < s.n_views; ++v) {
            const int nb = static_cast<int>(s.link_off[v + 1] - s.link_off[v]);
            double o4[4];
            backsub_view_body(dims, nb, s.link_blk.data() + s.link_off[v], s.blk_cam.data(), blk_Z.data(), delta_sh, fixed[v] != 0,
                              &vL[36 * static_cast<size_t>(v)], &vy[6 * static_cast<size_t>(v)], &vD[6 * static_cast<size_t>(v)],
                              &vgp[6 * static_cast<size_t>(v)], &view[0][7 * static_cast<size_t>(v)], &vdelta[6 * static_cast<size_t>(v)],
                              &view[1][7 * static_cast<size_t>(v)], o4);
            st->step2 += o4[0];
            st->xnorm2 += o4[1];
            st->gd += o4[2];
            st->dHd += o4[3];
        }
        consts(1);
        for (int b = 0; b < s.n_blocks; ++b) {
            double rho, w;
            huber(block_s(1, b), hub, &rho, &w);
            st->cost += 0.5 * rho;
        }
    }
    void accept() override { view[0] = view[1]; }
    // the speculative step (lm_core.hpp Backend::sys_step): statistics from the current factors, then the linearisation and
    // the elimination at the trial point; the current block sums and weights are kept until accept_step()
    std::vector<double> alt_acc, alt_w;
    bool sys_step(const double* delta_sh, double hub, double radius_next, bool constrained, const PackLayout& L, const AllReduce& ar,
                  int rank, double* pack) override {
        return sys_step_at(delta_sh, hub, radius_next, constrained, L, ar, rank, pack);
    }
    bool sys_step_at(const double* delta_sh, double hub, double radius_next, bool constrained, const PackLayout& L, const AllReduce& ar,
                     int rank, double* pack) {
        std::fill(pack, pack + L.size, 0.0);
        for (int v = 0; v < s.n_views; ++v) {
            const int nb = static_cast<int>(s.link_off[v + 1] - s.link_off[v]);
            double o4[4];
            backsub_view_body(dims, nb, s.link_blk.data() + s.link_off[v], s.blk_cam.data(), blk_Z.data(), delta_sh, fixed[v] != 0,
                              &vL[36 * static_cast<size_t>(v)], &vy[6 * static_cast<size_t>(v)], &vD[6 * static_cast<size_t>(v)],
                              &vgp[6 * static_cast<size_t>(v)], &view[0][7 * static_cast<size_t>(v)], &vdelta[6 * static_cast<size_t>(v)],
                              &view[1][7 * static_cast<size_t>(v)], o4);
            pack[L.stats + PackLayout::STEP2] += o4[0];
            pack[L.stats + PackLayout::XNORM2] += o4[1];
            pack[L.stats + PackLayout::GD] += o4[2];
            pack[L.stats + PackLayout::DHD] += o4[3];
        }
        const std::vector<double> keep_acc = blk_acc, keep_w = blk_w;
        std::vector<double> cam_acc, S, g;
        double cost2[2], gm = 0;
        int nf = 0;
        normal_eq_at(1, hub, cam_acc, cost2);
        schur_at(1, radius_next, false, constrained, S, g, &gm, &nf);
        alt_acc = blk_acc; alt_w = blk_w;
        blk_acc = keep_acc; blk_w = keep_w;
        pack[L.stats + PackLayout::TRIAL_COST] = cost2[0];
        std::copy(cam_acc.begin(), cam_acc.end(), pack + L.cam);
        pack[L.cost] = cost2[0];
        pack[L.nfail] = nf;
        L.pack_S(S, pack + L.S);
        std::copy(g.begin(), g.end(), pack + L.g);
        pack[L.gmax + rank] = gm;
        ar(pack, L.size);
        return true;
    }
    void accept_step() override { blk_acc = alt_acc; blk_w = alt_w; view[0] = view[1]; }
    void line_eval(double a, double hub, bool want_slope, const PackLayout& L, const AllReduce& ar, int rank, double* pack) override {
        (void)rank;
        std::fill(pack, pack + L.size, 0.0);
        for (int v = 0; v < s.n_views; ++v) {
            double o2[2];
            scale_step_view_body(fixed[v] != 0, a, &view[0][7 * static_cast<size_t>(v)], &vdelta[6 * static_cast<size_t>(v)],
                                 &view[1][7 * static_cast<size_t>(v)], o2);
            pack[L.stats + PackLayout::STEP2] += o2[0];
            pack[L.stats + PackLayout::XNORM2] += o2[1];
        }
        if (want_slope) {
            const std::vector<double> keep_acc = blk_acc, keep_w = blk_w;
            std::vector<double> cam_acc;
            double cost2[2];
            normal_eq_at(1, hub, cam_acc, cost2);
            for (int v = 0; v < s.n_views; ++v)
                pack[L.stats + PackLayout::SLOPE] += view_slope_body(dims, static_cast<int>(s.link_off[v + 1] - s.link_off[v]),
                                                                      s.link_blk.data() + s.link_off[v], blk_acc.data(), blk_w.data(),
                                                                      fixed[v] != 0, &vdelta[6 * static_cast<size_t>(v)]);
            std::copy(cam_acc.begin(), cam_acc.end(), pack + L.cam);
            pack[L.stats + PackLayout::TRIAL_COST] = cost2[0];
            blk_acc = keep_acc; blk_w = keep_w;
        } else {
            consts(1);
            double c = 0.0;
            for (int b = 0; b < s.n_blocks; ++b) {
                double rho, w;
                huber(block_s(1, b), hub, &rho, &w);
                c += 0.5 * rho;
            }
            pack[L.stats + PackLayout::TRIAL_COST] = c;
        }
        ar(pack, L.size);
    }
    void download_private(double* vp) override { std::memcpy(vp, view[0].data(), sizeof(double) * view[0].size()); }
    void download_blocks(std::vector<double>& acc, std::vector<double>& w) override { acc = blk_acc; w = blk_w; }

    // ---- the controller form of the iteration: lm_ctl.hpp run by a team of one host thread -----------------------------------
    struct Ctl {
        std::vector<double> scal, camc, hdiag, gc, scale2, xs, rdiag, Ld, x[3], A, pack, rec;
        std::vector<int> colcam, collc;
        std::vector<int8_t> eff, active, cam_var;
        std::vector<int> idx;
        double lmp[2] = {1e4, 1.0};
        int okflag = 1;
        CtlView V;
        bool constrained = false;
    } ctl;
    bool use_ctl = true;
    void ctl_sync_params() {  // the controller's packs -> the per-copy arrays the evaluation reads
        for (int k = 0; k < 2; ++k) {
            std::memcpy(intr[k].data(), ctl.x[k].data(), sizeof(double) * intr[k].size());
            std::memcpy(cam[k].data(), ctl.x[k].data() + ctl.V.pk_cam, sizeof(double) * cam[k].size());
            std::memcpy(target[k].data(), ctl.x[k].data() + ctl.V.pk_target, sizeof(double) * 7);
        }
    }
    void ctl_invoke(int mode, int flag) {
        SerialTeam tm;
        ctl_run(tm, ctl.V, mode, flag);
    }
    bool ctl_begin(const CtlSetup& cs, const PackLayout& L) override {
        if (!use_ctl) return false;
        const int n = s.nsh;
        CtlView& V = ctl.V;
        V.n = n; V.n_cams = s.n_cams; V.PI = s.PI; V.PL = s.PL; V.NH = s.NH; V.NACC = s.NACC; V.PC = s.PC; V.sh_base = s.sh_base;
        V.chain = s.chain; V.n_ranks = L.n_ranks;
        V.off_stats = L.stats; V.off_cam = L.cam; V.off_cost = L.cost; V.off_nfail = L.nfail; V.off_S = L.S; V.off_g = L.g; V.off_gmax = L.gmax;
        V.eps = cs.eps; V.max_iterations = cs.max_iterations; V.constrained = cs.constrained; V.line_search = cs.line_search;
        V.speculate = cs.speculate; V.intr_var = cs.intr_var; V.target_var = cs.target_var;
        V.pk_cam = static_cast<int64_t>(intr[0].size()); V.pk_target = V.pk_cam + static_cast<int64_t>(cam[0].size()); V.pk_delta = V.pk_target + 7;
        ctl.constrained = cs.constrained;
        ctl.scal.assign(CS_COUNT, 0.0); ctl.rec.assign(CS_COUNT, 0.0);
        ctl_reset(ctl.scal.data());
        const int M8 = ctl_padded(n);
        ctl.camc.assign(static_cast<size_t>(s.n_cams) * s.NACC, 0.0); ctl.hdiag.assign(n, 0.0); ctl.gc.assign(n, 0.0); ctl.scale2.assign(n, 1.0); ctl.xs.assign(M8, 0.0);
        ctl.rdiag.assign(M8, 0.0); ctl.Ld.assign(static_cast<size_t>(M8) * CTL_NB, 0.0);
        for (int k = 0; k < 3; ++k) ctl.x[k].assign(static_cast<size_t>(V.pk_delta) + n, 0.0);
        V.lda = ctl_lda(n);
        ctl.A.assign(static_cast<size_t>(M8 + 1) * V.lda, 0.0);
        ctl.pack.assign(static_cast<size_t>(L.size), 0.0);
        ctl.eff.assign(n, 0); ctl.idx.assign(n, 0);
        ctl.active.assign(cs.active->begin(), cs.active->end());
        ctl.cam_var.assign(cs.cam_var->begin(), cs.cam_var->end());
        std::memcpy(ctl.x[0].data(), cs.intr, sizeof(double) * intr[0].size());
        if (s.chain != CBA_CHAIN_INTRINSIC) std::memcpy(ctl.x[0].data() + V.pk_cam, cs.cam, sizeof(double) * cam[0].size());
        if (s.chain == CBA_CHAIN_BUNDLE) std::memcpy(ctl.x[0].data() + V.pk_target, cs.target, sizeof(double) * 7);
        ctl.x[1] = ctl.x[0];
        ctl.lmp[0] = 1e4; ctl.lmp[1] = 1.0;
        V.x_cur = ctl.x[0].data(); V.x_trial = ctl.x[1].data(); V.x_tmp = ctl.x[2].data();
        ctl.colcam.assign(n, 0); ctl.collc.assign(n, 0);
        for (int i = 0; i < n; ++i) ctl_decode(V, i, &ctl.colcam[i], &ctl.collc[i]);
        V.colcam = ctl.colcam.data(); V.collc = ctl.collc.data();
        V.scal = ctl.scal.data(); V.pack = ctl.pack.data(); V.camc = ctl.camc.data(); V.hdiag = ctl.hdiag.data(); V.gc = ctl.gc.data(); V.scale2 = ctl.scale2.data();
        V.eff = ctl.eff.data(); V.active = ctl.active.data(); V.cam_var = ctl.cam_var.data(); V.idx = ctl.idx.data();
        V.A = ctl.A.data(); V.rdiag = ctl.rdiag.data(); V.xs = ctl.xs.data(); V.Ld = ctl.Ld.data(); V.lmp = ctl.lmp; V.rec = ctl.rec.data(); V.okflag = &ctl.okflag;
        return true;
    }
    void ctl_new(double hub, bool first, const PackLayout& L, const AllReduce& ar, int rank) override {
        ctl_sync_params();
        std::vector<double> cam_acc, S, g;
        double cost2[2] = {0, 0}, gm = 0;
        int nf = 0;
        normal_eq_at(0, hub, cam_acc, cost2);
        schur_at(0, ctl.lmp[0], ctl.lmp[1] != 0.0, ctl.constrained, S, g, &gm, &nf);
        double* pack = ctl.pack.data();
        std::fill(pack, pack + L.size, 0.0);
        std::copy(cam_acc.begin(), cam_acc.end(), pack + L.cam);
        pack[L.cost] = cost2[0];
        pack[L.nfail] = nf;
        L.pack_S(S, pack + L.S);
        std::copy(g.begin(), g.end(), pack + L.g);
        pack[L.gmax + rank] = gm;
        ar(pack + L.cam, L.size - L.cam);
        ctl_invoke(CTL_NEW, first ? 1 : 0);
    }
    void ctl_resolve(const PackLayout& L, const AllReduce& ar, int rank) override {
        (void)rank;
        std::vector<double> S, g;
        double gm = 0;
        int nf = 0;
        schur_at(0, ctl.lmp[0], false, ctl.constrained, S, g, &gm, &nf);
        double* pack = ctl.pack.data();
        pack[L.nfail] = nf;
        L.pack_S(S, pack + L.S);
        std::copy(g.begin(), g.end(), pack + L.g);
        ar(pack + L.nfail, L.gmax - L.nfail);
        ctl_invoke(CTL_RESOLVED, 0);
    }
    void ctl_step(double hub, bool speculative, const PackLayout& L, const AllReduce& ar, int rank) override {
        ctl_sync_params();
        const double* delta_sh = ctl.x[1].data() + ctl.V.pk_delta;
        if (speculative) {
            sys_step_at(delta_sh, hub, ctl.lmp[0], ctl.constrained, L, ar, rank, ctl.pack.data());
        } else {
            TrialStats st;
            trial(delta_sh, hub, &st);
            double* pack = ctl.pack.data();
            for (int k = 0; k < 6; ++k) pack[L.stats + k] = 0.0;
            pack[L.stats + PackLayout::GD] = st.gd; pack[L.stats + PackLayout::DHD] = st.dHd;
            pack[L.stats + PackLayout::STEP2] = st.step2; pack[L.stats + PackLayout::XNORM2] = st.xnorm2;
            pack[L.stats + PackLayout::TRIAL_COST] = st.cost;
            ar(pack + L.stats, 6);
        }
        ctl_invoke(CTL_STEP, speculative ? 1 : 0);
    }
    void ctl_accept(bool blocks) override {
        view[0] = view[1];
        if (blocks) { blk_acc = alt_acc; blk_w = alt_w; }
    }
    const double* ctl_wait() override { return ctl.rec.data(); }
    void ctl_fetch(double* i, double* c, double* t, double* delta) override {
        std::memcpy(i, ctl.x[0].data(), sizeof(double) * intr[0].size());
        if (s.chain != CBA_CHAIN_INTRINSIC) std::memcpy(c, ctl.x[0].data() + ctl.V.pk_cam, sizeof(double) * cam[0].size());
        if (s.chain == CBA_CHAIN_BUNDLE) std::memcpy(t, ctl.x[0].data() + ctl.V.pk_target, sizeof(double) * 7);
        std::memcpy(delta, ctl.x[1].data() + ctl.V.pk_delta, sizeof(double) * s.nsh);
    }
    void ctl_line_search_done(const double* scal) override {
        // the last sample's shared blocks (upload_shared(1, ...)) are the trial point the decision is about
        std::memcpy(ctl.x[1].data(), intr[1].data(), sizeof(double) * intr[1].size());
        std::memcpy(ctl.x[1].data() + ctl.V.pk_cam, cam[1].data(), sizeof(double) * cam[1].size());
        std::memcpy(ctl.x[1].data() + ctl.V.pk_target, target[1].data(), sizeof(double) * 7);
        std::memcpy(ctl.scal.data(), scal, sizeof(double) * CS_COUNT);
        ctl_invoke(CTL_LS_DONE, 0);
    }
};

struct Session {
    Structure s;
    std::vector<double> intr, cam, view, target;
};

thread_local std::string g_err;

template <typename F>
int guarded(F&& f) {
    try { f(); return CBA_OK; }
    catch (const std::invalid_argument& e) { g_err = e.what(); return CBA_ERR_INVALID_ARGUMENT; }
    catch (const std::runtime_error& e) { g_err = e.what(); return CBA_ERR_RUNTIME; }
    catch (const std::exception& e) { g_err = e.what(); return CBA_ERR_INTERNAL; }
}

void load(const cba_reproj_problem& d, Session& ss) {
    build_structure(d, ss.s);
    ss.intr.assign(d.intr, d.intr + static_cast<size_t>(d.n_cams) * ss.s.PI);
    if (d.chain != CBA_CHAIN_INTRINSIC) ss.cam.assign(d.cam_pose, d.cam_pose + 7 * static_cast<size_t>(d.n_cams));
    else ss.cam.assign(7 * static_cast<size_t>(d.n_cams), 0.0);
    if (d.chain != CBA_CHAIN_BUNDLE && d.n_views > 0) ss.view.assign(d.view_pose, d.view_pose + 7 * static_cast<size_t>(d.n_views));
    if (d.chain == CBA_CHAIN_BUNDLE) ss.target.assign(d.target_pose, d.target_pose + 7);
    else ss.target.assign(7, 0.0);
}
void store(const cba_reproj_problem& d, const Session& ss) {
    std::memcpy(d.intr, ss.intr.data(), sizeof(double) * ss.intr.size());
    if (d.chain != CBA_CHAIN_INTRINSIC) std::memcpy(d.cam_pose, ss.cam.data(), sizeof(double) * ss.cam.size());
    if (d.chain != CBA_CHAIN_BUNDLE && d.n_views > 0) std::memcpy(d.view_pose, ss.view.data(), sizeof(double) * ss.view.size());
    if (d.chain == CBA_CHAIN_BUNDLE) std::memcpy(d.target_pose, ss.target.data(), sizeof(double) * 7);
}

}  // namespace

extern "C" {

const char* hm_last_error(void) { return g_err.c_str(); }

// LM solve with the product's host driver + the CPU test backend.  allreduce may be NULL.
// speculate: 1 / 0 = linearise trial points ahead of the accept decision or not, -1 = the driver's default (CBA_LM_SPECULATE)
// stats8 (may be NULL) = {all-reduce calls, all-reduced doubles, speculative steps, hits, misses, rejected steps, line searches,
// line-search evaluations}
// controller: 1 = the controller form of the iteration (lm_ctl.hpp, what libcalibba runs), 0 = the host-side form, -1 = default
int hm_reproj_solve_mode(const cba_reproj_problem* d, const cba_options* o, cba_allreduce_fn fn, void* user, int n_ranks, int rank,
                         int speculate, int controller, cba_summary* out, int64_t* stats8) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        if (controller >= 0) be.use_ctl = controller != 0;
        AllReduce ar = [&](double* buf, int64_t n) {
            if (fn && fn(buf, n, user) != 0) throw std::runtime_error("allreduce callback failed");
        };
        LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, ar, n_ranks, rank);
        if (speculate >= 0) drv.set_speculate(speculate != 0);
        drv.solve(*o, out);
        store(*d, ss);
        if (stats8) {
            const ExchangeStats& x = drv.exchange_stats();
            stats8[0] = x.allreduce_calls; stats8[1] = x.allreduce_doubles; stats8[2] = x.speculative_steps;
            stats8[3] = x.speculation_hits; stats8[4] = x.speculation_misses; stats8[5] = x.rejected_steps;
            stats8[6] = x.line_searches; stats8[7] = x.line_search_evaluations;
        }
    });
}
int hm_reproj_solve_ex(const cba_reproj_problem* d, const cba_options* o, cba_allreduce_fn fn, void* user, int n_ranks, int rank,
                       int speculate, cba_summary* out, int64_t* stats8) {
    return hm_reproj_solve_mode(d, o, fn, user, n_ranks, rank, speculate, -1, out, stats8);
}
int hm_reproj_solve(const cba_reproj_problem* d, const cba_options* o, cba_allreduce_fn fn, void* user, int n_ranks, int rank,
                    cba_summary* out) {
    return hm_reproj_solve_ex(d, o, fn, user, n_ranks, rank, -1, out, nullptr);
}

// The controller's reduced solve alone (lm_ctl.hpp ctl_cholesky + ctl_backsolve, one-thread team): x = A^-1 b for the symmetric
// n x n matrix A (row-major; the lower triangle is read).  Returns 1 when a pivot is not positive.
int hm_ctl_dense_solve(int n, const double* A, const double* b, double* x) {
    int bad = 0;
    const int rc = guarded([&] {
        const int M = ctl_padded(n), lda = ctl_lda(n);
        std::vector<double> W(static_cast<size_t>(M + 1) * lda, 0.0), Ld(static_cast<size_t>(M) * CTL_NB, 0.0), rd(M, 0.0), xs(M, 0.0);
        for (int r = 0; r < M; ++r)
            for (int c = 0; c <= r; ++c) W[static_cast<size_t>(r) * lda + c] = r < n ? A[static_cast<size_t>(r) * n + c] : (r == c ? 1.0 : 0.0);
        for (int c = 0; c < n; ++c) W[static_cast<size_t>(M) * lda + c] = b[c];
        SerialTeam tm;
        int ok = 1;
        if (!ctl_cholesky(tm, W.data(), lda, M, Ld.data(), rd.data(), &ok)) { bad = 1; return; }
        ctl_backsolve(tm, W.data(), lda, M, Ld.data(), rd.data(), xs.data());
        for (int c = 0; c < n; ++c) x[c] = xs[c];
    });
    return rc != 0 ? rc : bad;
}

// per-block packed [H | g | s] rows at the problem's parameters, through the direct (moments = 0) or the moment form
int hm_reproj_block_normal_eq(const cba_reproj_problem* d, int moments, double* out) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        be.use_moments = moments != 0;
        be.upload_shared(0, ss.intr.data(), ss.cam.data(), ss.target.data());
        std::vector<double> cam_acc;
        double c2[2];
        be.normal_eq(1.0, cam_acc, c2);
        std::memcpy(out, be.blk_acc.data(), sizeof(double) * be.blk_acc.size());
    });
}

// build_structure's validation alone (have_records: the observations arrive as per-block records, cba_reproj_create_aos)
long long hm_choose_mode_b_tile(long long n_blocks, long long n_obs, int two_wavefront_form) {
    return choose_mode_b_tile(n_blocks, n_obs, two_wavefront_form != 0, 2048);
}

int hm_structure_check(const cba_reproj_problem* d, int have_records) {
    return guarded([&] {
        Structure s;
        build_structure(*d, s, have_records != 0);
    });
}

// the constant / gauge masks LMDriver hands to a solver that runs the iteration itself (resident_lm.hip): active[nsh], cam_var[n_cams],
// flags[3] = {intr_var, target_var, constrained}
int hm_reproj_masks(const cba_reproj_problem* d, const cba_options* o, int8_t* active, int8_t* cam_var, int32_t* flags) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
        const LMDriver::Masks m = drv.masks(*o);
        for (int i = 0; i < ss.s.nsh; ++i) active[i] = m.active[i];
        for (int c = 0; c < ss.s.n_cams; ++c) cam_var[c] = m.cam_var[c];
        flags[0] = m.intr_var; flags[1] = m.target_var; flags[2] = m.constrained;
    });
}

int64_t hm_reproj_covariance_dim(const cba_reproj_problem* d) {
    Session ss;
    try { load(*d, ss); } catch (...) { return -1; }
    CpuBackend be(ss.s, *d, ss.view);
    LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
    return drv.covariance_dim();
}

int hm_reproj_covariance(const cba_reproj_problem* d, const cba_options* o, double* cov) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
        drv.covariance(*o, cov);
    });
}

int64_t hm_reproj_covariance_shared_dim(const cba_reproj_problem* d) {
    Session ss;
    try { load(*d, ss); } catch (...) { return -1; }
    CpuBackend be(ss.s, *d, ss.view);
    LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
    return drv.shared_covariance_dim();
}

int hm_reproj_covariance_shared(const cba_reproj_problem* d, const cba_options* o, double* cov) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
        drv.covariance(*o, cov, true);
    });
}


// the marginal 7 x 7 pose covariance of the listed views (LMDriver::covariance, the path of cba_reproj_covariance_views)
int hm_reproj_covariance_views(const cba_reproj_problem* d, const cba_options* o, int n_sel, const int32_t* views, double* cov7x7) {
    return guarded([&] {
        Session ss;
        load(*d, ss);
        CpuBackend be(ss.s, *d, ss.view);
        LMDriver drv(ss.s, be, ss.intr, ss.cam, ss.view, ss.target, [](double*, int64_t) {}, 1, 0);
        drv.covariance(*o, nullptr, true, views, n_sel, cov7x7);
    });
}

}  // extern "C"
