// lm_state.hpp — device-side LM / Schur state behind an Engine (backend_hip.hip builds it; resident_lm.hip reads it).
#pragma once
#include "engine.hpp"
#include "lm_ctl.hpp"
#include "schur_math.hpp"
#include "structure.hpp"

namespace cba {

struct HipLMState {
    Structure s;
    SchurDims dims;
    int n_vchunks = 0, n_tiles = 0, n_pairs = 0, n_cchunks = 0;
    int fuse_small = 1; // the small stages of a linear solve share launches (backend_hip.hip k_sys_stage2 / 3 / k_sys_pack / k_step_head);
                        // 0: one launch per stage (experiment builds: CBA_LM_FUSE)
    int schur_wave = 1; // per-view elimination / back-substitution with one wavefront per view (0: one thread per view, CBA_SCHUR_WAVE)
    int syrk_mfma = 1;  // Schur contraction on the matrix cores when nsh >= 64 (CBA_SYRK_MFMA=0: register-blocked VALU form)
    DevBuf<int32_t> view_cam_blk, cam_blk;
    DevBuf<int64_t> cchunk_off, cam_seg, link_off;
    DevBuf<int32_t> link_blk;
    DevBuf<double> cam_partial, view_gmax, view_delta, view_stats, syrk_partial;  // syrk_partial[chunk] = [tiles | g_schur]
    // Results of a stage land DIRECTLY in page-locked host memory (device-visible): the last kernels of the stage write there,
    // so there is no copy command between the kernels and the one stream synchronisation.
    //   pin    [syrk tiles (n_pairs*4096) | g_schur (nsh) | gmax, #failed views]      pin_ne [camera sums | cost, sum s]
    //   pin_tr [8..12): step2, xnorm2, g^T d, d^T H d;  [24..26): trial cost, sum s
    PinnedBuf<double> pin, pin_ne, pin_tr;
    // the packed exchange buffer of a linear solve (lm_core.hpp PackLayout): assembled and all-reduced on the device
    DevBuf<double> pack_dev, sys_tiles, stat_dev;
    PinnedBuf<double> pin_packed;
    // the ONE wait of an LM step: hipStreamSynchronize sleeps on an interrupt (20-50 us to wake up on ROCm 7.2, more than the
    // whole host side of a step); polling an event recorded behind the stage returns within a few us (CBA_SYNC_SPIN=0: sleep)
    hipEvent_t step_done = nullptr;
    hipEvent_t ctl_done = nullptr;  // experiment builds (CBA_LM_CTL_EVENT=1): recorded behind every controller launch for ctl_wait to sleep on
    int ctl_event = 0;
    int sync_spin = 1;
    ~HipLMState() {
        if (step_done) (void)hipEventDestroy(step_done);
        if (ctl_done) (void)hipEventDestroy(ctl_done);
    }
    int64_t xs[8] = {0, 0, 0, 0, 0, 0, 0, 0};  // ExchangeStats of the last solve (cba_reproj_solve_stats)
    // The three launch sequences of an LM iteration are HIP graphs (captured from the stream on first use): a stage is one
    // hipGraphLaunch instead of 7-15 kernel launches and copies — the iteration is host-launch bound for small and mid-size
    // problems (C1: ~25 API calls of 5-10 us per iteration against ~100 us of kernels).  What changes between launches
    // travels through memory the graph's own copy nodes read at execution time: [radius, init_scale] and the shared step
    // in pinned host buffers.  Values baked into kernel arguments (huber, constrained, fp32 mode) key the graph.
    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        double huber = 0.0;
        int constrained = -1, scalar = -1;
        int uses = 0;  // plain launches of this stage with the current key so far
        ~GraphSlot() { if (exec) (void)hipGraphExecDestroy(exec); }
    };
    GraphSlot g_new, g_schur, g_trial;
    bool graphs_ok = true;
    int graph_after = 200;// capture + instantiate cost ~1 ms per stage on ROCm 7.2: only pays for itself on solves longer
                          // than a solve, so a stage runs as plain launches until it has been used this many times with
                          // the same key — i.e. on handles that are solved again and again (CBA_LM_GRAPH=<n>, 1 = at once, 0 = never)
    PinnedBuf<double> pin_lmp;      // [radius, init_scale]: k_schur_view reads it in place (wave-uniform, two numbers)
    PinnedBuf<double> pin_pack[2];  // staging of Engine::shared_pack[k]: [intr | cam | target | shared step]
    bool current_is_on_device = false;  // set by accept(): copy 0 on the device already equals the driver's next upload
    // resident LM (resident_lm.hip): per-camera block lists, the masks of the current options, reduced-system scratch
    int resident_mode = 1;             // 0 never, 1 when the problem is small (default), 2 whenever the kernel can run it
    int64_t resident_max_obs = -1;     // >= 0: the automatic mode takes problems up to this many observations
                                       // (CBA_LM_RESIDENT_MAX_OBS); -1: the measured crossover rule of resident_lm_eligible
    DevBuf<int64_t> cam_off;
    DevBuf<int8_t> res_active, res_cam_var;
    DevBuf<double> res_Hcc, res_Ssch, res_out;
    PinnedBuf<double> pin_res;
    PinnedBuf<int8_t> pin_mask;
    // LM controller (lm_ctl.hpp / lm_ctl.hip): the reduced system, the control scalars and the radius of the next elimination
    // live in device memory; the controller publishes its control record into page-locked host memory
    DevBuf<double> ctl_buf;     // [scal CS_COUNT | lmp 8 | Hcc n*n | gc n | scale2 n | xs n | rdiag n | x_tmp pk_size | A (n+1)*lda (n > CTL_LDS_MAX_N)]
    DevBuf<int32_t> ctl_idx;
    DevBuf<int8_t> ctl_eff;
    PinnedBuf<double> ctl_rec;  // [control record CS_COUNT | staging of the scalars CS_COUNT | staging of the start point pk_size]
    CtlView ctl_view{};
    int ctl_n = -1;             // reduced size the buffers above were laid out for
    int rccl_timeout_s = 120;   // ctl_wait gives a step's collective this long before it aborts the communicator (CBA_RCCL_TIMEOUT_S)
    int ctl_prelaunch = 1;      // queue the head of the next step behind the controller before its decision is known (CBA_LM_PRELAUNCH)
    int ctl_poll_us = 2000;     // how long ctl_wait spins on the record before it starts napping between looks
    int lm_ctl_mode = 1;        // 1 = the controller form of the host-driven iteration (default), 0 = the host-side form (CBA_LM_CTL)
};


inline HipLMState* lm_state(Engine& e) { return reinterpret_cast<HipLMState*>(e.lm_state); }

// resident_lm.hip: the whole LM in one single-workgroup kernel, for problems too small to fill the chip
bool resident_lm_eligible(const Engine& e, const cba_options& o);
void resident_lm_warm(Engine& e);
// false: a step of a bounds-constrained problem needs Ceres' line search (line_search.hpp) — nothing was changed, the caller runs
// the host-driven iteration instead
bool resident_lm_solve(Engine& e, const cba_options& o, cba_summary* out, bool keep_parameters = false);

// lm_ctl.hip: one invocation of the controller as a single workgroup on `stream`
void launch_lm_ctl(const CtlView& V, int mode, int flag, hipStream_t stream);
bool lm_ctl_fits_lds(int n);
void warm_lm_ctl();

}  // namespace cba
