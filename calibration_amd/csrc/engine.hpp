// engine.hpp — internal state behind the opaque cba_reproj handle.
//
// HBM layout (all fp64, SoA, one allocation per array):
//   u, v              [ld]          pixel observations; every residual block starts at an EVEN padded index
//                                    (so a lane's two observations are one 16-byte load/store) and ld is
//                                    the padded total rounded up to 256 elements
//   X, Y              [ld_xy]       target-plane points, DEDUPLICATED: residual blocks whose object_xy lists are
//                                    bitwise identical (the usual case: every view sees the same physical
//                                    target) share one copy, so the per-observation HBM read drops from 32 B
//                                    towards 16 B and the shared copy stays L2 / Infinity-Cache resident
//   J (Mode A output) [n_tilesA][2 + 2P][128]  tile-blocked: for every 128-observation tile one contiguous
//                                    (2+2P) KiB region = u/v residual rows, then the P Jacobian columns of
//                                    the u row, then of the v row (streams like a fill; +6 % over whole-array
//                                    columns r[2][ld], J[2P][ld], which CBA_EVAL_BLOCKED=0 still selects)
//   bc                [n_blocks][36] per-block chain constants (reproj_math.hpp BC_*)
//   sd                [n_cams][36]  Scheimpflug per-camera constants (SD_*)
//   intr/cam/view/target            parameter blocks, current [0] and trial [1] copies
//   tilesA / tilesB                 wave-tile tables: a tile is <=128 (Mode A) / >= TILE_B except a block's last (Mode B, R)
//                                    consecutive observations of ONE block, processed by ONE wavefront
//   partial           [n_tilesB][NACC]  per-tile Mode B sums;  blk_acc [n_blocks][NACC] per-block sums
#pragma once
#include <hip/hip_runtime.h>
#include "exp_env.hpp"

#include <cstdint>
#include <deque>
#include <cstdlib>
#include <exception>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/calibba.h"

namespace cba {

struct HipError : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct NoDevice : std::runtime_error {
    using std::runtime_error::runtime_error;
};

// Environment switches.  The shipped library reads a short, documented list of run-time switches with std::getenv (DESIGN.md
// section 9: every one of them selects between forms that give the same results).  Everything else that was ever tuned or ablated
// through the environment - part counts, layouts, timing-only ablations whose results are WRONG - is an experiment knob: read
// through cba_exp_env(), which answers only in a library built with -DCBA_EXPERIMENTS (make EXPERIMENTS=1; tools/exp.py uses such
// a build).  In the shipped library a stray variable in a user's environment cannot change what a calibration computes.

#define CBA_HIP(expr)                                                                                      \
    do {                                                                                                   \
        hipError_t e_ = (expr);                                                                            \
        if (e_ != hipSuccess)                                                                              \
            throw cba::HipError(std::string(#expr) + ": " + hipGetErrorString(e_) + " (" __FILE__ ":" +    \
                                std::to_string(__LINE__) + ")");                                           \
    } while (0)

// Process-wide cache of device / page-locked blocks and of streams (block_cache.cpp).  The reference's pipeline calls the
// one-shot entry points stage after stage; a handle is ~70 hipMalloc + ~10 hipHostMalloc, and giving them back cost 1.9 ms
// of a 5 ms C1-sized call (hipFree synchronises the device).  Released blocks up to 16 MiB are kept (at most 256 MiB per
// kind and device) in power-of-two size classes and handed to the next handle; cba_trim_cache() frees them.
void* cache_alloc(bool pinned, size_t bytes, size_t* granted);  // current device; throws HipError
void cache_release(bool pinned, int device, void* p, size_t granted) noexcept;
hipStream_t cache_stream();                                     // an idle non-blocking stream of the current device
void cache_stream_release(int device, hipStream_t s) noexcept;  // the caller has synchronised it
void cache_trim();

template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t granted = 0;  // bytes of the underlying block
    int device = 0;
    bool owned = true;   // false: a view into another buffer (view())
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { release(); }
    // A cached block may be handed to another handle at once: nothing may still be running on it.  Normal paths have
    // synchronised their stream before buffers go out of scope; unwinding from an exception and replacing a live buffer
    // have not, so those wait for the device.
    void release() {
        if (p && owned) {
            if (std::uncaught_exceptions() > 0) (void)hipDeviceSynchronize();
            cache_release(false, device, p, granted);
        }
        p = nullptr; n = 0; granted = 0; owned = true;
    }
    // non-owning window of `count` elements at `ptr` (inside a buffer that outlives this one)
    void view(T* ptr, size_t count) {
        release();
        p = ptr; n = count; owned = false;
    }
    void alloc(size_t count) {
        if (p && owned) (void)hipDeviceSynchronize();
        release();
        if (count == 0) count = 1;
        CBA_HIP(hipGetDevice(&device));
        p = static_cast<T*>(cache_alloc(false, count * sizeof(T), &granted));
        n = count;
    }
    void upload(const T* src, size_t count, hipStream_t s, size_t first = 0) {
        if (count) CBA_HIP(hipMemcpyAsync(p + first, src, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T* dst, size_t count, hipStream_t s, size_t first = 0) const {
        if (count) CBA_HIP(hipMemcpyAsync(dst, p + first, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s) { CBA_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// Page-locked host staging for the small device-to-host results of an LM step: a copy into pageable memory blocks the
// host once per call, a copy into pinned memory is queued on the stream and only the single hipStreamSynchronize waits.
template <typename T>
struct PinnedBuf {
    T* p = nullptr;
    size_t n = 0;
    size_t granted = 0;
    int device = 0;
    PinnedBuf() = default;
    PinnedBuf(const PinnedBuf&) = delete;
    PinnedBuf& operator=(const PinnedBuf&) = delete;
    ~PinnedBuf() { if (p) cache_release(true, device, p, granted); }
    void reserve(size_t count) {
        if (count <= n) return;
        if (p) cache_release(true, device, p, granted);
        p = nullptr; n = 0; granted = 0;
        CBA_HIP(hipGetDevice(&device));
        p = static_cast<T*>(cache_alloc(true, count * sizeof(T), &granted));
        n = count;
    }
};

// An idle non-blocking stream of the current device, leased from the process-wide pool; synchronised and returned on scope
// exit.  Declare it BEFORE the buffers that are used on it (members are released in reverse order).
struct StreamLease {
    hipStream_t s = nullptr;
    int device = 0;
    StreamLease() {
        CBA_HIP(hipGetDevice(&device));
        s = cache_stream();
    }
    StreamLease(const StreamLease&) = delete;
    StreamLease& operator=(const StreamLease&) = delete;
    ~StreamLease() {
        if (s) {
            (void)hipStreamSynchronize(s);
            cache_stream_release(device, s);
        }
    }
    operator hipStream_t() const { return s; }
};

struct Tile {          // 32 bytes, read with scalar loads (wave-uniform)
    int32_t blk;       // residual block
    int32_t count;     // observations in this tile (Mode A: padded, even; Mode B/R: valid count)
    int64_t start;     // padded observation index of the tile's first observation (u, v, outputs)
    int64_t xy_start;  // index of the tile's first target point in the DEDUPLICATED X, Y arrays
    int64_t reserved;
};

constexpr int TILE_A = 128;  // Mode A: 64 lanes x 2 adjacent observations
constexpr int OPL_B = 32;    // Mode B/R: observations per lane of the SHORTEST full tile (2048 observations: the wave reduction and the
                             // tile's partial row are paid once per tile; large problems use longer tiles, capi.cpp)
constexpr int TILE_B = 64 * OPL_B;

struct Engine {
    // ---- problem -----------------------------------------------------------------------------
    int chain = 0, model = 0;
    int n_blocks = 0, n_cams = 0, n_views = 0;
    int64_t first_view_global = 0;
    int64_t n_obs = 0, ld = 0, ld_xy = 0;
    int64_t n_xy_unique_blocks = 0;
    int PI = 10, PL = 16, NACC = 0;
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;

    std::vector<int64_t> blk_offset;  // unpadded CSR (host)
    std::vector<int64_t> pad_offset;  // padded start of every block (host)
    std::vector<int64_t> xy_offset;   // start of every block's target points in the deduplicated X, Y arrays (host)
    std::vector<int32_t> blk_cam, blk_view;
    std::vector<int64_t> blk_tile_off;  // Mode B tiles per block CSR (host)
    int64_t n_tilesA = 0, n_tilesB = 0;
    int32_t max_tileB = 0;  // observations of the largest Mode B tile (small-block problems take the single-group kernel)

    // host copies of the parameters (current accepted state); h_cam / h_target are always sized
    // (7 per camera / 7) and simply unused by chains that have no such block
    std::vector<double> h_intr, h_cam, h_view, h_target;

    // ---- device ------------------------------------------------------------------------------
    DevBuf<double> X, Y, u, v, r, J;
    // A tile-blocked Mode A output above 4 GiB is held as SEGMENTS of whole tiles, each a physically contiguous block of at most
    // 4 GiB (capi.cpp alloc_output: such blocks stream at the rate the memory system was laid out for and cost a millisecond to
    // obtain; one contiguous block of 59 GB costs 1.7 s, a plain one runs 5-8 % slower).  k_eval is launched once per segment with
    // the segment's tile range and base address: the kernel is unchanged.  Empty when the output is one block (J).
    std::deque<DevBuf<double>> Jseg;
    int64_t seg_tiles = 0;  // tiles per segment (the last one may hold fewer)
    double* eval_tile_ptr(int64_t w, int64_t tile_doubles) {  // device address of tile w's output
        return Jseg.empty() ? J.p + w * tile_doubles : Jseg[static_cast<size_t>(w / seg_tiles)].p + (w % seg_tiles) * tile_doubles;
    }
    // fp32 study (BASELINE config 5): rounded copies of the observations / constant tables, fp32 Mode A output
    int scalar = 0;  // 0 = fp64 per-observation arithmetic, 1 = fp32 (accumulators stay fp64)
    DevBuf<float> Xf, Yf, uf, vf, Jf, bcf, sdf, intrf;
    DevBuf<double> bc, sd, aux;  // aux: bundle b_T_g [n_blocks][12]
    // The SHARED parameter blocks of copy k live side by side in shared_pack[k] = [intr | cam poses | target pose | shared step]
    // (intr[k], cam[k], target[k], delta_sh are windows into it): a trial point goes up as ONE copy and is accepted by one
    // small kernel (backend_hip.hip k_accept).  Declared before its windows.
    DevBuf<double> shared_pack[2];
    size_t pk_cam = 0, pk_target = 0, pk_delta = 0, pk_size = 0;  // offsets (doubles) of the windows
    DevBuf<double> intr[2], cam[2], view[2], target[2];
    int eval_blocked = 1;  // Mode A output layout: 1 tile-blocked out[tile][2+2P][128] (default), 0 whole-array columns
    int eval_done = 0, eval_blocked_last = 0;
    int eval_ablate = 0;  // timing-only ablation of k_eval (CBA_EVAL_ABLATE; outputs are wrong when non-zero)
    int eval_variant = 1;  // k_eval variant: bit 0 = non-temporal stores, bits 1.. = log2(tiles per wave)
    int active = 0;  // parameter copy (0 current / 1 trial) the constants bc, sd were last built from
    const double* gate = nullptr;  // device flag the launches of block constants / Mode B check (0: do nothing); set by the LM driver
                                   // while it queues the head of a step ahead of the controller's decision, nullptr otherwise
    // >= 0: launch_normal_eq() also leaves the blocks' robust weights / |r|^2 (blk_w, blk_s) for this Huber parameter where one
    // of its kernels can take the work along; head_weights says whether the last call did (the caller skips k_weights then)
    double head_huber = -1.0;
    bool head_weights = false;
    DevBuf<int32_t> d_blk_cam, d_blk_view;
    DevBuf<Tile> tilesA, tilesB;
    DevBuf<int64_t> d_blk_tile_off;
    DevBuf<double> partial, blk_acc, blk_s, scalar_out;
    DevBuf<double> cost_part; // [2 * ceil(n_blocks / 2048)] partial cost pairs (only above 4096 blocks)
    DevBuf<double> blk_mom;   // [n_blocks][MomLayout::N] Mode B moment rows of the two-pose chains (kernels_reproj.hip)
    int modeb_moments = 1;    // 0 = accumulate the 12 pose columns directly (CBA_MODEB_MOMENTS=0, for A/B comparison)
    int modeb_shared = 1;     // 1 = one workgroup per tile, the parts as its wavefronts, rows evaluated once and shared through LDS
                              // (kernels_modeb.hip); 0 = one launch per part, every part re-evaluates the rows (CBA_MODEB_SHARED)
    int modeb_split = -1;     // one-pose chain: 1 = pose rows | intrinsics block (mode_b.hpp SplitPoseIntr), 0 = round-robin halves,
                              // -1 = per camera model as measured (CBA_MODEB_SPLIT)

    // ---- LM / Schur state (backend_hip.hip, resident_lm.hip; the rest lives in HipLMState, lm_state.hpp) ----------
    DevBuf<double> blk_w;       // [n_blocks] Huber weights rho'(s_b)
    DevBuf<double> blk_acc_alt, blk_w_alt;  // the trial point's block sums / weights of a speculative LM step (backend_hip.hip sys_step)
    DevBuf<double> cam_acc;     // [n_cams][NACC] weighted per-camera sums
    DevBuf<double> view_L;      // [n_views][36] Cholesky factor of damped H_pp (lower, row-major)
    DevBuf<double> view_y;      // [n_views][6]
    DevBuf<double> view_D;      // [n_views][6]  damping added to diag(H_pp)
    DevBuf<double> view_gp;     // [n_views][6]  g_p
    DevBuf<double> view_scale2; // [n_views][6]  jacobi scale^2
    DevBuf<double> blk_Z;       // [n_blocks][6][PSH]
    DevBuf<int32_t> view_fixed; // [n_views]
    DevBuf<double> delta_sh;    // [nsh] shared step of the current trial

    // ---- collectives -------------------------------------------------------------------------
    cba_allreduce_fn allreduce = nullptr;
    void* allreduce_user = nullptr;
    void* rccl_comm = nullptr;
    int n_ranks = 1, rank = 0;
    DevBuf<double> coll_buf;
    PinnedBuf<double> coll_pin;  // page-locked staging of the packed all-reduce buffer (both copies are queued, one sync)
    void* lm_state = nullptr;  // HipLMState (backend_hip.hip)

    ~Engine();
};

// kernels_reproj.hip
void ensure_f32_buffers(Engine& e);                      // fp32 copies of the observations + float tables
void launch_block_consts(Engine& e, int which);         // params[which] -> bc, sd
void launch_camera_consts(Engine& e, int which);        // ... the per-camera part alone (sd)
void launch_eval(Engine& e);                            // Mode A: r, J at bc/sd
void launch_resid(Engine& e);                           // Mode R: blk_s[b] = |r_b|^2
void warm_reproj_kernels();                              // forces the code object of kernels_reproj.hip to load
void launch_normal_eq(Engine& e);                       // Mode B: blk_acc[b] = [H | g | s]
bool launch_normal_eq_shared_rows(Engine& e, double* rows);  // kernels_modeb.hip: the parts of a tile as one workgroup, rows through LDS
void launch_cost(Engine& e, double huber_delta, double* out = nullptr);  // out (default scalar_out) = {1/2 sum rho(blk_s), sum blk_s}

// backend_hip.hip
void init_lm_state(Engine& e, const cba_reproj_problem& d, bool have_records = false);
void destroy_lm_state(Engine& e);
void warm_lm(Engine& e);
void solve_lm(Engine& e, const cba_options& o, cba_summary* out);
void solve_stats(const Engine& e, int64_t stats8[8]);  // ExchangeStats of the last host-driven solve
void set_lm_mode(Engine& e, int mode);  // 0 host-driven iteration, 1 automatic (default), 2 resident kernel whenever it can run the problem
void compute_covariance(Engine& e, const cba_options& o, double* cov, bool shared_only = false);
void compute_covariance_views(Engine& e, const cba_options& o, const int32_t* views, int n_sel, double* view_cov);
int64_t covariance_dim(const Engine& e);
int64_t shared_covariance_dim(const Engine& e);
void engine_allreduce(Engine& e, double* host_buf, int64_t count);
void rccl_unique_id(uint8_t* id);
void rccl_init(Engine& e, const uint8_t* id, int n_ranks, int rank);
void rccl_destroy(Engine& e);
void rccl_abort(Engine& e);  // ncclCommAbort: this rank leaves a multi-rank solve abnormally; the peers' collectives fail instead of hanging
void planar_pose_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                       const double* kmtx5, int num_radial, double* pose7, const cba_options* o, cba_summary* summaries,
                       double* distortion, double* rms, double* cov, int device);
void homography_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                      double* h9, const cba_options* o, cba_summary* summaries, double* cov64, int device);
void semidlt_solve(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                   double* kappa5, double* poses7, int num_radial, const double* bounds_lo, const double* bounds_hi,
                   const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o, cba_summary* summary,
                   double* distortion, double* view_errors, double* cov, int device);
void semidlt_solve_sharded(int n_local, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                           int n_views_total, int first_view, double* kappa5, double* poses7, int num_radial, const double* bounds_lo,
                           const double* bounds_hi, const int32_t* fixed_idx, const double* fixed_val, int n_fixed, const cba_options* o,
                           cba_summary* summary, double* distortion, double* view_errors, double* cov, int device, cba_allreduce_fn fn,
                           void* user, void* rccl_comm);
void dlt_homography_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                          double* H9, int32_t* ok, int device);
void planar_seed_batch(int n_views, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                       const double* kmtx5, double* pose7, int device);
// fn / user / n_ranks / rank: multi-GPU form — this rank's share of the pairs, sums all-reduced through the host callback
// (rccl_comm: an ncclComm_t over the ranks' devices - the sums are all-reduced on the device instead of through fn)
void handeye_dlt(int n_poses, const double* bTg, const double* cTt, double min_angle_deg, double* pose7, int device,
                 cba_allreduce_fn fn = nullptr, void* user = nullptr, int n_ranks = 1, int rank = 0, void* rccl_comm = nullptr);
void handeye_solve(int n_poses, const double* bTg, const double* cTt, double* pose7, const cba_options* o, cba_summary* s,
                   double* cov, int device, cba_allreduce_fn fn = nullptr, void* user = nullptr, int n_ranks = 1, int rank = 0,
                   void* rccl_comm = nullptr);
void* rccl_comm_create(const uint8_t* id, int n_ranks, int rank);  // backend_hip.hip: ncclCommInitRank on the current device
void rccl_comm_destroy(void* comm, bool abort);

}  // namespace cba
