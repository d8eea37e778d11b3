// capi.cpp — the extern "C" boundary of libcalibba.so (include/calibba.h).
//
// Host-side duties only: validate like the reference, lay observations out in padded SoA, build the
// wave-tile tables, move buffers, launch kernels, translate C++ exceptions into status codes.
// There is no CPU arithmetic path: without a HIP device every compute call returns
// CBA_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <exception>
#include <mutex>
#include <thread>
#include <cstring>
#include <memory>
#include <string>
#include <unordered_map>

#include "engine.hpp"
#include "structure.hpp"

using namespace cba;

static thread_local std::string g_err;

template <typename F>
static cba_status guarded(F&& f) {
    try {
        f();
        return CBA_OK;
    } catch (const std::invalid_argument& e) {
        g_err = e.what();
        return CBA_ERR_INVALID_ARGUMENT;
    } catch (const NoDevice& e) {
        g_err = e.what();
        return CBA_ERR_NO_DEVICE;
    } catch (const HipError& e) {
        g_err = e.what();
        return CBA_ERR_HIP;
    } catch (const std::runtime_error& e) {
        g_err = e.what();
        return CBA_ERR_RUNTIME;
    } catch (const std::exception& e) {
        g_err = e.what();
        return CBA_ERR_INTERNAL;
    }
}

// the device of the entry points that take no handle and no device argument (one process per GPU: cba_set_device(LOCAL_RANK))
static std::atomic<int> g_default_device{0};
static int default_device() { return g_default_device.load(); }

static int device_count() {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

Engine::~Engine() {
    // the buffers below go back to the process-wide cache (engine.hpp): nothing may still be running on them
    if (stream) (void)hipStreamSynchronize(stream);
    rccl_destroy(*this);
    destroy_lm_state(*this);
    if (ev0) (void)hipEventDestroy(ev0);
    if (ev1) (void)hipEventDestroy(ev1);
    if (stream) cache_stream_release(device, stream);
}

static Engine* as_engine(cba_reproj* h) {
    if (!h) throw std::invalid_argument("null handle");
    return reinterpret_cast<Engine*>(h);
}

static void upload_params(Engine& e) {
    e.intr[0].upload(e.h_intr.data(), e.h_intr.size(), e.stream);
    e.cam[0].upload(e.h_cam.data(), e.h_cam.size(), e.stream);
    if (!e.h_view.empty()) e.view[0].upload(e.h_view.data(), e.h_view.size(), e.stream);
    e.target[0].upload(e.h_target.data(), e.h_target.size(), e.stream);
}

namespace {
struct PhaseTimer {  // CBA_CREATE_TIMING=1: print where cba_reproj_create spends its time (host staging vs PCIe vs set-up)
    bool on = getenv("CBA_CREATE_TIMING") != nullptr;
    std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
    void lap(const char* what) {
        if (!on) return;
        const auto n = std::chrono::steady_clock::now();
        std::fprintf(stderr, "[cba create] %-28s %8.2f ms\n", what, std::chrono::duration<double, std::milli>(n - t).count());
        t = n;
    }
};
}  // namespace

// run fn(b) for b in [0, n) on up to 8 host threads (contiguous ranges); small n stays on the calling thread
template <class F>
static void parallel_blocks(int n, F&& fn) {
    const unsigned hw = std::thread::hardware_concurrency();
    const int nt = std::max(1, std::min<int>({8, static_cast<int>(hw ? hw : 1), n / 64}));
    if (nt == 1) { for (int b = 0; b < n; ++b) fn(b); return; }
    std::vector<std::thread> th;
    std::exception_ptr err;
    std::mutex mu;
    for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            try { for (int b = static_cast<int>(static_cast<int64_t>(n) * t / nt); b < static_cast<int>(static_cast<int64_t>(n) * (t + 1) / nt); ++b) fn(b); }
            catch (...) { std::lock_guard<std::mutex> g(mu); err = std::current_exception(); }
        });
    for (auto& x : th) x.join();
    if (err) std::rethrow_exception(err);
}

// Scatter per-block host arrays into a zero-padded device array of `len` doubles through two page-locked bounce buffers:
// while chunk k travels over PCIe (hipMemcpyAsync from pinned memory) host threads assemble chunk k + 1.
struct StagedUpload {
    static constexpr int64_t CHUNK = int64_t(1) << 23;  // 8 Mi doubles = 64 MiB
    hipStream_t stream;
    PinnedBuf<double> pin[2];
    hipEvent_t done[2] = {nullptr, nullptr};
    explicit StagedUpload(hipStream_t s) : stream(s) {
        for (auto& ev : done) CBA_HIP(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    }
    ~StagedUpload() { for (auto& ev : done) if (ev) (void)hipEventDestroy(ev); }
    std::vector<double> small;  // arrays under 4 MiB: one pageable staging buffer (page-locking memory costs ~1 ms, more than it saves)
    void reserve(int64_t longest) {
        if (longest < (int64_t(1) << 19)) return;
        const size_t c = static_cast<size_t>(std::min<int64_t>(CHUNK, longest));
        pin[0].reserve(c); pin[1].reserve(c);
    }
    // source_of(b)[i * stride] = element i of block b (stride 1: SoA input; 4: interleaved {X, Y, u, v} records)
    static void gather(double* dst, const double* src, int64_t n, int64_t stride) {
        if (stride == 1) { std::memcpy(dst, src, sizeof(double) * static_cast<size_t>(n)); return; }
        for (int64_t i = 0; i < n; ++i) dst[i] = src[i * stride];
    }
    template <class Off, class Len, class Src>
    void run(double* dst, int64_t len, int n_blocks, Off&& offset_of, Len&& length_of, Src&& source_of, int64_t stride = 1) {
        // blocks in increasing destination order (offsets are monotone in b for the blocks that are stored)
        std::vector<int> order;
        for (int b = 0; b < n_blocks; ++b) if (offset_of(b) >= 0) order.push_back(b);
        if (!pin[0].p) {
            small.assign(static_cast<size_t>(len), 0.0);
            for (int b : order) gather(&small[static_cast<size_t>(offset_of(b))], source_of(b), length_of(b), stride);
            CBA_HIP(hipMemcpyAsync(dst, small.data(), sizeof(double) * static_cast<size_t>(len), hipMemcpyHostToDevice, stream));
            CBA_HIP(hipStreamSynchronize(stream));
            return;
        }
        const int64_t chunk = static_cast<int64_t>(pin[0].n);
        size_t first = 0;  // first block that may intersect the current chunk
        int k = 0;
        for (int64_t c0 = 0; c0 < len; c0 += chunk, ++k) {
            const int64_t c1 = std::min(len, c0 + chunk);
            double* buf = pin[k & 1].p;
            if (k >= 2) CBA_HIP(hipEventSynchronize(done[k & 1]));  // the copy that last used this bounce buffer
            while (first < order.size() && offset_of(order[first]) + length_of(order[first]) <= c0) ++first;
            size_t last = first;
            while (last < order.size() && offset_of(order[last]) < c1) ++last;
            std::memset(buf, 0, sizeof(double) * static_cast<size_t>(c1 - c0));
            const int nb = static_cast<int>(last - first);
            auto copy_block = [&](int i) {
                const int b = order[first + i];
                const int64_t o = offset_of(b), n = length_of(b);
                const int64_t lo = std::max(o, c0), hi = std::min(o + n, c1);
                if (hi > lo) gather(buf + (lo - c0), source_of(b) + (lo - o) * stride, hi - lo, stride);
            };
            if (c1 - c0 >= (int64_t(1) << 20)) parallel_blocks(nb, copy_block);
            else for (int i = 0; i < nb; ++i) copy_block(i);
            CBA_HIP(hipMemcpyAsync(dst + c0, buf, sizeof(double) * static_cast<size_t>(c1 - c0), hipMemcpyHostToDevice, stream));
            CBA_HIP(hipEventRecord(done[k & 1], stream));
        }
        CBA_HIP(hipStreamSynchronize(stream));
    }
};

// `aos` (optional): aos[b] = block b's observations as interleaved {object_x, object_y, image_u, image_v} records — the
// memory of the reference's std::vector<PlanarObservation> (include/calib/estimation/linear/planarpose.h:22-26) — read in
// place; d.X, d.Y, d.u, d.v are then ignored.  SURVEY.md §8(f) rank 4: no AoS -> SoA copy at the caller.
static void build_engine(const cba_reproj_problem& d, int device, Engine& e, const double* const* aos = nullptr) {
    PhaseTimer pt;
    if (const char* ev = getenv("CBA_EVAL_VARIANT")) e.eval_variant = atoi(ev);
    if (const char* eb = getenv("CBA_EVAL_BLOCKED")) e.eval_blocked = atoi(eb);
    Structure st;
    if (aos)
        for (int b = 0; b < d.n_blocks; ++b)
            if (!aos[b]) throw std::invalid_argument("missing observation records of a block");
    build_structure(d, st, aos != nullptr);  // validation mirroring the reference (SURVEY.md §8b "Errors")
    e.chain = st.chain; e.model = st.model;
    e.PI = st.PI; e.PL = st.PL; e.NACC = st.NACC;
    e.n_blocks = st.n_blocks; e.n_cams = st.n_cams; e.n_views = st.n_views;
    e.first_view_global = st.first_view_global;
    e.blk_offset = st.blk_offset; e.blk_cam = st.blk_cam; e.blk_view = st.blk_view;
    e.pad_offset.resize(d.n_blocks + 1);
    int64_t pad = 0;
    for (int b = 0; b < d.n_blocks; ++b) {
        e.pad_offset[b] = pad;
        pad += (e.blk_offset[b + 1] - e.blk_offset[b] + 1) & ~int64_t(1);
    }
    e.pad_offset[d.n_blocks] = pad;
    e.n_obs = st.n_obs;
    e.ld = (pad + 255) & ~int64_t(255);
    if (e.ld == 0) e.ld = 256;
    if (const char* lp = cba_exp_env("CBA_LD_MOD")) {
        // experiment: force (ld * 8) mod 2 MiB == CBA_LD_MOD bytes (multiple of 256)
        const int64_t period = (2 << 20) / 8, want = atoll(lp) / 8;
        int64_t ld = (e.ld / period) * period + want;
        while (ld < e.ld) ld += period;
        e.ld = ld;
    }

    const int ndev = device_count();
    if (ndev <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    e.device = device;
    CBA_HIP(hipSetDevice(device));
    e.stream = cache_stream();
    CBA_HIP(hipEventCreate(&e.ev0));
    CBA_HIP(hipEventCreate(&e.ev1));

    pt.lap("structure + device + stream");
    // ---- observations: padded SoA; X, Y deduplicated across blocks ------------------------------------
    {
        // residual blocks whose (X, Y) lists are bitwise identical share one device copy.  Hashing and the byte-for-byte
        // confirmation run on a few host threads (at C3 this is 2.6 GB of target points: 0.4 s on one core).
        e.xy_offset.assign(d.n_blocks, 0);
        const int64_t stride = aos ? 4 : 1;
        // component c (0 X, 1 Y, 2 u, 3 v) of block b: element i at comp(c, b)[i * stride]
        auto comp = [&](int c, int b) -> const double* {
            if (aos) return aos[b] + c;
            const double* base[4] = {d.X, d.Y, d.u, d.v};
            return base[c] + e.blk_offset[b];
        };
        auto same_bits = [&](const double* p, const double* q, int64_t n) {
            if (stride == 1) return std::memcmp(p, q, sizeof(double) * static_cast<size_t>(n)) == 0;
            for (int64_t i = 0; i < n; ++i)
                if (std::memcmp(p + i * stride, q + i * stride, 8) != 0) return false;
            return true;
        };
        std::vector<uint64_t> hashes(d.n_blocks);
        parallel_blocks(d.n_blocks, [&](int b) {
            const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b];
            uint64_t hsh = 1469598103934665603ULL ^ static_cast<uint64_t>(n);
            for (const double* p : {comp(0, b), comp(1, b)})
                for (int64_t i = 0; i < n; ++i) {
                    uint64_t w;
                    std::memcpy(&w, p + i * stride, 8);
                    hsh = (hsh ^ w) * 1099511628211ULL;
                    hsh ^= hsh >> 29;
                }
            hashes[b] = hsh;
        });
        // candidate owner = the first block with the same (hash, length); confirmed below
        std::unordered_map<uint64_t, int> first_of;
        std::vector<int> cand(d.n_blocks, -1);
        for (int b = 0; b < d.n_blocks; ++b) {
            const auto it = first_of.find(hashes[b]);
            if (it == first_of.end()) first_of.emplace(hashes[b], b);
            else cand[b] = it->second;
        }
        std::vector<char> same(d.n_blocks, 0);
        parallel_blocks(d.n_blocks, [&](int b) {
            const int c = cand[b];
            if (c < 0) return;
            const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b], nc = e.blk_offset[c + 1] - e.blk_offset[c];
            same[b] = nc == n && same_bits(comp(0, c), comp(0, b), n) && same_bits(comp(1, c), comp(1, b), n);
        });
        std::vector<int> owner(d.n_blocks, -1);
        int64_t xy_pad = 0;
        for (int b = 0; b < d.n_blocks; ++b) {
            if (cand[b] >= 0 && same[b]) {  // (a hash collision with different bytes simply keeps its own copy)
                e.xy_offset[b] = e.xy_offset[cand[b]];
            } else {
                const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b];
                owner[b] = b;
                e.xy_offset[b] = xy_pad;
                xy_pad += (n + 1) & ~int64_t(1);
                ++e.n_xy_unique_blocks;
            }
        }
        pt.lap("hash + dedup of X, Y");
        e.ld_xy = std::max<int64_t>(256, (xy_pad + 255) & ~int64_t(255));
        e.X.alloc(e.ld_xy); e.Y.alloc(e.ld_xy); e.u.alloc(e.ld); e.v.alloc(e.ld);
        // padded SoA arrays go up through two page-locked bounce buffers: host threads fill chunk k+1 while chunk k is in flight
        StagedUpload up(e.stream);
        up.reserve(std::max(e.ld, e.ld_xy));
        for (int a = 0; a < 2; ++a)
            up.run(a == 0 ? e.X.p : e.Y.p, e.ld_xy, d.n_blocks,
                   [&](int b2) { return owner[b2] == b2 ? e.xy_offset[b2] : int64_t(-1); },
                   [&](int b2) { return e.blk_offset[b2 + 1] - e.blk_offset[b2]; }, [&](int b2) { return comp(a, b2); }, stride);
        for (int a = 0; a < 2; ++a)
            up.run(a == 0 ? e.u.p : e.v.p, e.ld, d.n_blocks, [&](int b2) { return e.pad_offset[b2]; },
                   [&](int b2) { return e.blk_offset[b2 + 1] - e.blk_offset[b2]; }, [&](int b2) { return comp(2 + a, b2); }, stride);
    }
    pt.lap("stage + upload X, Y, u, v");
    // ---- tile tables ----------------------------------------------------------------------------
    {
        std::vector<Tile> ta, tb;
        e.blk_tile_off.assign(d.n_blocks + 1, 0);
        // Mode B / R tile length (structure.hpp choose_mode_b_tile): one tile per block at C2 (1000 blocks of 10 000) and C3 (32 000
        // blocks of 5000; also its 8-GPU share), five per block for 600 blocks of 10 000, 2048 observations for small problems
        const bool two_parts = d.chain == CBA_CHAIN_INTRINSIC && d.camera_model == CBA_CAMERA_PINHOLE_BC;
        int64_t tile_b = choose_mode_b_tile(d.n_blocks, d.n_blocks > 0 ? e.blk_offset[d.n_blocks] : 0, two_parts, TILE_B);
        if (d.n_blocks > 0) {
            const int64_t np_obs = two_parts ? 128 : 256;
            const int64_t n_avg = std::max<int64_t>(1, e.blk_offset[d.n_blocks] / d.n_blocks);
            if (const char* env = cba_exp_env("CBA_MODEB_TILES_PER_BLOCK")) {  // experiment builds: force k tiles per average block
                const int64_t k = std::max(1, std::atoi(env));
                tile_b = std::max<int64_t>(64, ((n_avg + k - 1) / k + np_obs - 1) / np_obs * np_obs);
            }
        }
        for (int b = 0; b < d.n_blocks; ++b) {
            const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b];
            const int64_t np = (n + 1) & ~int64_t(1);
            for (int64_t s = 0; s < np; s += TILE_A)
                ta.push_back(Tile{b, static_cast<int32_t>(std::min<int64_t>(TILE_A, np - s)), e.pad_offset[b] + s, e.xy_offset[b] + s, 0});
            e.blk_tile_off[b] = static_cast<int64_t>(tb.size());
            for (int64_t s = 0; s < n; s += tile_b)
                tb.push_back(Tile{b, static_cast<int32_t>(std::min<int64_t>(tile_b, n - s)), e.pad_offset[b] + s, e.xy_offset[b] + s, 0});
        }
        e.blk_tile_off[d.n_blocks] = static_cast<int64_t>(tb.size());
        e.n_tilesA = static_cast<int64_t>(ta.size());
        e.n_tilesB = static_cast<int64_t>(tb.size());
        e.max_tileB = 0;
        for (const Tile& t : tb) e.max_tileB = std::max(e.max_tileB, t.count);
        e.tilesA.alloc(ta.size()); e.tilesA.upload(ta.data(), ta.size(), e.stream);
        e.tilesB.alloc(tb.size()); e.tilesB.upload(tb.data(), tb.size(), e.stream);
        e.d_blk_tile_off.alloc(e.blk_tile_off.size());
        e.d_blk_tile_off.upload(e.blk_tile_off.data(), e.blk_tile_off.size(), e.stream);
        e.d_blk_cam.alloc(d.n_blocks); e.d_blk_cam.upload(e.blk_cam.data(), e.blk_cam.size(), e.stream);
        e.d_blk_view.alloc(d.n_blocks); e.d_blk_view.upload(e.blk_view.data(), e.blk_view.size(), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
    }
    pt.lap("tile tables");
    // ---- parameters -----------------------------------------------------------------------------
    e.h_intr.assign(d.intr, d.intr + static_cast<size_t>(d.n_cams) * e.PI);
    e.h_cam.assign(7 * static_cast<size_t>(d.n_cams), 0.0);
    e.h_target.assign(7, 0.0);
    if (d.chain != CBA_CHAIN_INTRINSIC) e.h_cam.assign(d.cam_pose, d.cam_pose + 7 * static_cast<size_t>(d.n_cams));
    if (d.chain != CBA_CHAIN_BUNDLE && d.n_views > 0) e.h_view.assign(d.view_pose, d.view_pose + 7 * static_cast<size_t>(d.n_views));
    if (d.chain == CBA_CHAIN_BUNDLE) e.h_target.assign(d.target_pose, d.target_pose + 7);
    {
        auto even = [](size_t n) { return (n + 1) & ~size_t(1); };
        e.pk_cam = even(e.h_intr.size());
        e.pk_target = e.pk_cam + even(e.h_cam.size());
        e.pk_delta = e.pk_target + 8;
        e.pk_size = e.pk_delta + even(static_cast<size_t>(st.nsh));
    }
    for (int k = 0; k < 2; ++k) {
        e.shared_pack[k].alloc(e.pk_size);
        e.shared_pack[k].zero(e.stream);
        e.intr[k].view(e.shared_pack[k].p, e.h_intr.size());
        e.cam[k].view(e.shared_pack[k].p + e.pk_cam, e.h_cam.size());
        e.target[k].view(e.shared_pack[k].p + e.pk_target, 7);
        e.view[k].alloc(std::max<size_t>(e.h_view.size(), 7));
    }
    e.delta_sh.view(e.shared_pack[1].p + e.pk_delta, static_cast<size_t>(st.nsh));
    upload_params(e);
    e.bc.alloc(static_cast<size_t>(d.n_blocks) * 36);
    e.sd.alloc(static_cast<size_t>(d.n_cams) * 36);
    e.sd.zero(e.stream);
    e.aux.alloc(static_cast<size_t>(d.n_blocks) * 12);
    if (d.chain == CBA_CHAIN_BUNDLE) e.aux.upload(d.blk_b_T_g, static_cast<size_t>(d.n_blocks) * 12, e.stream);
    e.partial.alloc(static_cast<size_t>(e.n_tilesB) * e.NACC);
    e.blk_acc.alloc(static_cast<size_t>(d.n_blocks) * e.NACC);
    if (d.chain != CBA_CHAIN_INTRINSIC) e.blk_mom.alloc(static_cast<size_t>(std::max(1, d.n_blocks)) * 256);
    if (const char* env = std::getenv("CBA_MODEB_MOMENTS")) e.modeb_moments = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_MODEB_SPLIT")) e.modeb_split = std::atoi(env);
    if (const char* env = cba_exp_env("CBA_MODEB_SHARED")) e.modeb_shared = std::atoi(env);
    e.blk_s.alloc(d.n_blocks);
    e.scalar_out.alloc(8);
    e.cost_part.alloc(static_cast<size_t>(2 * ((d.n_blocks + 2047) / 2048 + 1)));  // allocated here: launch_cost may run inside a graph capture
    CBA_HIP(hipStreamSynchronize(e.stream));
    pt.lap("parameters + buffers");
    init_lm_state(e, d, aos != nullptr);
    pt.lap("LM state");
    warm_lm(e);
    pt.lap("warm-up pass");
}

// The Mode A output block.  Where the driver puts a multi-GB buffer decides whether k_eval streams into it at 6.2 or at 6.5 TB/s:
// a property of the allocation, stable over its lifetime (tools/exp.py placement2: six handles of the same problem in one process
// 6.2 6.2 6.5 6.2 6.5 6.5 TB/s, the same again on re-measurement; shifting the output window inside a block by 256 B ... 64 MiB
// changes nothing) - the pages of a plain hipMalloc are scattered over the stacks differently every time.  A physically
// CONTIGUOUS block (hipExtMallocWithFlags, hipDeviceMallocContiguous) gets the interleaving the memory system was laid out for:
// 6.4 - 6.5 TB/s on every handle (18 of 18; plain: 4 of 18).  Up to 4 GiB only: such a block comes back in a millisecond, a larger
// one costs ~31 ms per GiB (tools/exp.py first_eval_c3: 6.9 GiB 0.21 s, 59 GB 1.7 s on the first evaluation) for the +0.4 ... 2 %
// it gains at those sizes.  Falls back to the plain allocation when the runtime cannot find a contiguous range
// (CBA_EVAL_CONTIGUOUS=0: always plain).
template <typename T>
static void alloc_output(DevBuf<T>& b, size_t count) {
    static const bool contiguous = [] { const char* v = getenv("CBA_EVAL_CONTIGUOUS"); return !(v && atoi(v) == 0); }();
    const size_t bytes = count * sizeof(T);
    if (contiguous && bytes >= (size_t(64) << 20) && bytes <= (size_t(4) << 30)) {
        void* p = nullptr;
        if (b.p && b.owned) (void)hipDeviceSynchronize();
        b.release();
        if (hipExtMallocWithFlags(&p, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            b.p = static_cast<T*>(p);
            b.n = count;
            b.granted = bytes;  // above the block cache's limit: goes back to the runtime with hipFree
            CBA_HIP(hipGetDevice(&b.device));
            return;
        }
        (void)hipGetLastError();
    }
    b.alloc(count);
}

extern "C" {

const char* cba_version(void) { return CBA_VERSION_STRING; }
const char* cba_last_error(void) { return g_err.c_str(); }
int32_t cba_device_count(void) { return device_count(); }
void cba_trim_cache(void) { try { cache_trim(); } catch (...) {} }
cba_status cba_set_device(int32_t device) {
    return guarded([&] {
        const int n = device_count();
        if (n <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        if (device < 0 || device >= n) throw std::invalid_argument("device index out of range");
        g_default_device.store(device);
    });
}
int32_t cba_get_device(void) { return default_device(); }

void cba_options_default(cba_options* o) {
    std::memset(o, 0, sizeof(*o));
    o->optimizer = 0;
    o->max_iterations = 1000;  // optimize.h:26
    o->huber_delta = 1.0;      // optimize.h:28
    o->epsilon = 1e-9;         // optimize.h:25
    o->compute_covariance = 1;
    o->verbose = 0;
    o->optimize_intrinsics = 1;
    o->optimize_skew = 0;
    o->optimize_extrinsics = 1;
    o->optimize_target_pose = 1;
}

int32_t cba_intrinsics_size(int32_t camera_model) { return camera_model == CBA_CAMERA_SCHEIMPFLUG ? 12 : 10; }
int32_t cba_local_columns(int32_t chain, int32_t camera_model) {
    return (chain == CBA_CHAIN_INTRINSIC ? 6 : 12) + cba_intrinsics_size(camera_model);
}

// Eigen::Quaterniond(Matrix3d) (third-party; restated) — populate_quat_tran, observationutils.h:43-48
void cba_pose_from_matrix(const double* m, double* p) {
    auto M = [&](int r, int c) { return m[c * 4 + r]; };  // column-major 4x4
    double q[4];
    double t = M(0, 0) + M(1, 1) + M(2, 2);
    if (t > 0.0) {
        t = std::sqrt(t + 1.0);
        q[0] = 0.5 * t;
        t = 0.5 / t;
        q[1] = (M(2, 1) - M(1, 2)) * t;
        q[2] = (M(0, 2) - M(2, 0)) * t;
        q[3] = (M(1, 0) - M(0, 1)) * t;
    } else {
        int i = 0;
        if (M(1, 1) > M(0, 0)) i = 1;
        if (M(2, 2) > M(i, i)) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(M(i, i) - M(j, j) - M(k, k) + 1.0);
        q[1 + i] = 0.5 * t;
        t = 0.5 / t;
        q[0] = (M(k, j) - M(j, k)) * t;
        q[1 + j] = (M(j, i) + M(i, j)) * t;
        q[1 + k] = (M(k, i) + M(i, k)) * t;
    }
    for (int i = 0; i < 4; ++i) p[i] = q[i];
    p[4] = M(0, 3); p[5] = M(1, 3); p[6] = M(2, 3);
}

// restore_pose, observationutils.h:50-62 (normalise, then Eigen toRotationMatrix)
void cba_pose_to_matrix(const double* p, double* m) {
    const double n = std::sqrt(p[0] * p[0] + p[1] * p[1] + p[2] * p[2] + p[3] * p[3]);
    const double w = p[0] / n, x = p[1] / n, y = p[2] / n, z = p[3] / n;
    const double tx = 2 * x, ty = 2 * y, tz = 2 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w, txx = tx * x, txy = ty * x, txz = tz * x, tyy = ty * y,
                 tyz = tz * y, tzz = tz * z;
    const double R[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                         txz - twy, tyz + twx, 1 - (txx + tyy)};
    for (int i = 0; i < 16; ++i) m[i] = 0.0;
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) m[c * 4 + r] = R[3 * r + c];
    m[12] = p[4]; m[13] = p[5]; m[14] = p[6]; m[15] = 1.0;
}

cba_status cba_reproj_create(const cba_reproj_problem* desc, int32_t device, cba_reproj** out) {
    return guarded([&] {
        if (!desc || !out) throw std::invalid_argument("null argument");
        auto e = std::make_unique<Engine>();
        build_engine(*desc, device, *e);
        *out = reinterpret_cast<cba_reproj*>(e.release());
    });
}

cba_status cba_reproj_create_aos(const cba_reproj_problem* desc, const double* const* blk_obs, int32_t device, cba_reproj** out) {
    return guarded([&] {
        if (!desc || !out || (desc->n_blocks > 0 && !blk_obs)) throw std::invalid_argument("null argument");
        auto e = std::make_unique<Engine>();
        build_engine(*desc, device, *e, blk_obs);
        *out = reinterpret_cast<cba_reproj*>(e.release());
    });
}

void cba_reproj_destroy(cba_reproj* h) {
    if (!h) return;
    Engine* e = reinterpret_cast<Engine*>(h);
    (void)hipSetDevice(e->device);
    if (e->stream) (void)hipStreamSynchronize(e->stream);
    delete e;
}

cba_status cba_reproj_set_params(cba_reproj* h, const double* intr, const double* cam_pose, const double* view_pose,
                                 const double* target_pose) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        if (intr) std::memcpy(e.h_intr.data(), intr, sizeof(double) * e.h_intr.size());
        if (cam_pose && e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(e.h_cam.data(), cam_pose, sizeof(double) * e.h_cam.size());
        if (view_pose && !e.h_view.empty()) std::memcpy(e.h_view.data(), view_pose, sizeof(double) * e.h_view.size());
        if (target_pose && e.chain == CBA_CHAIN_BUNDLE) std::memcpy(e.h_target.data(), target_pose, sizeof(double) * 7);
        upload_params(e);
        CBA_HIP(hipStreamSynchronize(e.stream));
    });
}

cba_status cba_reproj_get_params(cba_reproj* h, double* intr, double* cam_pose, double* view_pose, double* target_pose) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (intr) std::memcpy(intr, e.h_intr.data(), sizeof(double) * e.h_intr.size());
        if (cam_pose && e.chain != CBA_CHAIN_INTRINSIC) std::memcpy(cam_pose, e.h_cam.data(), sizeof(double) * e.h_cam.size());
        if (view_pose && !e.h_view.empty()) std::memcpy(view_pose, e.h_view.data(), sizeof(double) * e.h_view.size());
        if (target_pose && e.chain == CBA_CHAIN_BUNDLE) std::memcpy(target_pose, e.h_target.data(), sizeof(double) * 7);
    });
}

int64_t cba_reproj_num_observations(const cba_reproj* h) { return h ? reinterpret_cast<const Engine*>(h)->n_obs : 0; }

static void ensure_eval_buffers(Engine& e) {
    if (e.scalar) {
        const size_t jn = static_cast<size_t>(e.n_tilesA) * (2 + 2 * e.PL) * TILE_A;
        if (e.Jf.n < jn) alloc_output(e.Jf, jn);
        return;
    }
    if (!e.eval_blocked && e.r.n < static_cast<size_t>(2 * e.ld)) e.r.alloc(static_cast<size_t>(2 * e.ld));
    const size_t tw = static_cast<size_t>(2 + 2 * e.PL) * TILE_A;
    const size_t jn = e.eval_blocked ? static_cast<size_t>(e.n_tilesA) * tw : static_cast<size_t>(2 * e.PL) * e.ld;
    // (measured at C3, 59 GB: 15 segments 11.12 ms per pass against 10.66 ms for one plain block - the fifteen launches' ramps and tails
    // cost more than the placement gains: an experiment knob, CBA_EVAL_SEGMENTS=1 in an EXPERIMENTS build)
    static const bool segmented = [] { const char* v = cba_exp_env("CBA_EVAL_SEGMENTS"); return v && atoi(v) != 0; }();
    if (e.eval_blocked && segmented && jn * sizeof(double) > (size_t(4) << 30)) {  // segments of whole tiles, each <= 4 GiB and contiguous
        const int64_t per = static_cast<int64_t>((size_t(4) << 30) / (tw * sizeof(double))) / 64 * 64;
        const size_t nseg = static_cast<size_t>((e.n_tilesA + per - 1) / per);
        if (e.seg_tiles != per || e.Jseg.size() != nseg) {
            e.J.release();
            e.Jseg.clear();
            e.Jseg.resize(nseg);
            for (size_t k = 0; k < nseg; ++k) {
                const int64_t nt = std::min<int64_t>(per, e.n_tilesA - static_cast<int64_t>(k) * per);
                alloc_output(e.Jseg[k], static_cast<size_t>(nt) * tw);
            }
            e.seg_tiles = per;
        }
        return;
    }
    if (!e.Jseg.empty()) { e.Jseg.clear(); e.seg_tiles = 0; }
    if (e.J.n < jn) alloc_output(e.J, jn);
}

cba_status cba_reproj_eval(cba_reproj* h) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        ensure_eval_buffers(e);
        launch_block_consts(e, 0);
        launch_eval(e);
        e.eval_done = 1;
        e.eval_blocked_last = e.eval_blocked;
        CBA_HIP(hipStreamSynchronize(e.stream));
    });
}

cba_status cba_reproj_eval_timed(cba_reproj* h, int32_t warmup, int32_t iters, double* ms_per_eval) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (iters <= 0) throw std::invalid_argument("iters must be positive");
        CBA_HIP(hipSetDevice(e.device));
        // tuning knobs are re-read here so one handle (one set of buffers) can time every variant
        if (const char* ev = getenv("CBA_EVAL_VARIANT")) e.eval_variant = atoi(ev);
        if (const char* eb = getenv("CBA_EVAL_BLOCKED")) e.eval_blocked = atoi(eb);
        if (const char* ea = cba_exp_env("CBA_EVAL_ABLATE")) e.eval_ablate = atoi(ea);
        ensure_eval_buffers(e);
        launch_block_consts(e, 0);
        for (int i = 0; i < warmup; ++i) launch_eval(e);
        CBA_HIP(hipEventRecord(e.ev0, e.stream));
        for (int i = 0; i < iters; ++i) launch_eval(e);
        CBA_HIP(hipEventRecord(e.ev1, e.stream));
        CBA_HIP(hipEventSynchronize(e.ev1));
        e.eval_done = 1;
        e.eval_blocked_last = e.eval_blocked;
        float ms = 0.f;
        CBA_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
        *ms_per_eval = static_cast<double>(ms) / iters;
    });
}

cba_status cba_reproj_normal_eq_timed(cba_reproj* h, int32_t warmup, int32_t iters, double* ms_per_pass) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (iters <= 0) throw std::invalid_argument("iters must be positive");
        CBA_HIP(hipSetDevice(e.device));
        launch_block_consts(e, 0);
        for (int i = 0; i < warmup; ++i) launch_normal_eq(e);
        CBA_HIP(hipEventRecord(e.ev0, e.stream));
        for (int i = 0; i < iters; ++i) launch_normal_eq(e);
        CBA_HIP(hipEventRecord(e.ev1, e.stream));
        CBA_HIP(hipEventSynchronize(e.ev1));
        float ms = 0.f;
        CBA_HIP(hipEventElapsedTime(&ms, e.ev0, e.ev1));
        *ms_per_pass = static_cast<double>(ms) / iters;
    });
}

// tile-blocked layout out[tile][2 + 2P][128]: the residuals / Jacobian rows of residual blocks [b0, b1) into r / J, indexed
// from the first observation of block b0 (walks the host copy of the tile table; one 34-KiB copy per tile)
static void fetch_blocked_range(Engine& e, int b0, int b1, double* r, double* J) {
    const int P = e.PL;
    const int64_t tw = static_cast<int64_t>(2 + 2 * P) * TILE_A;
    std::vector<double> buf(static_cast<size_t>(tw));
    int64_t w = 0;
    for (int b = 0; b < b0; ++b) w += (e.blk_offset[b + 1] - e.blk_offset[b] + TILE_A - 1) / TILE_A;
    const int64_t base = e.blk_offset[b0];
    for (int b = b0; b < b1; ++b) {
        const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b];
        for (int64_t s0 = 0; s0 < n; s0 += TILE_A, ++w) {
            CBA_HIP(hipMemcpyAsync(buf.data(), e.eval_tile_ptr(w, tw), sizeof(double) * tw, hipMemcpyDeviceToHost, e.stream));
            CBA_HIP(hipStreamSynchronize(e.stream));
            const int64_t cnt = std::min<int64_t>(TILE_A, n - s0);
            for (int64_t j = 0; j < cnt; ++j) {
                const int64_t i = e.blk_offset[b] + s0 + j - base;
                if (r) { r[2 * i] = buf[j]; r[2 * i + 1] = buf[TILE_A + j]; }
                if (J)
                    for (int k = 0; k < P; ++k) {
                        J[(2 * i) * P + k] = buf[(2 + k) * TILE_A + j];
                        J[(2 * i + 1) * P + k] = buf[(2 + P + k) * TILE_A + j];
                    }
            }
        }
    }
}

cba_status cba_reproj_eval_fetch_blocks(cba_reproj* h, int32_t b0, int32_t b1, double* r, double* J) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        if (e.scalar) throw std::runtime_error("fp32 arithmetic selected: use cba_reproj_eval_fetch_f32");
        if ((e.J.n == 0 && e.Jseg.empty()) || !e.eval_done) throw std::runtime_error("cba_reproj_eval has not been called");
        if (b0 < 0 || b1 < b0 || b1 > e.n_blocks) throw std::invalid_argument("block range outside the problem");
        if (!e.eval_blocked_last) throw std::runtime_error("block-range fetch needs the tile-blocked output layout (the default)");
        fetch_blocked_range(e, b0, b1, r, J);
    });
}

cba_status cba_reproj_eval_fetch(cba_reproj* h, double* r, double* J) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        if (e.scalar) throw std::runtime_error("fp32 arithmetic selected: use cba_reproj_eval_fetch_f32");
        if ((e.J.n == 0 && e.Jseg.empty()) || !e.eval_done) throw std::runtime_error("cba_reproj_eval has not been called");
        const int P = e.PL;
        if (e.eval_blocked_last) {
            fetch_blocked_range(e, 0, e.n_blocks, r, J);
            return;
        }
        std::vector<double> hr(static_cast<size_t>(2 * e.ld));
        e.r.download(hr.data(), hr.size(), e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
        if (r)
            for (int b = 0; b < e.n_blocks; ++b)
                for (int64_t i = e.blk_offset[b]; i < e.blk_offset[b + 1]; ++i) {
                    const int64_t pi = e.pad_offset[b] + (i - e.blk_offset[b]);
                    r[2 * i] = hr[pi];
                    r[2 * i + 1] = hr[e.ld + pi];
                }
        if (J) {
            std::vector<double> row(static_cast<size_t>(e.ld));
            for (int k = 0; k < 2 * P; ++k) {
                CBA_HIP(hipMemcpyAsync(row.data(), e.J.p + static_cast<size_t>(k) * e.ld, sizeof(double) * e.ld,
                                       hipMemcpyDeviceToHost, e.stream));
                CBA_HIP(hipStreamSynchronize(e.stream));
                const int uvrow = k / P, col = k % P;
                for (int b = 0; b < e.n_blocks; ++b)
                    for (int64_t i = e.blk_offset[b]; i < e.blk_offset[b + 1]; ++i)
                        J[(2 * i + uvrow) * P + col] = row[e.pad_offset[b] + (i - e.blk_offset[b])];
            }
        }
    });
}

cba_status cba_reproj_set_scalar(cba_reproj* h, int32_t scalar) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (scalar != 0 && scalar != 1) throw std::invalid_argument("scalar must be 0 (fp64) or 1 (fp32)");
        CBA_HIP(hipSetDevice(e.device));
        if (scalar) ensure_f32_buffers(e);
        e.scalar = scalar;
        e.eval_done = 0;
        if (scalar) warm_lm(e);  // the fp32 kernel family is a code object of its own: set it up here, not inside the first fp32 solve
    });
}

cba_status cba_reproj_eval_fetch_f32(cba_reproj* h, float* r, float* J) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        if (!e.scalar || e.Jf.n == 0 || !e.eval_done) throw std::runtime_error("no fp32 evaluation available");
        const int P = e.PL;
        const int64_t tw = static_cast<int64_t>(2 + 2 * P) * TILE_A;
        std::vector<float> buf(static_cast<size_t>(tw));
        int64_t w = 0;
        for (int b = 0; b < e.n_blocks; ++b) {
            const int64_t n = e.blk_offset[b + 1] - e.blk_offset[b];
            for (int64_t s0 = 0; s0 < n; s0 += TILE_A, ++w) {
                CBA_HIP(hipMemcpyAsync(buf.data(), e.Jf.p + w * tw, sizeof(float) * tw, hipMemcpyDeviceToHost, e.stream));
                CBA_HIP(hipStreamSynchronize(e.stream));
                const int64_t cnt = std::min<int64_t>(TILE_A, n - s0);
                for (int64_t j = 0; j < cnt; ++j) {
                    const int64_t i = e.blk_offset[b] + s0 + j;
                    if (r) { r[2 * i] = buf[j]; r[2 * i + 1] = buf[TILE_A + j]; }
                    if (J)
                        for (int k = 0; k < P; ++k) {
                            J[(2 * i) * P + k] = buf[(2 + k) * TILE_A + j];
                            J[(2 * i + 1) * P + k] = buf[(2 + P + k) * TILE_A + j];
                        }
                }
            }
        }
    });
}

cba_status cba_reproj_cost(cba_reproj* h, double huber_delta, double* cost) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        launch_block_consts(e, 0);
        launch_resid(e);
        launch_cost(e, huber_delta);
        double out[2];
        e.scalar_out.download(out, 2, e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
        engine_allreduce(e, out, 1);
        *cost = out[0];
    });
}

int64_t cba_reproj_block_normal_eq_size(const cba_reproj* h) { return h ? reinterpret_cast<const Engine*>(h)->NACC : 0; }

cba_status cba_reproj_block_normal_eq(cba_reproj* h, double* out) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        launch_block_consts(e, 0);
        launch_normal_eq(e);
        e.blk_acc.download(out, static_cast<size_t>(e.n_blocks) * e.NACC, e.stream);
        CBA_HIP(hipStreamSynchronize(e.stream));
    });
}

cba_status cba_reproj_solve(cba_reproj* h, const cba_options* opts, cba_summary* summary) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (!opts || !summary) throw std::invalid_argument("null argument");
        CBA_HIP(hipSetDevice(e.device));
        solve_lm(e, *opts, summary);
    });
}

cba_status cba_reproj_set_lm_mode(cba_reproj* h, int32_t mode) {
    return guarded([&] { set_lm_mode(*as_engine(h), mode); });
}

cba_status cba_reproj_solve_stats(const cba_reproj* h, int64_t stats8[8]) {
    return guarded([&] {
        if (!h || !stats8) throw std::invalid_argument("null argument");
        solve_stats(*reinterpret_cast<const Engine*>(h), stats8);
    });
}

int64_t cba_reproj_covariance_dim(const cba_reproj* h) { return h ? covariance_dim(*reinterpret_cast<const Engine*>(h)) : 0; }

cba_status cba_reproj_covariance(cba_reproj* h, const cba_options* opts, double* cov) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (!opts || !cov) throw std::invalid_argument("null argument");
        CBA_HIP(hipSetDevice(e.device));
        compute_covariance(e, *opts, cov);
    });
}

int64_t cba_reproj_covariance_shared_dim(const cba_reproj* h) { return h ? shared_covariance_dim(*reinterpret_cast<const Engine*>(h)) : 0; }

cba_status cba_reproj_covariance_shared(cba_reproj* h, const cba_options* opts, double* cov) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (!opts || !cov) throw std::invalid_argument("null argument");
        CBA_HIP(hipSetDevice(e.device));
        compute_covariance(e, *opts, cov, true);
    });
}

cba_status cba_reproj_covariance_views(cba_reproj* h, const cba_options* opts, int32_t n_sel, const int32_t* view_idx, double* cov7x7) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (!opts || n_sel < 0 || (n_sel > 0 && (!view_idx || !cov7x7))) throw std::invalid_argument("null argument");
        if (e.chain == CBA_CHAIN_BUNDLE) throw std::invalid_argument("the bundle chain has no per-view poses");
        CBA_HIP(hipSetDevice(e.device));
        if (n_sel > 0) compute_covariance_views(e, *opts, view_idx, n_sel, cov7x7);
    });
}

cba_status cba_reproj_set_allreduce(cba_reproj* h, cba_allreduce_fn fn, void* user, int32_t n_ranks, int32_t rank) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
        e.allreduce = fn;
        e.allreduce_user = user;
        e.n_ranks = n_ranks;
        e.rank = rank;
    });
}

cba_status cba_rccl_unique_id(uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES]) {
    return guarded([&] { rccl_unique_id(id); });
}

cba_status cba_reproj_init_rccl(cba_reproj* h, const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES], int32_t n_ranks, int32_t rank) {
    return guarded([&] {
        Engine& e = *as_engine(h);
        CBA_HIP(hipSetDevice(e.device));
        rccl_init(e, id, n_ranks, rank);
    });
}

// ---- one-shot entry points -----------------------------------------------------------------------
static void one_shot(const cba_reproj_problem& d, const cba_options* opts, cba_summary* summary, double* cov) {
    if (!opts || !summary) throw std::invalid_argument("null argument");
    PhaseTimer pt;
    auto e = std::make_unique<Engine>();
    build_engine(d, default_device(), *e);
    pt.lap("one-shot: handle");
    solve_lm(*e, *opts, summary);
    pt.lap("one-shot: solve");
    if (d.intr) std::memcpy(d.intr, e->h_intr.data(), sizeof(double) * e->h_intr.size());
    if (d.cam_pose && d.chain != CBA_CHAIN_INTRINSIC) std::memcpy(d.cam_pose, e->h_cam.data(), sizeof(double) * e->h_cam.size());
    if (d.view_pose && !e->h_view.empty()) std::memcpy(d.view_pose, e->h_view.data(), sizeof(double) * e->h_view.size());
    if (d.target_pose && d.chain == CBA_CHAIN_BUNDLE) std::memcpy(d.target_pose, e->h_target.data(), sizeof(double) * 7);
    if (cov && opts->compute_covariance) {
        try {
            compute_covariance(*e, *opts, cov);
        } catch (const HipError&) {
            throw;
        } catch (const std::runtime_error&) {  // rank deficient: the reference leaves the matrix empty
            const int64_t n = covariance_dim(*e);
            std::memset(cov, 0, sizeof(double) * static_cast<size_t>(n * n));
        }
    }
    CBA_HIP(hipStreamSynchronize(e->stream));
    pt.lap("one-shot: covariance");
    e.reset();
    pt.lap("one-shot: release");
}

cba_status cba_optimize_intrinsics(int32_t camera_model, int32_t n_views, const int64_t* view_offset, const double* X,
                                   const double* Y, const double* u, const double* v, double* intr, double* c_T_t,
                                   const cba_options* opts, cba_summary* summary, double* cov) {
    return guarded([&] {
        if (n_views < 4)  // intrinsics.cpp:92-96
            throw std::invalid_argument("Insufficient views for calibration (at least 4 required).");
        cba_reproj_problem d;
        std::memset(&d, 0, sizeof(d));
        d.chain = CBA_CHAIN_INTRINSIC; d.camera_model = camera_model;
        d.n_blocks = n_views; d.n_cams = 1; d.n_views = n_views;
        d.blk_offset = view_offset; d.X = X; d.Y = Y; d.u = u; d.v = v;
        d.intr = intr; d.view_pose = c_T_t;
        cba_options o = *opts;
        o.optimize_intrinsics = 1;
        one_shot(d, &o, summary, cov);
    });
}

cba_status cba_optimize_extrinsics(int32_t camera_model, int32_t n_cams, int32_t n_views, int32_t n_blocks,
                                   const int64_t* blk_offset, const int32_t* blk_view, const int32_t* blk_cam,
                                   const double* X, const double* Y, const double* u, const double* v, double* intr,
                                   double* c_T_r, double* r_T_t, const cba_options* opts, cba_summary* summary,
                                   double* cov) {
    return guarded([&] {
        cba_reproj_problem d;
        std::memset(&d, 0, sizeof(d));
        d.chain = CBA_CHAIN_EXTRINSIC; d.camera_model = camera_model;
        d.n_blocks = n_blocks; d.n_cams = n_cams; d.n_views = n_views;
        d.blk_offset = blk_offset; d.blk_view = blk_view; d.blk_cam = blk_cam;
        d.X = X; d.Y = Y; d.u = u; d.v = v;
        d.intr = intr; d.cam_pose = c_T_r; d.view_pose = r_T_t;
        one_shot(d, opts, summary, cov);
    });
}

cba_status cba_optimize_bundle(int32_t camera_model, int32_t n_cams, int32_t n_blocks, const int64_t* blk_offset,
                               const int32_t* blk_cam, const double* blk_b_T_g, const double* X, const double* Y,
                               const double* u, const double* v, double* intr, double* g_T_c, double* b_T_t,
                               const cba_options* opts, cba_summary* summary, double* cov) {
    return guarded([&] {
        cba_reproj_problem d;
        std::memset(&d, 0, sizeof(d));
        d.chain = CBA_CHAIN_BUNDLE; d.camera_model = camera_model;
        d.n_blocks = n_blocks; d.n_cams = n_cams; d.n_views = 0;
        d.blk_offset = blk_offset; d.blk_cam = blk_cam; d.blk_b_T_g = blk_b_T_g;
        d.X = X; d.Y = Y; d.u = u; d.v = v;
        d.intr = intr; d.cam_pose = g_T_c; d.target_pose = b_T_t;
        one_shot(d, opts, summary, cov);
    });
}

cba_status cba_optimize_handeye(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target, double* g_T_c,
                                const cba_options* opts, cba_summary* summary, double* cov) {
    return guarded([&] {
        if (!opts || !summary || !g_T_c) throw std::invalid_argument("null argument");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        handeye_solve(n_poses, base_T_gripper, cam_T_target, g_T_c, opts, summary, cov, default_device());
    });
}

cba_status cba_estimate_handeye_dlt(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target, double min_angle_deg,
                                    double* g_T_c) {
    return guarded([&] {
        if (!g_T_c) throw std::invalid_argument("null argument");
        if (n_poses < 2 || !base_T_gripper || !cam_T_target) throw std::runtime_error("Inconsistent hand-eye input sizes");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        handeye_dlt(n_poses, base_T_gripper, cam_T_target, min_angle_deg, g_T_c, default_device());
    });
}

cba_status cba_estimate_and_optimize_handeye(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                             double min_angle_deg, double* g_T_c, const cba_options* opts, cba_summary* summary,
                                             double* cov) {
    const cba_status st = cba_estimate_handeye_dlt(n_poses, base_T_gripper, cam_T_target, min_angle_deg, g_T_c);
    if (st != CBA_OK) return st;
    return cba_optimize_handeye(n_poses, base_T_gripper, cam_T_target, g_T_c, opts, summary, cov);
}

cba_status cba_estimate_and_optimize_handeye_sharded(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                                     double min_angle_deg, int32_t estimate, double* g_T_c, const cba_options* opts,
                                                     cba_summary* summary, double* cov, cba_allreduce_fn fn, void* user, int32_t n_ranks,
                                                     int32_t rank, int32_t device) {
    return guarded([&] {
        if (!opts || !summary || !g_T_c) throw std::invalid_argument("null argument");
        const int ndev = device_count();
        if (ndev <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
        if (estimate) handeye_dlt(n_poses, base_T_gripper, cam_T_target, min_angle_deg, g_T_c, device, fn, user, n_ranks, rank);
        handeye_solve(n_poses, base_T_gripper, cam_T_target, g_T_c, opts, summary, cov, device, fn, user, n_ranks, rank);
    });
}

cba_status cba_estimate_and_optimize_handeye_rccl(int32_t n_poses, const double* base_T_gripper, const double* cam_T_target,
                                                  double min_angle_deg, int32_t estimate, double* g_T_c, const cba_options* opts,
                                                  cba_summary* summary, double* cov, const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES],
                                                  int32_t n_ranks, int32_t rank, int32_t device) {
    return guarded([&] {
        if (!opts || !summary || !g_T_c || !id) throw std::invalid_argument("null argument");
        const int ndev = device_count();
        if (ndev <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
        CBA_HIP(hipSetDevice(device));
        void* comm = rccl_comm_create(id, n_ranks, rank);
        try {
            if (estimate) handeye_dlt(n_poses, base_T_gripper, cam_T_target, min_angle_deg, g_T_c, device, nullptr, nullptr, n_ranks, rank, comm);
            handeye_solve(n_poses, base_T_gripper, cam_T_target, g_T_c, opts, summary, cov, device, nullptr, nullptr, n_ranks, rank, comm);
        } catch (...) {
            rccl_comm_destroy(comm, true);  // this rank leaves: its peers' collectives must fail, not hang
            throw;
        }
        rccl_comm_destroy(comm, false);
    });
}

cba_status cba_optimize_planar_pose_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                          const double* u, const double* v, const double* kmtx5, int32_t num_radial,
                                          double* pose7, const cba_options* opts, cba_summary* summaries, double* distortion,
                                          double* reprojection_error, double* cov36) {
    return guarded([&] {
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        planar_pose_batch(n_views, view_offset, X, Y, u, v, kmtx5, num_radial, pose7, opts, summaries, distortion, reprojection_error,
                          cov36, default_device());
    });
}

cba_status cba_optimize_planar_pose(int32_t n, const double* X, const double* Y, const double* u, const double* v,
                                    const double* kmtx5, int32_t num_radial, double* pose7, const cba_options* opts,
                                    cba_summary* summary, double* distortion, double* reprojection_error, double* cov36) {
    const int64_t off[2] = {0, n};
    return cba_optimize_planar_pose_batch(1, off, X, Y, u, v, kmtx5, num_radial, pose7, opts, summary, distortion, reprojection_error,
                                          cov36);
}

cba_status cba_optimize_homography_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                         const double* u, const double* v, double* h9, const cba_options* opts,
                                         cba_summary* summaries, double* cov64) {
    return guarded([&] {
        if (!view_offset || !X || !Y || !u || !v || !h9 || !opts) throw std::invalid_argument("null argument");
        if (n_views <= 0) throw std::invalid_argument("At least 4 correspondences are required.");
        for (int i = 0; i < n_views; ++i)  // homography.cpp:146-148, checked before any device work like the reference
            if (view_offset[i + 1] - view_offset[i] < 4) throw std::invalid_argument("At least 4 correspondences are required.");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        homography_batch(n_views, view_offset, X, Y, u, v, h9, opts, summaries, cov64, default_device());
    });
}

cba_status cba_optimize_homography(int32_t n, const double* X, const double* Y, const double* u, const double* v, double* h9,
                                   const cba_options* opts, cba_summary* summary, double* cov64) {
    const int64_t off[2] = {0, n};
    return cba_optimize_homography_batch(1, off, X, Y, u, v, h9, opts, summary, cov64);
}

cba_status cba_optimize_intrinsics_semidlt(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                           const double* u, const double* v, double* kmtx5, double* c_T_t, int32_t num_radial,
                                           const double* bounds_lo5, const double* bounds_hi5, const int32_t* fixed_idx,
                                           const double* fixed_val, int32_t n_fixed, const cba_options* opts, cba_summary* summary,
                                           double* distortion, double* view_errors, double* cov) {
    return guarded([&] {
        if (!opts || !summary) throw std::invalid_argument("null argument");
        if (n_views < 4) {  // intrinsicssemidlt.cpp:163-166: message on stderr and a default-constructed result, no exception
            std::memset(summary, 0, sizeof(*summary));
            summary->termination = CBA_TERM_FAILURE;
            std::snprintf(summary->report, sizeof(summary->report), "Insufficient views for calibration (at least 4 required).");
            return;
        }
        if (!view_offset || !X || !Y || !u || !v || !kmtx5 || !c_T_t) throw std::invalid_argument("null argument");
        if (num_radial < 0 || num_radial > 3) throw std::invalid_argument("num_radial must be in [0, 3]");
        if ((bounds_lo5 == nullptr) != (bounds_hi5 == nullptr)) throw std::invalid_argument("bounds need both ends");
        if (n_fixed < 0 || (n_fixed > 0 && !fixed_idx)) throw std::invalid_argument("bad fixed distortion list");
        for (int i = 0; i < n_views; ++i)
            if (view_offset[i + 1] < view_offset[i] || view_offset[i + 1] - view_offset[i] > 0x7fffffff)
                throw std::invalid_argument("bad view offsets");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        semidlt_solve(n_views, view_offset, X, Y, u, v, kmtx5, c_T_t, num_radial, bounds_lo5, bounds_hi5, fixed_idx, fixed_val, n_fixed,
                      opts, summary, distortion, view_errors, (cov && opts->compute_covariance) ? cov : nullptr, default_device());
    });
}

namespace {
// argument checks shared by the two sharded semi-DLT entry points; false: fewer than 4 views in total (result left default)
bool semidlt_sharded_args(int32_t n_local, const int64_t* view_offset, const double* X, const double* Y, const double* u, const double* v,
                          int32_t n_total, int32_t first_view, const double* kmtx5, const double* c_T_t, int32_t num_radial,
                          const double* lo, const double* hi, const int32_t* fixed_idx, int32_t n_fixed, const cba_options* opts,
                          cba_summary* summary, int32_t n_ranks, int32_t rank, int32_t device) {
    if (!opts || !summary) throw std::invalid_argument("null argument");
    if (n_ranks < 1 || rank < 0 || rank >= n_ranks) throw std::invalid_argument("bad rank / n_ranks");
    if (n_total < 4) {
        std::memset(summary, 0, sizeof(*summary));
        summary->termination = CBA_TERM_FAILURE;
        std::snprintf(summary->report, sizeof(summary->report), "Insufficient views for calibration (at least 4 required).");
        return false;
    }
    if (n_local < 0 || first_view < 0 || first_view + n_local > n_total) throw std::invalid_argument("view range outside the problem");
    if (!view_offset || !kmtx5 || !c_T_t || (n_local > 0 && (!X || !Y || !u || !v))) throw std::invalid_argument("null argument");
    if (num_radial < 0 || num_radial > 3) throw std::invalid_argument("num_radial must be in [0, 3]");
    if ((lo == nullptr) != (hi == nullptr)) throw std::invalid_argument("bounds need both ends");
    if (n_fixed < 0 || (n_fixed > 0 && !fixed_idx)) throw std::invalid_argument("bad fixed distortion list");
    for (int i = 0; i < n_local; ++i)
        if (view_offset[i + 1] < view_offset[i] || view_offset[i + 1] - view_offset[i] > 0x7fffffff) throw std::invalid_argument("bad view offsets");
    const int ndev = device_count();
    if (ndev <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
    if (device < 0 || device >= ndev) throw std::invalid_argument("device index out of range");
    return true;
}
}  // namespace

cba_status cba_optimize_intrinsics_semidlt_sharded(int32_t n_views_local, const int64_t* view_offset, const double* X, const double* Y,
                                                   const double* u, const double* v, int32_t n_views_total, int32_t first_view,
                                                   double* kmtx5, double* c_T_t, int32_t num_radial, const double* bounds_lo5,
                                                   const double* bounds_hi5, const int32_t* fixed_idx, const double* fixed_val,
                                                   int32_t n_fixed, const cba_options* opts, cba_summary* summary, double* distortion,
                                                   double* view_errors, double* cov, cba_allreduce_fn fn, void* user, int32_t n_ranks,
                                                   int32_t rank, int32_t device) {
    return guarded([&] {
        if (!semidlt_sharded_args(n_views_local, view_offset, X, Y, u, v, n_views_total, first_view, kmtx5, c_T_t, num_radial, bounds_lo5,
                                  bounds_hi5, fixed_idx, n_fixed, opts, summary, n_ranks, rank, device))
            return;
        if (!fn) throw std::invalid_argument("null allreduce callback");
        semidlt_solve_sharded(n_views_local, view_offset, X, Y, u, v, n_views_total, first_view, kmtx5, c_T_t, num_radial, bounds_lo5,
                              bounds_hi5, fixed_idx, fixed_val, n_fixed, opts, summary, distortion, view_errors,
                              (cov && opts->compute_covariance) ? cov : nullptr, device, fn, user, nullptr);
    });
}

cba_status cba_optimize_intrinsics_semidlt_rccl(int32_t n_views_local, const int64_t* view_offset, const double* X, const double* Y,
                                                const double* u, const double* v, int32_t n_views_total, int32_t first_view,
                                                double* kmtx5, double* c_T_t, int32_t num_radial, const double* bounds_lo5,
                                                const double* bounds_hi5, const int32_t* fixed_idx, const double* fixed_val,
                                                int32_t n_fixed, const cba_options* opts, cba_summary* summary, double* distortion,
                                                double* view_errors, double* cov, const uint8_t id[CBA_RCCL_UNIQUE_ID_BYTES],
                                                int32_t n_ranks, int32_t rank, int32_t device) {
    return guarded([&] {
        if (!id) throw std::invalid_argument("null argument");
        if (!semidlt_sharded_args(n_views_local, view_offset, X, Y, u, v, n_views_total, first_view, kmtx5, c_T_t, num_radial, bounds_lo5,
                                  bounds_hi5, fixed_idx, n_fixed, opts, summary, n_ranks, rank, device))
            return;
        CBA_HIP(hipSetDevice(device));
        void* comm = rccl_comm_create(id, n_ranks, rank);
        try {
            semidlt_solve_sharded(n_views_local, view_offset, X, Y, u, v, n_views_total, first_view, kmtx5, c_T_t, num_radial, bounds_lo5,
                                  bounds_hi5, fixed_idx, fixed_val, n_fixed, opts, summary, distortion, view_errors,
                                  (cov && opts->compute_covariance) ? cov : nullptr, device, nullptr, nullptr, comm);
        } catch (...) {
            rccl_comm_destroy(comm, true);  // this rank leaves: its peers' collectives must fail, not hang
            throw;
        }
        rccl_comm_destroy(comm, false);
    });
}

cba_status cba_estimate_homography_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                         const double* u, const double* v, double* h9, int32_t* success) {
    return guarded([&] {
        if (n_views <= 0 || !view_offset || !X || !Y || !u || !v || !h9 || !success) throw std::invalid_argument("null argument");
        for (int i = 0; i < n_views; ++i)
            if (view_offset[i + 1] < view_offset[i] || view_offset[i + 1] - view_offset[i] > 0x7fffffff)
                throw std::invalid_argument("bad view offsets");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        dlt_homography_batch(n_views, view_offset, X, Y, u, v, h9, success, default_device());
    });
}

cba_status cba_estimate_planar_pose_batch(int32_t n_views, const int64_t* view_offset, const double* X, const double* Y,
                                          const double* u, const double* v, const double* kmtx5, double* pose7) {
    return guarded([&] {
        if (n_views <= 0 || !view_offset || !X || !Y || !u || !v || !kmtx5 || !pose7) throw std::invalid_argument("null argument");
        for (int i = 0; i < n_views; ++i)
            if (view_offset[i + 1] < view_offset[i] || view_offset[i + 1] - view_offset[i] > 0x7fffffff)
                throw std::invalid_argument("bad view offsets");
        if (device_count() <= 0) throw NoDevice("no HIP device visible: libcalibba has no CPU fallback");
        planar_seed_batch(n_views, view_offset, X, Y, u, v, kmtx5, pose7, default_device());
    });
}

}  // extern "C"
