// lm_ctl.hip — the LM controller (lm_ctl.hpp) as ONE workgroup on the engine's stream: k_lm_ctl runs right behind the packed
// exchange of an LM step (k_pack -> ncclAllReduce -> k_lm_ctl), takes the step decision, adopts the next system, factorises the
// reduced system and leaves the next trial point of the shared blocks in device memory.  This is what the reference delegates to
// ceres::Solve's trust-region loop and its dense / sparse normal-equation solver (src/estimation/detail/ceresutils.h:27-43).
//
// The reduced matrix (effective columns + the right-hand side as one more row) lives in LDS while it fits (n <= CTL_LDS_MAX_N:
// 128 x 129 doubles = 132 KB of the CU's 160 KB; the 8-camera rig of BASELINE config 3 is 128 wide), in global memory (L2) above.
#include <mutex>
#include <set>

#include "engine.hpp"
#include "lm_ctl.hpp"
#include "lm_state.hpp"
#include "wave_reduce.hpp"

namespace cba {

constexpr int CTL_THREADS = 512;
constexpr int CTL_SMALL_N = 136;  // up to this reduced size the short per-column arrays live in LDS
constexpr int CTL_WAVES = CTL_THREADS / 64;

struct BlockTeam {
    double* red;      // LDS [CTL_WAVES]
    double* g_scal;   // the control scalars in device memory (the kernel works on an LDS copy)
    unsigned long long* last;  // LDS: thread 0's clock at the previous tick
    __device__ __forceinline__ int tid() const { return threadIdx.x; }
    __device__ __forceinline__ int size() const { return CTL_THREADS; }
    __device__ __forceinline__ void sync() const { __syncthreads(); }
    // fixed-order totals, the same value in every thread
    __device__ __forceinline__ double sum(double v) const {
        const double t = wave_sum63(v);
        if ((threadIdx.x & 63) == 63) red[threadIdx.x >> 6] = t;
        __syncthreads();
        double s = 0.0;
#pragma unroll
        for (int w = 0; w < CTL_WAVES; ++w) s += red[w];
        __syncthreads();
        return s;
    }
    __device__ __forceinline__ double max(double v) const {
        for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
        if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
        __syncthreads();
        double m = red[0];
#pragma unroll
        for (int w = 1; w < CTL_WAVES; ++w) m = fmax(m, red[w]);
        __syncthreads();
        return m;
    }
    // exclusive prefix count of a flag over the team in thread order (ballot inside the wavefront, the wavefronts' totals through LDS)
    __device__ __forceinline__ int count_before(bool flag, int* total) const {
        const unsigned long long mask = __ballot(flag);
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const int before = __popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) red[wave] = static_cast<double>(__popcll(mask));
        __syncthreads();
        int base = 0, tot = 0;
#pragma unroll
        for (int w = 0; w < CTL_WAVES; ++w) {
            const int cw = static_cast<int>(red[w]);
            base += w < wave ? cw : 0;
            tot += cw;
        }
        __syncthreads();
        *total = tot;
        return base + before;
    }
    __device__ __forceinline__ void mark(int) const {}
    // time since the previous tick -> the phase's slot (thread 0; 100 MHz constant clock)
    __device__ __forceinline__ void tick(const CtlView& V, int slot) const {
        if (threadIdx.x == 0) {
            const unsigned long long t = wall_clock64();
            V.scal[CS_PROF + slot] += static_cast<double>(t - *last);
            *last = t;
        }
    }
    // The control scalars go back to device memory and, as the control record, to page-locked host memory: the payload first,
    // then (system-scope fence) the sequence number the host polls.
    __device__ __forceinline__ void publish(const CtlView& V) const {
        tick(V, CP_PUBLISH);
        __syncthreads();
        const int k = threadIdx.x;
        if (k < CS_COUNT) {
            const double v = V.scal[k];
            g_scal[k] = v;
            if (k != CS_SEQ) V.rec[k] = v;
        }
        __threadfence_system();
        __syncthreads();
        if (k == 0) __hip_atomic_store(&V.rec[CS_SEQ], V.scal[CS_SEQ], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

template <bool LDS_A>
__global__ __launch_bounds__(CTL_THREADS) void k_lm_ctl(CtlView V, int mode, int flag) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[CTL_WAVES];
    __shared__ double s_Ld[LDS_A ? CTL_LDS_MAX_N * CTL_NB : 1];  // the factors of the diagonal blocks
    __shared__ double s_scal[CS_COUNT];
    // the short per-column arrays are read in dependent chains / by one thread: LDS copies (reciprocal pivots, compact solution,
    // g_c, Jacobi scale, diag(H_cc), effective-column flags and list, masks, column tables), the persistent ones written back
    // at the end
    __shared__ double s_small[5 * CTL_SMALL_N];
    __shared__ int s_idx[3 * CTL_SMALL_N];
    __shared__ int8_t s_flags[3 * CTL_SMALL_N];
    __shared__ int s_ok;
    __shared__ unsigned long long s_last;
    const int tid = threadIdx.x, n = V.n;
    const bool small = LDS_A || n <= CTL_SMALL_N;  // (LDS_A: n <= CTL_LDS_MAX_N <= CTL_SMALL_N, known at compile time: every pointer below
                                                   // is then an LDS pointer on every path and the compiler addresses it as one)
    static_assert(CTL_LDS_MAX_N <= CTL_SMALL_N, "the LDS form keeps the short arrays in LDS too");
    double *g_gc = V.gc, *g_scale2 = V.scale2, *g_hdiag = V.hdiag;
    int* g_idx = V.idx;
    int8_t* g_eff = V.eff;
    // the control scalars are read dozens of times between barriers: work on an LDS copy, written back by publish()
    if (tid < CS_COUNT) s_scal[tid] = V.scal[tid];
    if (tid == 0) s_last = wall_clock64();
    if (small) {
        for (int i = tid; i < n; i += CTL_THREADS) {
            s_small[2 * CTL_SMALL_N + i] = g_gc[i];
            s_small[3 * CTL_SMALL_N + i] = g_scale2[i];
            s_small[4 * CTL_SMALL_N + i] = g_hdiag[i];
            s_idx[i] = g_idx[i];
            s_idx[CTL_SMALL_N + i] = V.colcam[i];
            s_idx[2 * CTL_SMALL_N + i] = V.collc[i];
            s_flags[i] = g_eff[i];
            s_flags[CTL_SMALL_N + i] = V.active[i];
        }
        for (int c = tid; c < V.n_cams && c < CTL_SMALL_N; c += CTL_THREADS) s_flags[2 * CTL_SMALL_N + c] = V.cam_var[c];
        V.rdiag = s_small; V.xs = s_small + CTL_SMALL_N; V.gc = s_small + 2 * CTL_SMALL_N; V.scale2 = s_small + 3 * CTL_SMALL_N;
        V.hdiag = s_small + 4 * CTL_SMALL_N;
        V.idx = s_idx; V.colcam = s_idx + CTL_SMALL_N; V.collc = s_idx + 2 * CTL_SMALL_N;
        V.eff = s_flags; V.active = s_flags + CTL_SMALL_N;
        if (V.n_cams <= CTL_SMALL_N) V.cam_var = s_flags + 2 * CTL_SMALL_N;
    }
    BlockTeam tm{red, V.scal, &s_last};
    V.scal = s_scal;
    V.okflag = &s_ok;
    if (LDS_A) { V.A = lds_dyn; V.Ld = s_Ld; }
    __syncthreads();
    ctl_run(tm, V, mode, flag);
    if (small)
        for (int i = tid; i < n; i += CTL_THREADS) {
            g_gc[i] = s_small[2 * CTL_SMALL_N + i];
            g_scale2[i] = s_small[3 * CTL_SMALL_N + i];
            g_hdiag[i] = s_small[4 * CTL_SMALL_N + i];
            g_idx[i] = s_idx[i];
            g_eff[i] = s_flags[i];
        }
}

size_t lm_ctl_lds_bytes(int n) { return static_cast<size_t>(ctl_padded(n) + 1) * ctl_lda(n) * sizeof(double); }
bool lm_ctl_fits_lds(int n) { return n <= CTL_LDS_MAX_N; }

void launch_lm_ctl(const CtlView& V, int mode, int flag, hipStream_t stream) {
    if (lm_ctl_fits_lds(V.n))
        hipLaunchKernelGGL(k_lm_ctl<true>, dim3(1), dim3(CTL_THREADS), lm_ctl_lds_bytes(V.n), stream, V, mode, flag);
    else
        hipLaunchKernelGGL(k_lm_ctl<false>, dim3(1), dim3(CTL_THREADS), 0, stream, V, mode, flag);
    CBA_HIP(hipGetLastError());
}

// once per device (init_lm_state): more than 64 KB of dynamic LDS needs the attribute; touching the kernel also loads this
// translation unit's code object ahead of the first solve
void warm_lm_ctl() {
    static std::mutex mu;
    static std::set<int> done;
    int dev = 0;
    CBA_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(mu);
    if (!done.insert(dev).second) return;
    CBA_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(k_lm_ctl<true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                static_cast<int>(lm_ctl_lds_bytes(CTL_LDS_MAX_N))));
    hipFuncAttributes fa;
    (void)hipFuncGetAttributes(&fa, reinterpret_cast<const void*>(k_lm_ctl<false>));
}

}  // namespace cba
