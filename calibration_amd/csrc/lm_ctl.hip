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
constexpr int CTL_WAVES = CTL_THREADS / 64;

struct BlockTeam {
    double* red;  // LDS [CTL_WAVES]
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
    // the record goes to page-locked host memory: the payload first, then (system-scope fence) the sequence number the host polls
    __device__ __forceinline__ void publish(const CtlView& V) const {
        for (int k = 1; k < CS_COUNT; ++k) V.rec[k] = V.scal[k];
        __threadfence_system();
        __hip_atomic_store(&V.rec[CS_SEQ], V.scal[CS_SEQ], __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
};

template <bool LDS_A>
__global__ __launch_bounds__(CTL_THREADS) void k_lm_ctl(CtlView V, int mode, int flag) {
    extern __shared__ double lds_dyn[];
    __shared__ double red[CTL_WAVES];
    __shared__ double dk[CTL_NB * CTL_NB];
    BlockTeam tm{red};
    V.Dk = dk;
    if (LDS_A) V.A = lds_dyn;
    ctl_run(tm, V, mode, flag);
}

size_t lm_ctl_lds_bytes(int n) {
    const int lda = (n | 1);
    return static_cast<size_t>(n + 1) * lda * sizeof(double);
}
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
