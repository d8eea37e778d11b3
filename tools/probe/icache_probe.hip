// icache_probe.hip — what does COLD CODE cost a single-workgroup kernel?  A kernel of N KB of straight-line code (dependent fp64
// FMAs, executed once) is timed from inside (s_memrealtime) (a) right after an L2-thrashing stream kernel, (b) back to back with
// itself, (c) on a stream whose CU mask nothing else uses, with the thrasher confined to the other CUs.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/icache_probe.hip -o tools/probe/icache_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int N>
__global__ void k_straight(double seed, double* out, unsigned long long* ticks) {
    double x = seed + threadIdx.x * 1e-9;
    const double y = 1.0000001, z = 1e-9;
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
#pragma unroll
    for (int i = 0; i < N; ++i) x = __builtin_fma(x, y, z);  // N x 8 bytes of code, one dependent chain
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) ticks[0] = r1 - r0;
}

__global__ void k_thrash(const double* __restrict__ in, double* __restrict__ out, size_t n) {
    size_t i = static_cast<size_t>(blockIdx.x) * blockDim.x + threadIdx.x;
    const size_t stride = static_cast<size_t>(gridDim.x) * blockDim.x;
    double s = 0.0;
    for (; i < n; i += stride) s += in[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int N>
static int run(const char* label, hipStream_t sk, hipStream_t st, const double* big, double* sink, size_t nbig, double* out, unsigned long long* ticks, int thrash) {
    double us[6];
    for (int rep = 0; rep < 6; ++rep) {
        if (thrash) { hipLaunchKernelGGL(k_thrash, dim3(2048), dim3(256), 0, st, big, sink, nbig); CHECK(hipStreamSynchronize(st)); }
        hipLaunchKernelGGL(k_straight<N>, dim3(1), dim3(64), 0, sk, 1.0, out, ticks);
        CHECK(hipStreamSynchronize(sk));
        unsigned long long h = 0;
        CHECK(hipMemcpy(&h, ticks, sizeof(h), hipMemcpyDeviceToHost));
        us[rep] = double(h) * 0.01;
    }
    std::printf("%-48s code %3d KB: %7.1f %7.1f %7.1f %7.1f %7.1f %7.1f us   (chain alone: %.1f us at 8 cycles/FMA, 2.4 GHz)\n", label, N * 8 / 1024,
                us[0], us[1], us[2], us[3], us[4], us[5], N * 8.0 / 2400.0);
    return 0;
}

int main() {
    const size_t nbig = size_t(1) << 27;  // 1 GiB of doubles
    double *big, *sink, *out;
    unsigned long long* ticks;
    CHECK(hipMalloc(&big, nbig * sizeof(double)));
    CHECK(hipMemset(big, 0, nbig * sizeof(double)));
    CHECK(hipMalloc(&sink, 2048 * 256 * sizeof(double)));
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    CHECK(hipMalloc(&ticks, sizeof(unsigned long long)));
    hipStream_t s0, s1, sm_ctl, sm_rest;
    CHECK(hipStreamCreate(&s0));
    CHECK(hipStreamCreate(&s1));
    // CU masks: the controller stream gets CUs 0, 1 of XCD 0 (bits 0, 1); the other stream everything else
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    const int ncu = prop.multiProcessorCount;
    std::vector<uint32_t> m_ctl((ncu + 31) / 32, 0u), m_rest((ncu + 31) / 32, 0xFFFFFFFFu);
    m_ctl[0] = 0x3u;
    m_rest[0] &= ~0x3u;
    bool masked = hipExtStreamCreateWithCUMask(&sm_ctl, static_cast<uint32_t>(m_ctl.size()), m_ctl.data()) == hipSuccess &&
                  hipExtStreamCreateWithCUMask(&sm_rest, static_cast<uint32_t>(m_rest.size()), m_rest.data()) == hipSuccess;
    std::printf("CUs %d, CU-masked streams %s\n", ncu, masked ? "ok" : "not available");
    if (run<512>("back to back, no thrashing", s0, s1, big, sink, nbig, out, ticks, 0)) return 1;
    if (run<2048>("back to back, no thrashing", s0, s1, big, sink, nbig, out, ticks, 0)) return 1;
    if (run<4096>("back to back, no thrashing", s0, s1, big, sink, nbig, out, ticks, 0)) return 1;
    if (run<512>("after a 1 GiB stream kernel", s0, s1, big, sink, nbig, out, ticks, 1)) return 1;
    if (run<2048>("after a 1 GiB stream kernel", s0, s1, big, sink, nbig, out, ticks, 1)) return 1;
    if (run<4096>("after a 1 GiB stream kernel", s0, s1, big, sink, nbig, out, ticks, 1)) return 1;
    if (masked) {
        if (run<2048>("own CU pair, stream kernel on the other CUs", sm_ctl, sm_rest, big, sink, nbig, out, ticks, 1)) return 1;
        if (run<4096>("own CU pair, stream kernel on the other CUs", sm_ctl, sm_rest, big, sink, nbig, out, ticks, 1)) return 1;
        if (run<4096>("own CU pair, no thrashing", sm_ctl, sm_rest, big, sink, nbig, out, ticks, 0)) return 1;
    }
    return 0;
}
