// chain_probe.hip — latencies that bound a single-workgroup kernel (the LM controller, lm_ctl.hip): dependent fp64 FMAs,
// v_rsq_f64, the library sqrt + division, an LDS write -> barrier -> read round trip, a bare workgroup barrier; per operation in
// shader cycles (s_memtime) and ns (s_memrealtime, 100 MHz), for a workgroup of 64 .. 1024 threads.
// build: hipcc -O3 --offload-arch=gfx950 tools/probe/chain_probe.hip -o tools/probe/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ unsigned long long stamp() {
    unsigned long long t;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
    return t;
}

template <int WHAT>
__global__ void k_probe(int n, double seed, double* out, unsigned long long* ticks) {
    __shared__ double sh[1024];
    double x = seed + threadIdx.x * 1e-9, y = 1.0000001;
    sh[threadIdx.x] = x;
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = stamp();
    for (int i = 0; i < n; ++i) {
        if (WHAT == 0) x = __builtin_fma(x, y, 1e-9);                        // dependent FMA
        if (WHAT == 1) x = __builtin_amdgcn_rsq(x) + 1.5;                    // v_rsq_f64 + add
        if (WHAT == 2) x = 1.0 / sqrt(x) + 1.5;                               // library sqrt + division + add
        if (WHAT == 3) { sh[threadIdx.x] = x; __syncthreads(); x = sh[(threadIdx.x + 1) % blockDim.x] + 1e-9; __syncthreads(); }
        if (WHAT == 4) { __syncthreads(); }
        if (WHAT == 5) { x = sh[(static_cast<int>(x) + threadIdx.x) & 1023] + 1.0; }  // dependent LDS read (address from the value)
        if (WHAT == 6) { x = __builtin_fma(x, y, 1e-9); y = __builtin_fma(y, 1.0000001, 1e-12); }  // two independent chains
    }
    const unsigned long long t1 = stamp();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    out[threadIdx.x] = x + y;
    if (threadIdx.x == 0) { ticks[0] = t1 - t0; ticks[1] = r1 - r0; }
}

int main() {
    double* out;
    unsigned long long* ticks;
    CHECK(hipMalloc(&out, 1024 * sizeof(double)));
    CHECK(hipMalloc(&ticks, 2 * sizeof(unsigned long long)));
    const char* names[] = {"dependent fp64 fma", "v_rsq_f64 + add", "1/sqrt (library) + add", "LDS write-barrier-read-barrier", "barrier",
                           "dependent LDS read", "two independent fma chains (per pair)"};
    const int n = 2000;
    for (int threads : {64, 128, 256, 512, 1024}) {
        for (int what = 0; what < 7; ++what) {
            unsigned long long h[2] = {0, 0};
            for (int rep = 0; rep < 3; ++rep) {
                switch (what) {
                    case 0: hipLaunchKernelGGL(k_probe<0>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    case 1: hipLaunchKernelGGL(k_probe<1>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    case 2: hipLaunchKernelGGL(k_probe<2>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    case 3: hipLaunchKernelGGL(k_probe<3>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    case 4: hipLaunchKernelGGL(k_probe<4>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    case 5: hipLaunchKernelGGL(k_probe<5>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                    default: hipLaunchKernelGGL(k_probe<6>, dim3(1), dim3(threads), 0, 0, n, 1.0, out, ticks); break;
                }
                CHECK(hipDeviceSynchronize());
                CHECK(hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost));
            }
            std::printf("threads %4d  %-40s %8.1f cycles  %8.1f ns per iteration (clock %.2f GHz)\n", threads, names[what],
                        double(h[0]) / n, double(h[1]) * 10.0 / n, double(h[0]) / (double(h[1]) * 10.0));
        }
    }
    return 0;
}
