// ctl_probe.hip — the reduced solve of the LM controller (lm_ctl.hpp: ctl_cholesky + ctl_backsolve) alone, one workgroup, the
// matrix in LDS: time per repetition inside ONE launch (the first repetition runs cold code, the later ones warm).
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -mllvm -pragma-unroll-threshold=1000000 -mllvm -unroll-threshold=1000000 tools/probe/ctl_probe.hip -o tools/probe/ctl_probe
#include "../../calibration_amd/csrc/lm_ctl.hip"
#include <cstdio>
#include <vector>

using namespace cba;

// BlockTeam with stamps: thread 0's shader clock at the phase boundaries of ctl_cholesky, summed per boundary over the panels
struct StampTeam : BlockTeam {
    unsigned long long* acc;   // LDS [8]: cycles spent before reaching mark k (since the previous mark)
    unsigned long long* prev;
    __device__ __forceinline__ void mark(int k) const {
        if (threadIdx.x == 0) {
            unsigned long long t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
            acc[k] += t - *prev;
            *prev = t;
        }
    }
};

__global__ __launch_bounds__(CTL_THREADS) void k_probe(int m, int lda, const double* A0, double* xout, double* times, int reps) {
    extern __shared__ double A[];
    __shared__ double red[CTL_WAVES];
    __shared__ double Ld[CTL_LDS_MAX_N * CTL_NB];
    __shared__ double small[2 * 256];
    __shared__ int okf;
    __shared__ unsigned long long last;
    __shared__ double scal[CS_COUNT];
    __shared__ unsigned long long acc[8], prev;
    StampTeam tm{{red, scal, &last}, acc, &prev};
    if (threadIdx.x < 8) acc[threadIdx.x] = 0;
    if (threadIdx.x == 0) prev = 0;
    for (int rep = 0; rep < reps; ++rep) {
        for (int e = threadIdx.x; e < (m + 1) * lda; e += CTL_THREADS) A[e] = A0[e];
        __syncthreads();
        const unsigned long long t0 = wall_clock64();
        const bool ok = ctl_cholesky(tm, A, lda, m, Ld, small, &okf);
        const unsigned long long t1 = wall_clock64();
        ctl_backsolve(tm, A, lda, m, Ld, small, small + 256);
        const unsigned long long t2 = wall_clock64();
        if (threadIdx.x == 0) { times[3 * rep] = (t1 - t0) * 0.01; times[3 * rep + 1] = (t2 - t1) * 0.01; times[3 * rep + 2] = ok ? 1.0 : 0.0; }
        if (threadIdx.x == 0 && rep == reps - 1) for (int k = 0; k < 7; ++k) times[3 * reps + k] = static_cast<double>(acc[k]);
        if (threadIdx.x == 0 && rep == reps - 2) for (int k = 0; k < 7; ++k) acc[k] = 0;
        __syncthreads();
    }
    for (int r = threadIdx.x; r < m; r += CTL_THREADS) xout[r] = small[256 + r];
}

int main() {
    for (int m : {8, 16, 24, 72, 120, 128}) {  // (multiples of the panel width: the caller pads)
        const int lda = m | 1;
        std::vector<double> B(m * m), A((m + 1) * lda, 0.0), b(m);
        unsigned s = 12345u;
        auto rnd = [&] { s = s * 1664525u + 1013904223u; return (s >> 8) * (1.0 / 16777216.0) - 0.5; };
        for (double& v : B) v = rnd();
        for (int i = 0; i < m; ++i) b[i] = rnd();
        for (int i = 0; i < m; ++i)
            for (int j = 0; j <= i; ++j) {
                double t = 0;
                for (int k = 0; k < m; ++k) t += B[i * m + k] * B[j * m + k];
                A[i * lda + j] = t + (i == j ? m : 0.0);
            }
        for (int j = 0; j < m; ++j) A[m * lda + j] = b[j];
        double *dA, *dx, *dt;
        const int reps = 6;
        hipMalloc(&dA, A.size() * sizeof(double));
        hipMalloc(&dx, m * sizeof(double));
        hipMalloc(&dt, (3 * reps + 8) * sizeof(double));
        hipMemcpy(dA, A.data(), A.size() * sizeof(double), hipMemcpyHostToDevice);
        hipFuncSetAttribute(reinterpret_cast<const void*>(k_probe), hipFuncAttributeMaxDynamicSharedMemorySize, 140000);
        hipLaunchKernelGGL(k_probe, dim3(1), dim3(CTL_THREADS), (m + 1) * lda * sizeof(double), 0, m, lda, dA, dx, dt, reps);
        if (hipDeviceSynchronize() != hipSuccess) { std::printf("launch failed\n"); return 1; }
        std::vector<double> x(m), t(3 * reps + 8);
        hipMemcpy(x.data(), dx, m * sizeof(double), hipMemcpyDeviceToHost);
        hipMemcpy(t.data(), dt, (3 * reps + 8) * sizeof(double), hipMemcpyDeviceToHost);
        // residual |A x - b|
        double err = 0, nrm = 0;
        for (int i = 0; i < m; ++i) {
            double r = -b[i];
            for (int j = 0; j < m; ++j) r += (j <= i ? A[i * lda + j] : A[j * lda + i]) * x[j];
            err = std::fmax(err, std::fabs(r));
            nrm = std::fmax(nrm, std::fabs(b[i]));
        }
        std::printf("m %3d  residual %.2e  factorise us:", m, err / nrm);
        for (int r = 0; r < reps; ++r) std::printf(" %.1f", t[3 * r]);
        std::printf("   back-substitute us:");
        for (int r = 0; r < reps; ++r) std::printf(" %.1f", t[3 * r + 1]);
        std::printf("  ok %d\n", int(t[2]));
        const int np = (m + CTL_NB - 1) / CTL_NB;
        std::printf("        thread 0, cycles per panel (last repetition, %d panels): loop top %.0f | diagonal block loads %.0f | factor %.0f | panel row %.0f | barrier %.0f | "
                    "trailing update %.0f | barrier %.0f\n", np, t[3 * reps + 0] / np, t[3 * reps + 1] / np, t[3 * reps + 2] / np, t[3 * reps + 3] / np,
                    t[3 * reps + 4] / np, t[3 * reps + 5] / np, t[3 * reps + 6] / np);
    }
    return 0;
}
