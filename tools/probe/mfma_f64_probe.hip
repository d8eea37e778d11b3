// tools/probe/mfma_f64_probe.hip — measurement probe (not product code).
//  (1) operand / result lane layout of v_mfma_f64_4x4x4_4b_f64 and v_mfma_f64_16x16x4_f64, found with one-hot operands;
//  (2) how many independent fp64 VALU FMAs issue in the shadow of an fp64 MFMA on one SIMD (1 and 2 waves per SIMD).
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_f64_probe mfma_f64_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double double4_t __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ void k_layout_4x4(int* out /*[64 la][64 lb] -> D lane or -1*/, double* val) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            const double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
            if (d != 0.0) { out[la * 64 + lb] = lane; val[la * 64 + lb] = d; }
        }
}
__global__ void k_layout_16(int* out /*[64][64] -> lane*4+reg or -1*/) {
    const int lane = threadIdx.x;
    for (int la = 0; la < 64; ++la)
        for (int lb = 0; lb < 64; ++lb) {
            const double a = lane == la ? 1.0 : 0.0, b = lane == lb ? 1.0 : 0.0;
            double4_t c = {0, 0, 0, 0};
            c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
            for (int r = 0; r < 4; ++r)
                if (c[r] != 0.0) out[la * 64 + lb] = lane * 4 + r;
        }
}

// MK: 0 none, 1 = 16x16x4, 2 = 4x4x4_4b.  Per loop iteration: M MFMAs (independent accumulators) and V independent fp64 FMAs.
template <int MK, int M, int V>
__global__ void __launch_bounds__(256) k_rate(double* out, long long* cyc, int iters, double seed) {
    double f[V > 0 ? V : 1];
    for (int i = 0; i < V; ++i) f[i] = seed + i + threadIdx.x;
    double4_t c16[M > 0 ? M : 1];
    double c4[M > 0 ? M : 1];
    for (int i = 0; i < M; ++i) { c16[i] = double4_t{0, 0, 0, 0}; c4[i] = 0; }
    const double a = seed * threadIdx.x, b = seed + 1.0;
    const long long t0 = __builtin_readcyclecounter();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {  // 4 repetitions per trip so the loop overhead is small
#pragma unroll
            for (int i = 0; i < M; ++i) {
                if (MK == 1) c16[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c16[i], 0, 0, 0);
                if (MK == 2) c4[i] = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, c4[i], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < V; ++i) f[i] = __builtin_fma(f[i], a, b);
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    double s = 0;
    for (int i = 0; i < V; ++i) s += f[i];
    for (int i = 0; i < M; ++i) s += c16[i][0] + c16[i][1] + c16[i][2] + c16[i][3] + c4[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int MK, int M, int V>
int run_rate(const char* name, int waves_per_simd, double* dout, long long* dcyc) {
    const int iters = 2000, blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = one per SIMD
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    k_rate<MK, M, V><<<blocks, 256>>>(dout, dcyc, 10, 1.0);
    CK(hipEventRecord(e0));
    k_rate<MK, M, V><<<blocks, 256>>>(dout, dcyc, iters, 1.0);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<long long> c(blocks);
    CK(hipMemcpy(c.data(), dcyc, blocks * sizeof(long long), hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto x : c) avg += x;
    avg /= blocks;
    const double per_rep = avg / (iters * 4.0);
    printf("%-10s waves/SIMD %d  M=%d V=%2d : %.3f ms, %8.1f counter ticks per repetition (per wave), %.1f ns per repetition\n", name, waves_per_simd, M, V, ms,
           per_rep, ms * 1e6 / (iters * 4.0));
    return 0;
}

int main() {
    int *d4, *d16;
    double* dv;
    CK(hipMalloc(&d4, 4096 * 4)); CK(hipMalloc(&d16, 4096 * 4)); CK(hipMalloc(&dv, 4096 * 8));
    CK(hipMemset(d4, 0xFF, 4096 * 4)); CK(hipMemset(d16, 0xFF, 4096 * 4));
    k_layout_4x4<<<1, 64>>>(d4, dv);
    k_layout_16<<<1, 64>>>(d16);
    std::vector<int> h4(4096), h16(4096);
    CK(hipMemcpy(h4.data(), d4, 4096 * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(h16.data(), d16, 4096 * 4, hipMemcpyDeviceToHost));
    printf("4x4x4_4b: for each A-lane la, the B-lanes lb that give a product and the D lane it lands in\n");
    for (int la = 0; la < 64; ++la) {
        printf("  la %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h4[la * 64 + lb] >= 0) printf(" (lb %2d -> D %2d)", lb, h4[la * 64 + lb]);
        printf("\n");
    }
    printf("16x16x4: for A-lane la (first 20 shown), B lanes and D lane*4+reg\n");
    for (int la = 0; la < 64; la += 1) {
        if (la >= 20 && la % 16 != 0) continue;
        printf("  la %2d:", la);
        for (int lb = 0; lb < 64; ++lb)
            if (h16[la * 64 + lb] >= 0) printf(" (%d->%d.%d)", lb, h16[la * 64 + lb] / 4, h16[la * 64 + lb] % 4);
        printf("\n");
    }
    double* dout; long long* dcyc;
    CK(hipMalloc(&dout, 1024 * 256 * 8)); CK(hipMalloc(&dcyc, 1024 * 8));
    for (int w = 1; w <= 2; ++w) {
        run_rate<0, 0, 16>("valu", w, dout, dcyc);
        run_rate<1, 1, 0>("m16", w, dout, dcyc);
        run_rate<1, 1, 4>("m16", w, dout, dcyc);
        run_rate<1, 1, 8>("m16", w, dout, dcyc);
        run_rate<1, 1, 12>("m16", w, dout, dcyc);
        run_rate<1, 1, 16>("m16", w, dout, dcyc);
        run_rate<1, 2, 24>("m16", w, dout, dcyc);
        run_rate<2, 1, 0>("m4", w, dout, dcyc);
        run_rate<2, 4, 0>("m4", w, dout, dcyc);
        run_rate<2, 4, 4>("m4", w, dout, dcyc);
        run_rate<2, 4, 8>("m4", w, dout, dcyc);
        run_rate<2, 4, 12>("m4", w, dout, dcyc);
        run_rate<2, 4, 16>("m4", w, dout, dcyc);
        run_rate<2, 3, 12>("m4", w, dout, dcyc);
    }
    return 0;
}
