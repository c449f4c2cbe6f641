// How does the fp32 MFMA round?  And how accurate is v_rsq_f32?  (round 4: the factor's error against the vendor's
// fp32 potrf grows with the block column -- scripts/acc_diag.py -- which points at a biased rounding somewhere.)
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma_round scripts/ubench/mfma_round.hip && /tmp/mfma_round
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstring>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
// every lane supplies the same a, b per k step -> every output element = c + a0 b0 + a1 b1
__global__ void k32(const float* in, float* out) {
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = in[0];
    // lanes 0..31 supply k = 0, lanes 32..63 k = 1
    float a = threadIdx.x < 32 ? in[1] : in[3], b = threadIdx.x < 32 ? in[2] : in[4];
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void k16(const float* in, float* out) {        // 16x16x4: lanes g = lane >> 4 supply k = g
    f32x4 acc;
    for (int q = 0; q < 4; ++q) acc[q] = in[0];
    int g = threadIdx.x >> 4;
    float a = in[1 + 2 * g], b = in[2 + 2 * g];
    acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void kfma(const float* in, float* out) { out[0] = __builtin_fmaf(in[3], in[4], __builtin_fmaf(in[1], in[2], in[0])); }
__global__ void krsq(const float* d, float* r, float* s, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { float q = __builtin_amdgcn_rsqf(d[i]); r[i] = q; s[i] = d[i] * q; }
}
// a long accumulation chain: sum_{k} a_k b_k with c = -S in the accumulator (the Schur-complement shape), K steps
__global__ void kchain(const float* a, const float* b, float c0, int K, float* out) {
    f32x16 acc;
    for (int q = 0; q < 16; ++q) acc[q] = c0;
    for (int k = 0; k < K; k += 2) {
        float av = threadIdx.x < 32 ? a[k] : a[k + 1], bv = threadIdx.x < 32 ? b[k] : b[k + 1];
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
    }
    if (threadIdx.x == 0) out[0] = acc[0];
}
__global__ void kchain_fma(const float* a, const float* b, float c0, int K, float* out) {
    float acc = c0;
    for (int k = 0; k < K; ++k) acc = __builtin_fmaf(a[k], b[k], acc);
    out[0] = acc;
}
static float run(void (*kern)(const float*, float*), const float* h, int n) {
    float *di, *dout, o;
    hipMalloc(&di, 64); hipMalloc(&dout, 4);
    hipMemcpy(di, h, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(kern, dim3(1), dim3(64), 0, 0, di, dout);
    hipMemcpy(&o, dout, 4, hipMemcpyDeviceToHost);
    hipFree(di); hipFree(dout);
    return o;
}
int main() {
    const float u = ldexpf(1.f, -23);       // ulp(1)
    struct { const char* nm; float v[9]; } cases[] = {
        {"c=1, +0.75 ulp, +0        (RNE: 1+ulp, RTZ: 1)", {1.f, 0.75f, u, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=1, +0.3 ulp, +0.3 ulp   (products summed first: 1+ulp; one rounding per step: 1)", {1.f, 0.3f, u, 0.3f, u, 0, 0, 0, 0}},
        {"c=-1, -0.75 ulp, 0        (RNE: -1-ulp)", {-1.f, -0.75f, u, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=1, -0.25 ulp(of 1), 0   (below 1 the ulp halves: RNE 1-0.25u exact; )", {1.f, -0.25f, u, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=1, +1.5 ulp, 0          (tie: RNE to even = 1+2ulp)", {1.f, 1.5f, u, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=1, +0.5 ulp, 0          (tie: RNE to even = 1)", {1.f, 0.5f, u, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=1, +0.5 ulp, +tiny      (sticky: fused 1+ulp)", {1.f, 0.5f, u, 1e-3f, u, 0, 0, 0, 0}},
        {"c=2^-140 denormal + 0     (denormal kept?)", {ldexpf(1.f, -140), 0.f, 0.f, 0.f, 0.f, 0, 0, 0, 0}},
        {"c=0, 2^-70 * 2^-70        (denormal product kept?)", {0.f, ldexpf(1.f, -70), ldexpf(1.f, -70), 0.f, 0.f, 0, 0, 0, 0}},
    };
    for (auto& c : cases) {
        float m32 = run(k32, c.v, 9), m16 = run(k16, c.v, 9), f = run(kfma, c.v, 9);
        printf("%-86s mfma32x32x2 %.9g (%+.2f ulp)  mfma16x16x4 %.9g  v_fma chain %.9g\n", c.nm, m32, (m32 - c.v[0]) / u, m16, f);
    }
    // rsq
    const int n = 1 << 20;
    float *hd = (float*)malloc(n * 4), *hr = (float*)malloc(n * 4), *hs = (float*)malloc(n * 4), *dd, *dr, *ds;
    srand(1);
    for (int i = 0; i < n; ++i) hd[i] = expf(-12.f * rand() / RAND_MAX);
    hipMalloc(&dd, n * 4); hipMalloc(&dr, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(dd, hd, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(krsq, dim3(n / 256), dim3(256), 0, 0, dd, dr, ds, n);
    hipMemcpy(hr, dr, n * 4, hipMemcpyDeviceToHost); hipMemcpy(hs, ds, n * 4, hipMemcpyDeviceToHost);
    double b1 = 0, m1 = 0, b2 = 0, m2 = 0, b3 = 0, m3 = 0;
    for (int i = 0; i < n; ++i) {
        double ex = 1.0 / sqrt((double)hd[i]), e1 = (hr[i] - ex) / ex, e2 = (hs[i] - sqrt((double)hd[i])) / sqrt((double)hd[i]);
        double e3 = ((double)sqrtf(hd[i]) - sqrt((double)hd[i])) / sqrt((double)hd[i]);
        b1 += e1; b2 += e2; b3 += e3; m1 = fmax(m1, fabs(e1)); m2 = fmax(m2, fabs(e2)); m3 = fmax(m3, fabs(e3));
    }
    printf("v_rsq_f32: mean rel err %.3e  max %.3e | d*rsq(d) vs sqrt: mean %.3e max %.3e | correctly rounded sqrtf: mean %.3e max %.3e (eps = 5.96e-8)\n",
           b1 / n, m1, b2 / n, m2, b3 / n, m3);
    // long chain: c0 = -sum (rounded) then + a_k b_k: exact answer is the rounding residue
    for (int K : {128, 1024, 4096}) {
        float *ha = (float*)malloc(K * 4), *hb = (float*)malloc(K * 4), *da, *db, *dout;
        double S = 0; double worst_m = 0, worst_f = 0, bias_m = 0, bias_f = 0; int trials = 200;
        hipMalloc(&da, K * 4); hipMalloc(&db, K * 4); hipMalloc(&dout, 4);
        for (int t = 0; t < trials; ++t) {
            S = 0;
            for (int k = 0; k < K; ++k) { ha[k] = 0.5f + 0.5f * rand() / RAND_MAX; hb[k] = 0.5f + 0.5f * rand() / RAND_MAX; S += (double)ha[k] * hb[k]; }
            float c0 = -(float)(S * (1 - 1e-3));
            double exact = (double)c0 + S;
            hipMemcpy(da, ha, K * 4, hipMemcpyHostToDevice); hipMemcpy(db, hb, K * 4, hipMemcpyHostToDevice);
            float om, of;
            hipLaunchKernelGGL(kchain, dim3(1), dim3(64), 0, 0, da, db, c0, K, dout); hipMemcpy(&om, dout, 4, hipMemcpyDeviceToHost);
            hipLaunchKernelGGL(kchain_fma, dim3(1), dim3(1), 0, 0, da, db, c0, K, dout); hipMemcpy(&of, dout, 4, hipMemcpyDeviceToHost);
            double em = (om - exact) / S, ef = (of - exact) / S;
            bias_m += em; bias_f += ef; worst_m = fmax(worst_m, fabs(em)); worst_f = fmax(worst_f, fabs(ef));
        }
        printf("chain K=%4d: (result - exact)/S   mfma mean %+.2e max %.2e | v_fma mean %+.2e max %.2e  (eps 6e-8)\n", K, bias_m / trials, worst_m, bias_f / trials, worst_f);
    }
    return 0;
}
