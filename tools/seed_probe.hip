// seed_probe.hip -- accuracy of the gfx950 FP64 reciprocal / rsqrt seeds, and FMA issue rate of
// half-populated waves (32-thread blocks).  Sizing experiment, not part of the product.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <vector>

__global__ void seeds(const double *x, double *rcp, double *rsq, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) { rcp[i] = __builtin_amdgcn_rcp(x[i]); rsq[i] = __builtin_amdgcn_rsq(x[i]); }
}

template <int CHAINS>
__global__ void fma_chains(double *out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) x[i] = __builtin_fma(x[i], a, b);
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    const int n = 1 << 20;
    std::vector<double> hx(n), hr(n), hs(n);
    for (int i = 0; i < n; ++i) hx[i] = std::exp(-20.0 + 40.0 * (i + 0.5) / n);
    double *x, *r, *s;
    (void)hipMalloc(&x, n * 8); (void)hipMalloc(&r, n * 8); (void)hipMalloc(&s, n * 8);
    (void)hipMemcpy(x, hx.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(seeds, dim3(n / 256), dim3(256), 0, 0, x, r, s, n);
    (void)hipMemcpy(hr.data(), r, n * 8, hipMemcpyDeviceToHost);
    (void)hipMemcpy(hs.data(), s, n * 8, hipMemcpyDeviceToHost);
    double er = 0, es = 0;
    for (int i = 0; i < n; ++i) {
        er = std::fmax(er, std::fabs(hr[i] * hx[i] - 1.0));
        es = std::fmax(es, std::fabs(hs[i] * hs[i] * hx[i] - 1.0) * 0.5);
    }
    printf("v_rcp_f64 max rel err %.3e (%.1f bits)\nv_rsq_f64 max rel err %.3e (%.1f bits)\n", er, -std::log2(er), es, -std::log2(es));

    double *out; (void)hipMalloc(&out, sizeof(double) * 256 * 4 * 8 * 64 * 2);
    const int iters = 20000;
    for (int threads : {64, 32}) {
        for (int wps : {1, 2}) {
            const int blocks = 256 * 4 * wps;
            hipEvent_t e0, e1;
            (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
            hipLaunchKernelGGL(fma_chains<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999999, 1e-7);
            (void)hipDeviceSynchronize();
            (void)hipEventRecord(e0);
            hipLaunchKernelGGL(fma_chains<8>, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999999, 1e-7);
            (void)hipEventRecord(e1);
            (void)hipEventSynchronize(e1);
            float ms; (void)hipEventElapsedTime(&ms, e0, e1);
            printf("fma x8 chains, %d threads/block, %d waves/SIMD: %.3f ms -> %.4f wave-inst/clk/SIMD\n", threads, wps, ms,
                   (double)iters * 64 * wps / (ms * 1e-3 * 2.4e9));
        }
    }
    return 0;
}
