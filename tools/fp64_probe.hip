// fp64_probe.hip -- micro-benchmark: FP64 VALU issue rate on gfx950 as a function of waves per SIMD
// and of instruction-level parallelism (number of independent FMA chains per lane).
// Used to size the DLS kernels (one IK problem per lane, occupancy 1-2 waves/SIMD).  Not part of the product.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

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

template <int CHAINS>
__global__ void mul_add_chains(double *out, int iters, double a, double b) {
    double x[CHAINS];
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) x[i] = threadIdx.x * 1e-3 + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < CHAINS; ++i) { x[i] = x[i] * a; x[i] = x[i] + b; }
    }
    double s = 0;
#pragma unroll
    for (int i = 0; i < CHAINS; ++i) s += x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <class K>
double time_kernel(K k, int blocks, int threads, double *out, int iters) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999999, 1e-7);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(threads), 0, 0, out, iters, 0.999999, 1e-7);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    double *out; hipMalloc(&out, sizeof(double) * 256 * 4 * 8 * 64 * 4);
    const int iters = 20000;
    printf("kernel,chains,waves_per_simd,ms,inst_per_clk_per_simd(2.4GHz),TFLOPs\n");
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * 4 * wps;  // 64-thread blocks: one wave each
#define RUN(C)                                                                                        \
        {                                                                                             \
            double ms = time_kernel(fma_chains<C>, blocks, 64, out, iters);                           \
            double inst = (double)iters * 8 * C;                                                      \
            double clk = ms * 1e-3 * 2.4e9;                                                           \
            printf("fma,%d,%d,%.3f,%.4f,%.2f\n", C, wps, ms, inst * wps / clk,                        \
                   inst * 2 * 64.0 * blocks / (ms * 1e-3) / 1e12);                                    \
        }
        RUN(1) RUN(2) RUN(4) RUN(8) RUN(16)
#undef RUN
        {
            double ms = time_kernel(mul_add_chains<8>, blocks, 64, out, iters);
            double inst = (double)iters * 8 * 8;
            double clk = ms * 1e-3 * 2.4e9;
            printf("mul+add,8,%d,%.3f,%.4f,%.2f\n", wps, ms, inst * wps / clk, inst * 64.0 * blocks / (ms * 1e-3) / 1e12);
        }
    }
    hipFree(out);
    return 0;
}
