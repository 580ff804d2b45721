#include <hip/hip_runtime.h>
#include "device/chain_hot.hpp"
using namespace ikdev;
using S = ChainStruct<355820159695091300ul, 4132432860610451030ul, 1970497541718ul>;
constexpr int NJ = 7;
extern "C" __global__ void k_sincos(const double *in, double *out) {
    double s[NJ], c[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) dsincos_hot(in[j * 64 + threadIdx.x], s[j], c[j]);
#pragma unroll
    for (int j = 0; j < NJ; ++j) { out[(2 * j) * 64 + threadIdx.x] = s[j]; out[(2 * j + 1) * 64 + threadIdx.x] = c[j]; }
}
extern "C" __global__ void k_evaluate(const double *in, double *out, const HotTable t) {
    double q[NJ], oMt[12], e[6], col[NJ][6];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = in[j * 64 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < 12; ++j) oMt[j] = in[(NJ + j) * 64 + threadIdx.x];
    hot_evaluate<NJ, S>(t, q, oMt, e, col);
#pragma unroll
    for (int j = 0; j < 6; ++j) out[j * 64 + threadIdx.x] = e[j];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) out[(6 + 6 * j + i) * 64 + threadIdx.x] = col[j][i];
}
extern "C" __global__ void k_gram(const double *in, double *out) {
    double col[NJ][6], G[36];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) col[j][i] = in[(6 * j + i) * 64 + threadIdx.x];
    hot_gram<NJ, S>(col, in[4096], G);
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) out[(6 * i + j) * 64 + threadIdx.x] = G[6 * i + j];
}
extern "C" __global__ void k_chol(const double *in, double *out) {
    double G[36], e[6], y[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
        for (int j = 0; j <= i; ++j) G[6 * i + j] = in[(6 * i + j) * 64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 6; ++i) e[i] = in[(36 + i) * 64 + threadIdx.x];
    chol_solve<6>(G, e, y);
#pragma unroll
    for (int i = 0; i < 6; ++i) out[i * 64 + threadIdx.x] = y[i];
}
extern "C" __global__ void k_step(const double *in, double *out, const HotTable t, const LoopParams prm) {
    double col[NJ][6], y[6], q[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int i = 0; i < 6; ++i) col[j][i] = in[(6 * j + i) * 64 + threadIdx.x];
#pragma unroll
    for (int i = 0; i < 6; ++i) y[i] = in[(42 + i) * 64 + threadIdx.x];
#pragma unroll
    for (int j = 0; j < NJ; ++j) q[j] = in[(48 + j) * 64 + threadIdx.x];
    hot_step<NJ, S>(t, prm, col, y, q, true);
#pragma unroll
    for (int j = 0; j < NJ; ++j) out[j * 64 + threadIdx.x] = q[j];
}
