// issue_probe.hip -- micro-benchmark (not part of the product): what ONE wave per SIMD pays per instruction on gfx950.
// Long straight-line bodies (256 instructions per loop trip, so the loop-back branch is < 2 % of a trip; the first probe,
// fp64_probe.hip, had 8..128 instructions per trip and its numbers carry ~50 cycles of branch per trip) of inline asm, so
// the instruction sequence is exactly what is written.  Per-wave cycles from s_memtime; the core clock from s_memrealtime
// (100 MHz).  Output: CSV  name,chains,waves_per_simd,cycles_per_instruction,clock_GHz
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define REP4(X) X X X X
#define REP16(X) REP4(X) REP4(X) REP4(X) REP4(X)
#define REP64(X) REP16(X) REP16(X) REP16(X) REP16(X)

struct Stamp { long long cyc, real; };

#define PROBE_PROLOGUE                                                                     \
    double x0 = threadIdx.x * 1e-3 + 1.0, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, \
           x7 = x0 + 7;                                                                    \
    double a = av, b = bv;                                                                 \
    asm volatile("" : "+v"(a), "+v"(b));                                                   \
    const long long t0 = __builtin_readcyclecounter();                                     \
    const long long r0 = wall_clock64();                                                   \
    for (int it = 0; it < iters; ++it) {

#define PROBE_EPILOGUE                                                                     \
    }                                                                                      \
    const long long t1 = __builtin_readcyclecounter();                                     \
    const long long r1 = wall_clock64();                                                   \
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;    \
    if (threadIdx.x % 64 == 0) { st[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = Stamp{t1 - t0, r1 - r0}; }

// OP(x) : one instruction on chain register x
#define FMA(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define FMAC(x) asm volatile("v_fmac_f64 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define MUL(x) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
#define ADD(x) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define FMAS(x) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "s"(av), "v"(b));
#define FMASS(x) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(x) : "s"(av));
#define MAXF(x) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x) : "v"(b));
#define RNDNE(x) asm volatile("v_rndne_f64 %0, %0" : "+v"(x));
#define RCP(x) asm volatile("v_rcp_f64 %0, %0" : "+v"(x));
#define RSQ(x) asm volatile("v_rsq_f64 %0, %0" : "+v"(x));
#define SQRT(x) asm volatile("v_sqrt_f64 %0, %0" : "+v"(x));
#define MOV64(x) asm volatile("v_mov_b64 %0, %1" : "=v"(x) : "v"(a));
#define CNDM(x) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : );
#define FMA_SMOV(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x3ff00000" : "+v"(x) : "v"(a), "v"(b) : "s20");
#define FMA_SNOP(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n s_nop 0" : "+v"(x) : "v"(a), "v"(b));
#define FMA_2SMOV(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n s_mov_b32 s20, 0x3ff00000\n s_mov_b32 s21, 0x3ff00000" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21");
#define FMA_MOV32(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_mov_b32 v200, v201" : "+v"(x) : "v"(a), "v"(b) : "v200");
#define FMA_ACCR(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_accvgpr_read_b32 v200, a0" : "+v"(x) : "v"(a), "v"(b) : "v200");
#define FMA_ACCW(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_accvgpr_write_b32 a0, v201" : "+v"(x) : "v"(a), "v"(b) : "a0");
#define FMA_READLANE(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_readlane_b32 s20, v201, 3" : "+v"(x) : "v"(a), "v"(b) : "s20");
#define FMA_CMP(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_cmp_lt_f64 vcc, %1, %2" : "+v"(x) : "v"(a), "v"(b) : "vcc");
#define FMA_CND(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_cndmask_b32 v200, v201, v202, vcc" : "+v"(x) : "v"(a), "v"(b) : "v200");
// mul then a dependent fma on the same chain
#define MULFMA(x) asm volatile("v_mul_f64 %0, %0, %1\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define ADDMUL(x) asm volatile("v_add_f64 %0, %0, %2\n v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a), "v"(b));

// four dependent instructions in ONE asm statement: the compiler cannot put an s_nop between them
#define FMA4DEP(x) asm volatile("v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define MUL4DEP(x) asm volatile("v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1\n v_mul_f64 %0, %0, %1" : "+v"(x) : "v"(a));
#define FMA2x2(x) asm volatile("v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %0, %0, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(x7) : "v"(a), "v"(b));
#define RSQFMA(x) asm volatile("v_rsq_f64 %0, %0\n v_fma_f64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define RSQ_3FMA(x) asm volatile("v_rsq_f64 %0, %0\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %1, %1, %2, %3\n v_fma_f64 %1, %1, %2, %3" : "+v"(x), "+v"(x7) : "v"(a), "v"(b));
// DPP row broadcast (the cooperative kernels' register Cholesky / Gram): v_mov_b64_dpp alone, and feeding an FMA on another chain
#define DPPB(x) asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "=v"(x) : "v"(a));
#define DPPB_FMA(x) asm volatile("v_mov_b64_dpp %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf\n v_fma_f64 %0, %1, %2, %0" : "+v"(x), "+v"(x7) : "v"(a));
#define FMAC_DPP(x) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(x) : "v"(a), "v"(b));
#define C1Q(OP) REP64(OP(x0))
#define C1(OP) REP64(OP(x0) OP(x0) OP(x0) OP(x0))
#define C2(OP) REP64(OP(x0) OP(x1) OP(x0) OP(x1))
#define C3(OP) REP64(OP(x0) OP(x1) OP(x2)) REP16(OP(x0) OP(x1) OP(x2)) REP4(OP(x0) OP(x1) OP(x2))   // 252
#define C4(OP) REP64(OP(x0) OP(x1) OP(x2) OP(x3))
#define C8(OP) REP16(OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7)) REP16(OP(x0) OP(x1) OP(x2) OP(x3) OP(x4) OP(x5) OP(x6) OP(x7))

#define DEF(NAME, BODY)                                                                              \
    __global__ __launch_bounds__(64) void NAME(double *out, Stamp *st, int iters, double av, double bv) { \
        PROBE_PROLOGUE BODY PROBE_EPILOGUE                                                            \
    }

DEF(fma_c1, C1(FMA)) DEF(fma_c2, C2(FMA)) DEF(fma_c3, C3(FMA)) DEF(fma_c4, C4(FMA)) DEF(fma_c8, C8(FMA))
DEF(fmac_c1, C1(FMAC)) DEF(fmac_c2, C2(FMAC)) DEF(fmac_c8, C8(FMAC))
DEF(mul_c1, C1(MUL)) DEF(mul_c2, C2(MUL)) DEF(mul_c4, C4(MUL)) DEF(mul_c8, C8(MUL))
DEF(add_c1, C1(ADD)) DEF(add_c2, C2(ADD)) DEF(add_c4, C4(ADD)) DEF(add_c8, C8(ADD))
DEF(fmas_c8, C8(FMAS)) DEF(fmass_c8, C8(FMASS)) DEF(fmas_c2, C2(FMAS))
DEF(max_c1, C1(MAXF)) DEF(max_c8, C8(MAXF))
DEF(rndne_c1, C1(RNDNE)) DEF(rndne_c8, C8(RNDNE))
DEF(rcp_c1, C1(RCP)) DEF(rcp_c8, C8(RCP))
DEF(rsq_c1, C1(RSQ)) DEF(rsq_c8, C8(RSQ))
DEF(sqrt_c1, C1(SQRT)) DEF(sqrt_c8, C8(SQRT))
DEF(mov64_c8, C8(MOV64))
DEF(fma_smov_c8, C8(FMA_SMOV)) DEF(fma_2smov_c8, C8(FMA_2SMOV)) DEF(fma_snop_c8, C8(FMA_SNOP))
DEF(fma_smov_c1, C1(FMA_SMOV)) DEF(fma_smov_c2, C2(FMA_SMOV))
DEF(fma_mov32_c8, C8(FMA_MOV32)) DEF(fma_accr_c8, C8(FMA_ACCR)) DEF(fma_accw_c8, C8(FMA_ACCW))
DEF(fma_readlane_c8, C8(FMA_READLANE)) DEF(fma_cmp_c8, C8(FMA_CMP)) DEF(fma_cnd_c8, C8(FMA_CND))
DEF(fma_mov32_c1, C1(FMA_MOV32)) DEF(fma_mov32_c2, C2(FMA_MOV32))
DEF(mulfma_c1, C1(MULFMA)) DEF(mulfma_c2, C2(MULFMA)) DEF(mulfma_c4, C4(MULFMA)) DEF(mulfma_c8, C8(MULFMA))
DEF(addmul_c1, C1(ADDMUL)) DEF(addmul_c8, C8(ADDMUL))
DEF(dppb_c8, C8(DPPB)) DEF(dppb_fma, C1Q(DPPB_FMA)) DEF(dppb_fmac_c8, C8(FMAC_DPP)) DEF(dppb_fmac_c1, C1(FMAC_DPP))
DEF(fma4dep, C1Q(FMA4DEP)) DEF(mul4dep, C1Q(MUL4DEP)) DEF(fma2x2, C1Q(FMA2x2)) DEF(rsqfma, C1(RSQFMA)) DEF(rsq_3fma, C1Q(RSQ_3FMA))

typedef void (*Kern)(double *, Stamp *, int, double, double);
struct Entry { const char *name; Kern k; int chains; int inst_per_trip; };
#define E(NAME, CH, N) Entry{#NAME, NAME, CH, N}

int main(int argc, char **argv) {
    const int iters = 400;
    const int max_waves = 256 * 4 * 4;
    double *out; Stamp *st;
    hipMalloc(&out, sizeof(double) * max_waves * 64);
    hipMalloc(&st, sizeof(Stamp) * max_waves);
    std::vector<Stamp> h(max_waves);
    const Entry es[] = {
        E(fma_c1, 1, 256), E(fma_c2, 2, 256), E(fma_c3, 3, 252), E(fma_c4, 4, 256), E(fma_c8, 8, 256),
        E(fmac_c1, 1, 256), E(fmac_c2, 2, 256), E(fmac_c8, 8, 256),
        E(mul_c1, 1, 256), E(mul_c2, 2, 256), E(mul_c4, 4, 256), E(mul_c8, 8, 256),
        E(add_c1, 1, 256), E(add_c2, 2, 256), E(add_c4, 4, 256), E(add_c8, 8, 256),
        E(fmas_c8, 8, 256), E(fmass_c8, 8, 256), E(fmas_c2, 2, 256),
        E(max_c1, 1, 256), E(max_c8, 8, 256), E(rndne_c1, 1, 256), E(rndne_c8, 8, 256),
        E(rcp_c1, 1, 256), E(rcp_c8, 8, 256), E(rsq_c1, 1, 256), E(rsq_c8, 8, 256), E(sqrt_c1, 1, 256), E(sqrt_c8, 8, 256),
        E(mov64_c8, 8, 256),
        E(fma_smov_c8, 8, 512), E(fma_2smov_c8, 8, 768), E(fma_snop_c8, 8, 512), E(fma_smov_c1, 1, 512), E(fma_smov_c2, 2, 512),
        E(fma_mov32_c8, 8, 512), E(fma_accr_c8, 8, 512), E(fma_accw_c8, 8, 512), E(fma_readlane_c8, 8, 512),
        E(fma_cmp_c8, 8, 512), E(fma_cnd_c8, 8, 512), E(fma_mov32_c1, 1, 512), E(fma_mov32_c2, 2, 512),
        E(mulfma_c1, 1, 512), E(mulfma_c2, 2, 512), E(mulfma_c4, 4, 512), E(mulfma_c8, 8, 512),
        E(addmul_c1, 1, 512), E(addmul_c8, 8, 512),
        E(dppb_c8, 8, 256), E(dppb_fma, 1, 128), E(dppb_fmac_c8, 8, 256), E(dppb_fmac_c1, 1, 256),
        E(fma4dep, 1, 256), E(mul4dep, 1, 256), E(fma2x2, 2, 256), E(rsqfma, 1, 512), E(rsq_3fma, 2, 256),
    };
    printf("name,chains,waves_per_simd,cycles_per_instruction,cycles_per_trip,clock_GHz,event_ms\n");
    for (int wps : {1, 2}) {
        const int blocks = 256 * 4 * wps;
        for (const Entry &e : es) {
            if (argc > 1 && !strstr(e.name, argv[1])) continue;
            hipEvent_t e0, e1;
            hipEventCreate(&e0); hipEventCreate(&e1);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, st, iters, 0.999999, 1e-7);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            hipLaunchKernelGGL(e.k, dim3(blocks), dim3(64), 0, 0, out, st, iters, 0.999999, 1e-7);
            hipEventRecord(e1);
            hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            hipMemcpy(h.data(), st, sizeof(Stamp) * blocks, hipMemcpyDeviceToHost);
            double cyc = 0, real = 0;
            for (int i = 0; i < blocks; ++i) { cyc += h[i].cyc; real += h[i].real; }
            cyc /= blocks; real /= blocks;
            printf("%s,%d,%d,%.3f,%.1f,%.3f,%.4f\n", e.name, e.chains, wps, cyc / ((double)iters * e.inst_per_trip),
                   cyc / iters, cyc / (real * 10.0) , ms);   // real: 100 MHz ticks -> ns = real * 10
            hipEventDestroy(e0); hipEventDestroy(e1);
        }
    }
    hipFree(out); hipFree(st);
    return 0;
}
