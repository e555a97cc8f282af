// Issue cost of single VALU opcodes on gfx950 (standalone, no torch): 8 independent chains of ONE opcode per wave, 256 CUs x
// 4 SIMDs x W waves per SIMD, wall time per wave-instruction per SIMD.  The encoder kernels' "plain" instructions are not all
// v_fma_f32: this table says which of them issue at the 2-cycle rate (profiles/r03_valu_calibration.txt measured fma / exp /
// log / rcp / pk_fma only).
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/op_rates.hip -o /tmp/op_rates && /tmp/op_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

#define R8(OP)                                                                                                       \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                     \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
                 : "v"(b), "v"(c))
#define OP0(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP1(i) "v_mul_f32 %" #i ", %" #i ", %8\n"
#define OP2(i) "v_add_f32 %" #i ", %" #i ", %8\n"
#define OP3(i) "v_sub_f32 %" #i ", %8, %" #i "\n"
#define OP4(i) "v_fmac_f32 %" #i ", %8, %9\n"
#define OP5(i) "v_bfi_b32 %" #i ", %8, %" #i ", %9\n"
#define OP6(i) "v_cndmask_b32 %" #i ", %" #i ", %8, vcc\n"
#define OP7(i) "v_max_f32 %" #i ", %" #i ", %8\n"
#define OP8(i) "v_and_b32 %" #i ", %" #i ", %8\n"
#define OP9(i) "v_mov_b32 %" #i ", %8\n"
#define OP10(i) "v_add_f32_e64 %" #i ", %" #i ", |%8|\n"
#define OP11(i) "v_mul_f32_e64 %" #i ", -%" #i ", %8\n"
#define OP12(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP13(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP14(i) "v_add_f32_dpp %" #i ", %" #i ", %8 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
#define OP15(i) "v_add_f32_dpp %" #i ", %" #i ", %8 row_shr:1 row_mask:0xf bank_mask:0xf\n"
#define OP16(i) "v_add_u32 %" #i ", %" #i ", %8\n"
#define OP17(i) "v_lshlrev_b32 %" #i ", 1, %" #i "\n"
#define OP18(i) "v_mad_u32_u24 %" #i ", %" #i ", %8, %9\n"
#define OP19(i) "v_cmp_lt_f32 vcc, %" #i ", %8\n"
#define OP20(i) "v_exp_f32_e64 %" #i ", -|%" #i "|\n"
#define OP21(i) "v_log_f32 %" #i ", %" #i "\n"
#define OP22(i) "v_mul_legacy_f32 %" #i ", %" #i ", %8\n"
#define OP23(i) "v_fma_f32 %" #i ", %" #i ", %8, -%9\n"
#define OP24(i) "v_cndmask_b32_e64 %" #i ", %" #i ", %8, s[10:11]\n"
#define OP25(i) "v_cndmask_b32_e64 %" #i ", 0, %" #i ", s[10:11]\n"
#define OP28(i) "v_min_f32 %" #i ", %" #i ", %8\n"
#define OP29(i) "v_xor_b32 %" #i ", %" #i ", %8\n"
#define OP30(i) "v_cvt_f32_i32 %" #i ", %" #i "\n"
#define OP31(i) "v_mul_u32_u24 %" #i ", %" #i ", %8\n"

static const char* kNames[] = {"v_fma_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_fmac_f32", "v_bfi_b32 (copysign)", "v_cndmask_b32",
                               "v_max_f32", "v_and_b32", "v_mov_b32", "v_add_f32 |src| (VOP3)", "v_mul_f32 -src (VOP3)", "v_exp_f32", "v_rcp_f32",
                               "v_add_f32 dpp quad_perm", "v_add_f32 dpp row_shr:1", "v_add_u32", "v_lshlrev_b32", "v_mad_u32_u24", "v_cmp_lt_f32",
                               "v_exp_f32 -|src| (VOP3)", "v_log_f32", "v_mul_legacy_f32", "v_fma_f32 neg src (VOP3)",
                               "v_cndmask_b32 sgpr mask", "v_cndmask_b32 0, src, sgpr mask", "v_pk_mul_f32 (2 mul each)", "v_pk_add_f32 (2 add each)",
                               "v_min_f32", "v_xor_b32", "v_cvt_f32_i32", "v_mul_u32_u24"};
constexpr int kModes = 32;

template <int MODE>
__global__ __launch_bounds__(256) void op_kernel(float* out, int iters) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0f + 1e-3f * (float)((threadIdx.x + k) & 7);
    float b = 0.999f, c = 1e-3f;
    asm volatile("" : "+v"(b), "+v"(c));
    asm volatile("s_mov_b64 s[10:11], exec\n s_mov_b64 vcc, exec" ::: "s10", "s11", "vcc");
    for (int it = 0; it < iters; ++it) {
#define RUN(M, OP) if (MODE == M) { R8(OP); R8(OP); R8(OP); R8(OP); }
        RUN(0, OP0) RUN(1, OP1) RUN(2, OP2) RUN(3, OP3) RUN(4, OP4) RUN(5, OP5) RUN(6, OP6) RUN(7, OP7) RUN(8, OP8) RUN(9, OP9) RUN(10, OP10)
        RUN(11, OP11) RUN(12, OP12) RUN(13, OP13) RUN(14, OP14) RUN(15, OP15) RUN(16, OP16) RUN(17, OP17) RUN(18, OP18) RUN(19, OP19)
        RUN(20, OP20) RUN(21, OP21) RUN(22, OP22) RUN(23, OP23)
        RUN(24, OP24) RUN(25, OP25) RUN(28, OP28) RUN(29, OP29) RUN(30, OP30) RUN(31, OP31)
        if (MODE == 26 || MODE == 27) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]}, p2 = {a[4], a[5]}, p3 = {a[6], a[7]}, bb = {b, b};
#pragma unroll
            for (int r = 0; r < 8; ++r) {
                if (MODE == 26) asm volatile("v_pk_mul_f32 %0, %0, %4\nv_pk_mul_f32 %1, %1, %4\nv_pk_mul_f32 %2, %2, %4\nv_pk_mul_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
                else asm volatile("v_pk_add_f32 %0, %0, %4\nv_pk_add_f32 %1, %1, %4\nv_pk_add_f32 %2, %2, %4\nv_pk_add_f32 %3, %3, %4\n" : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb));
            }
            a[0] = p0.x; a[1] = p0.y; a[2] = p1.x; a[3] = p1.y; a[4] = p2.x; a[5] = p2.y; a[6] = p3.x; a[7] = p3.y;
        }
    }
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}

#define RD8(OP)                                                                                                      \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                     \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
                 : "v"(b), "v"(c))
#define DP0(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define DP1(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define DP2(i) "v_add_f64 %" #i ", %" #i ", %8\n"
#define DP3(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define DP4(i) "v_ldexp_f64 %" #i ", %" #i ", 1\n"
#define DP5(i) "v_max_f64 %" #i ", %" #i ", %8\n"
#define DP6(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define DP7(i) "v_fract_f64 %" #i ", %" #i "\n"
#define DP8(i) "v_add_f64 %" #i ", %" #i ", |%8|\n"
static const char* kNames64[] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rcp_f64", "v_ldexp_f64", "v_max_f64", "v_rsq_f64", "v_fract_f64", "v_add_f64 |src|"};
constexpr int kModes64 = 9;
template <int MODE>
__global__ __launch_bounds__(256) void op64_kernel(double* out, int iters) {
    double a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0 + 1e-3 * (double)((threadIdx.x + k) & 7);
    double b = 0.999, c = 1e-3;
    asm volatile("" : "+v"(b), "+v"(c));
    for (int it = 0; it < iters; ++it) {
#define RUN64(M, OP) if (MODE == M) { RD8(OP); RD8(OP); RD8(OP); RD8(OP); }
        RUN64(0, DP0) RUN64(1, DP1) RUN64(2, DP2) RUN64(3, DP3) RUN64(4, DP4) RUN64(5, DP5) RUN64(6, DP6) RUN64(7, DP7) RUN64(8, DP8)
    }
    double s = 0.0;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
static double run64(int wps, double* d_out, int iters) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    op64_kernel<MODE><<<blocks, 256>>>(d_out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        op64_kernel<MODE><<<blocks, 256>>>(d_out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    return (double)best * 1e6 / ((double)wps * iters * 32.0);
}
template <int MODE>
static void row64(double* d_out, int iters) {
    printf("%-28s", kNames64[MODE]);
    for (int wps : {1, 2, 4, 8}) printf("  %6.3f", run64<MODE>(wps, d_out, iters));
    printf("\n");
    if constexpr (MODE + 1 < kModes64) row64<MODE + 1>(d_out, iters);
}

template <int MODE>
static double run(int wps, float* d_out, int iters) {
    const int blocks = 256 * wps;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    op_kernel<MODE><<<blocks, 256>>>(d_out, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        op_kernel<MODE><<<blocks, 256>>>(d_out, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = ms < best ? ms : best;
    }
    // wave-instructions per SIMD = wps waves x iters x 32
    return (double)best * 1e6 / ((double)wps * iters * 32.0);      // ns per wave-instruction per SIMD
}

template <int MODE>
static void row(float* d_out, int iters) {
    printf("%-28s", kNames[MODE]);
    for (int wps : {1, 2, 4, 8}) printf("  %6.3f", run<MODE>(wps, d_out, iters));
    printf("\n");
    if constexpr (MODE + 1 < kModes) row<MODE + 1>(d_out, iters);
}

int main() {
    float* d_out;
    CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * sizeof(float)));
    printf("ns per wave-instruction per SIMD (8 independent chains per wave; v_fma_f32 = 2 cycles at the clock the chip holds)\n");
    printf("%-28s  %6s  %6s  %6s  %6s   (waves per SIMD)\n", "opcode", "1", "2", "4", "8");
    row<0>(d_out, 2000);
    double* d_out64;
    CK(hipMalloc(&d_out64, (size_t)256 * 8 * 256 * sizeof(double)));
    printf("fp64 (v_fma_f64 at the 78.6 TFLOP/s vector peak = 4 cycles per wave-instruction)\n");
    row64<0>(d_out64, 1000);
    return 0;
}
