// Dependent-chain cost of v_fma_f64 / v_fma_f32 on gfx950: C independent chains per wave (C = 1, 2, 4), W waves per SIMD;
// prints ns per wave-instruction per SIMD.  One chain and one wave = the back-to-back latency of the instruction; the
// number of (waves x chains) at which the figure reaches the issue rate says how much parallelism hides it.
// Build + run: hipcc --offload-arch=gfx950 -O3 tools/microbench/dep_latency.hip -o /tmp/dep_latency && /tmp/dep_latency
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int C, bool F64>
__global__ __launch_bounds__(256) void chain_kernel(double* out, int iters) {
    double a[4]; float f[4];
    for (int k = 0; k < 4; ++k) { a[k] = 1.0 + 1e-3 * (double)((threadIdx.x + k) & 7); f[k] = (float)a[k]; }
    double b = 0.999, c = 1e-3; float bf = 0.999f, cf = 1e-3f;
    asm volatile("" : "+v"(b), "+v"(c), "+v"(bf), "+v"(cf));
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 32 / C; ++r) {
            if (F64) {
                if (C == 1) asm volatile("v_fma_f64 %0, %0, %1, %2\n" : "+v"(a[0]) : "v"(b), "v"(c));
                if (C == 2) asm volatile("v_fma_f64 %0, %0, %2, %3\nv_fma_f64 %1, %1, %2, %3\n" : "+v"(a[0]), "+v"(a[1]) : "v"(b), "v"(c));
                if (C == 4) asm volatile("v_fma_f64 %0, %0, %4, %5\nv_fma_f64 %1, %1, %4, %5\nv_fma_f64 %2, %2, %4, %5\nv_fma_f64 %3, %3, %4, %5\n"
                                         : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]) : "v"(b), "v"(c));
            } else {
                if (C == 1) asm volatile("v_fma_f32 %0, %0, %1, %2\n" : "+v"(f[0]) : "v"(bf), "v"(cf));
                if (C == 2) asm volatile("v_fma_f32 %0, %0, %2, %3\nv_fma_f32 %1, %1, %2, %3\n" : "+v"(f[0]), "+v"(f[1]) : "v"(bf), "v"(cf));
                if (C == 4) asm volatile("v_fma_f32 %0, %0, %4, %5\nv_fma_f32 %1, %1, %4, %5\nv_fma_f32 %2, %2, %4, %5\nv_fma_f32 %3, %3, %4, %5\n"
                                         : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "v"(bf), "v"(cf));
            }
        }
    }
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = a[0] + a[1] + a[2] + a[3] + f[0] + f[1] + f[2] + f[3];
}

template <int C, bool F64>
static void row(double* d_out, int iters) {
    printf("%-10s %d chain(s) per wave", F64 ? "v_fma_f64" : "v_fma_f32", C);
    for (int wps : {1, 2, 3, 4, 6, 8}) {
        // W waves per SIMD: workgroups of 4 waves (one per SIMD), wps workgroups per CU
        const int blocks = 256 * wps;
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        chain_kernel<C, F64><<<blocks, 256>>>(d_out, iters);
        CK(hipDeviceSynchronize());
        float best = 1e30f;
        for (int r = 0; r < 3; ++r) {
            CK(hipEventRecord(e0));
            chain_kernel<C, F64><<<blocks, 256>>>(d_out, iters);
            CK(hipEventRecord(e1));
            CK(hipEventSynchronize(e1));
            float ms; CK(hipEventElapsedTime(&ms, e0, e1));
            best = ms < best ? ms : best;
        }
        printf("  %6.3f", (double)best * 1e6 / ((double)wps * iters * 32.0));
    }
    printf("\n");
}

int main() {
    double* d_out; CK(hipMalloc(&d_out, sizeof(double) * 256 * 256 * 8));
    const int iters = 20000;
    printf("ns per wave-instruction per SIMD; columns: 1 2 3 4 6 8 waves per SIMD\n");
    row<1, true>(d_out, iters); row<2, true>(d_out, iters); row<4, true>(d_out, iters);
    row<1, false>(d_out, iters); row<2, false>(d_out, iters); row<4, false>(d_out, iters);
    return 0;
}
