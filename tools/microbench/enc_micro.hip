// Standalone microbenchmarks for the categorical-encoder kernels (no torch, no library):
//   1. VALU issue calibration: independent / dependent v_fma_f32, v_exp_f32, v_log_f32, v_pk_fma_f32 and the encoder's own
//      instruction mix at 1 / 2 / 4 / 8 waves per SIMD — per-wave s_memtime cycles and wall time, so that the SQ counters
//      read through rocprofv3 over the same binary can be calibrated against a known instruction stream.
//   2. Arg-max decode variants (linear_encoding.py:184-196): the shipped loop (one token per lane, class constants by
//      ds_read2_b32) against T tokens per lane with the constants read once per class as 16-byte LDS vectors, and against
//      wave-uniform constants fetched through the scalar cache.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off tools/microbench/enc_micro.hip -o build/enc_micro
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <random>
#include <vector>

#define CK(x)                                                                             \
    do {                                                                                  \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess) {                                                           \
            fprintf(stderr, "%s:%d %s: %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); \
            exit(1);                                                                      \
        }                                                                                 \
    } while (0)

// ------------------------------------------------------------------------------------------------ calibration
#define R8(OP)                                                                                                       \
    asm volatile(OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7)                                                     \
                 : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])    \
                 : "v"(b), "v"(c))
#define OP_FMA(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define OP_EXP(i) "v_exp_f32 %" #i ", %" #i "\n"
#define OP_LOG(i) "v_log_f32 %" #i ", %" #i "\n"
#define OP_RCP(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define OP_FMA_DEP(i) "v_fma_f32 %0, %0, %8, %9\n"
#define OP_EXP_DEP(i) "v_exp_f32 %0, %0\n"

// MODE 0 independent fma (8 chains), 1 independent exp, 2 independent log, 3 dependent fma, 4 dependent exp,
//      5 v_pk_fma_f32 (4 pairs), 6 the encoder's mix per element [fma, add|.|, exp -|.|, fma] over 6 channels,
//      7 [3 fma : 1 exp] independent, 8 rcp
template <int MODE>
__global__ __launch_bounds__(256) void calib_kernel(float* out, long long* cyc, int iters) {
    float a[8];
    for (int k = 0; k < 8; ++k) a[k] = 1.0f + 1e-3f * (float)((threadIdx.x + k) & 7);
    float b = 0.999f, c = 1e-3f;
    asm volatile("" : "+v"(b), "+v"(c));
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) { R8(OP_FMA); R8(OP_FMA); R8(OP_FMA); R8(OP_FMA); }
        if (MODE == 1) { R8(OP_EXP); R8(OP_EXP); R8(OP_EXP); R8(OP_EXP); }
        if (MODE == 2) { R8(OP_LOG); R8(OP_LOG); R8(OP_LOG); R8(OP_LOG); }
        if (MODE == 8) { R8(OP_RCP); R8(OP_RCP); R8(OP_RCP); R8(OP_RCP); }
        if (MODE == 3) { R8(OP_FMA_DEP); R8(OP_FMA_DEP); R8(OP_FMA_DEP); R8(OP_FMA_DEP); }
        if (MODE == 4) { R8(OP_EXP_DEP); R8(OP_EXP_DEP); R8(OP_EXP_DEP); R8(OP_EXP_DEP); }
        if (MODE == 5) {
            typedef float f2 __attribute__((ext_vector_type(2)));
            f2 p0 = {a[0], a[1]}, p1 = {a[2], a[3]}, p2 = {a[4], a[5]}, p3 = {a[6], a[7]}, bb = {b, b}, cc = {c, c};
#pragma unroll
            for (int r = 0; r < 8; ++r)
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\nv_pk_fma_f32 %1, %1, %4, %5\nv_pk_fma_f32 %2, %2, %4, %5\nv_pk_fma_f32 %3, %3, %4, %5\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(bb), "v"(cc));
            a[0] = p0.x; a[1] = p0.y; a[2] = p1.x; a[3] = p1.y; a[4] = p2.x; a[5] = p2.y; a[6] = p3.x; a[7] = p3.y;
        }
        if (MODE == 6) {
            // a[0..5] = z of six channels, a[6] = acc, a[7] = prod; 24 VALU + 8 fillers -> count 32 as in the other modes?
            // no: this mode issues exactly 24 instructions per repetition, 4 repetitions = 96; reported per instruction.
            float t0_, t1_, t2_, t3_, t4_, t5_;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile(
                    "v_fma_f32 %8, %0, %14, -%15\n v_fma_f32 %9, %1, %14, -%15\n v_fma_f32 %10, %2, %14, -%15\n"
                    "v_fma_f32 %11, %3, %14, -%15\n v_fma_f32 %12, %4, %14, -%15\n v_fma_f32 %13, %5, %14, -%15\n"
                    "v_add_f32_e64 %6, %6, |%8|\n v_exp_f32_e64 %8, -|%8|\n"
                    "v_add_f32_e64 %6, %6, |%9|\n v_exp_f32_e64 %9, -|%9|\n"
                    "v_add_f32_e64 %6, %6, |%10|\n v_exp_f32_e64 %10, -|%10|\n"
                    "v_add_f32_e64 %6, %6, |%11|\n v_exp_f32_e64 %11, -|%11|\n"
                    "v_add_f32_e64 %6, %6, |%12|\n v_exp_f32_e64 %12, -|%12|\n"
                    "v_add_f32_e64 %6, %6, |%13|\n v_exp_f32_e64 %13, -|%13|\n"
                    "v_fma_f32 %7, %7, %8, %7\n v_fma_f32 %7, %7, %9, %7\n v_fma_f32 %7, %7, %10, %7\n"
                    "v_fma_f32 %7, %7, %11, %7\n v_fma_f32 %7, %7, %12, %7\n v_fma_f32 %7, %7, %13, %7\n"
                    : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7]),
                      "=&v"(t0_), "=&v"(t1_), "=&v"(t2_), "=&v"(t3_), "=&v"(t4_), "=&v"(t5_)
                    : "v"(b), "v"(c));
        }
        if (MODE == 7) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
                asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_exp_f32 %3, %3\n"
                             "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_exp_f32 %7, %7\n"
                             : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5]), "+v"(a[6]), "+v"(a[7])
                             : "v"(b), "v"(c));
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0.f;
    for (int k = 0; k < 8; ++k) s += a[k];
    out[(size_t)blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) cyc[(size_t)blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
}

static int per_iter_insts(int mode) { return mode == 6 ? 96 : (mode == 5 ? 32 : 32); }
static const char* mode_name(int m) {
    static const char* n[] = {"v_fma_f32 x8 independent", "v_exp_f32 x8 independent", "v_log_f32 x8 independent", "v_fma_f32 dependent chain",
                              "v_exp_f32 dependent chain", "v_pk_fma_f32 x4 independent (2 fma each)", "encoder mix [fma,add|.|,exp,fma] x6",
                              "[3 fma : 1 exp] independent", "v_rcp_f32 x8 independent"};
    return n[m];
}
template <int MODE>
static void run_calib(int wps, float* d_out, long long* d_cyc, int iters) {
    const int blocks = 256 * wps;                      // one 4-wave workgroup per CU and per wave-per-SIMD step
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    calib_kernel<MODE><<<blocks, 256>>>(d_out, d_cyc, iters);
    CK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < 3; ++r) {
        CK(hipEventRecord(e0));
        calib_kernel<MODE><<<blocks, 256>>>(d_out, d_cyc, iters);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = std::min(best, ms);
    }
    std::vector<long long> cyc((size_t)blocks * 4);
    CK(hipMemcpy(cyc.data(), d_cyc, cyc.size() * 8, hipMemcpyDeviceToHost));
    double avg = 0;
    for (auto v : cyc) avg += (double)v;
    avg /= (double)cyc.size();
    const double insts = (double)iters * per_iter_insts(MODE);
    printf("calib mode %d %-44s waves/SIMD %d: wall %8.1f us | wave cycles %10.0f | cycles/inst/wave %6.2f | SIMD cycles per inst %5.2f | implied clock %.2f GHz\n",
           MODE, mode_name(MODE), wps, best * 1e3, avg, avg / insts, avg / insts / wps, avg / (best * 1e6));
}

// ------------------------------------------------------------------------------------------------ decode variants
constexpr float kLog2e = 1.4426950408889634f;

// A: the shipped loop.  Derived table layout per class (stride 6D+3): (A, C) pairs at 4D, cst2 at 6D+1.
template <int D>
__global__ __launch_bounds__(256) void decode_a(const float* __restrict__ z_in, const float* __restrict__ dtab, int C, long ntok,
                                                int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem);
    constexpr int stride = 6 * D + 3;
    for (int i = threadIdx.x; i < C * stride; i += 256) tab[i] = dtab[i];
    __syncthreads();
    for (long tok = (long)blockIdx.x * 256 + threadIdx.x; tok < ntok; tok += (long)gridDim.x * 256) {
        float z[D];
#pragma unroll
        for (int d = 0; d < D; ++d) z[d] = z_in[tok * D + d];
        float best = -INFINITY;
        int arg = 0;
        for (int j = 0; j < C; ++j) {
            const float* t = tab + j * stride;
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float vs = fabsf(fmaf(z[d], t[4 * D + 2 * d], -t[4 * D + 2 * d + 1]));
                acc += vs;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            }
            const float v = t[6 * D + 1] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (j == 0 || v > best) { best = v; arg = j; }
        }
        out[tok] = (int64_t)arg;
    }
}

typedef float vf4 __attribute__((ext_vector_type(4)));
// B: T tokens per lane scored together; constants of a class read once as 16-byte LDS vectors.
// LDS / global layout per class: [A0 C0 A1 C1 ... cst2 pad] padded to a multiple of 4 floats.
template <int D, int T, int UNR>
__global__ __launch_bounds__(256) void decode_b(const float* __restrict__ z_in, const float* __restrict__ vtab, int C, long ntok,
                                                int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    float4* tab4 = reinterpret_cast<float4*>(smem);
    for (int i = threadIdx.x; i < C * S / 4; i += 256) tab4[i] = reinterpret_cast<const float4*>(vtab)[i];
    __syncthreads();
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) * T;
    if (base >= ntok) return;
    float z[T][D];
    {
        float buf[T * D];
        static_assert((T * D) % 4 == 0, "token group must be a multiple of 16 bytes");
        const vf4* src = reinterpret_cast<const vf4*>(z_in + base * D);
#pragma unroll
        for (int q = 0; q < T * D / 4; ++q) {
            const vf4 v = __builtin_nontemporal_load(src + q);
            buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int d = 0; d < D; ++d) z[t][d] = buf[t * D + d];
    }
    float best[T];
    int arg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
#pragma unroll UNR
    for (int j = 0; j < C; ++j) {
        float k[S];
#pragma unroll
        for (int q = 0; q < S / 4; ++q) {
            const float4 v = tab4[j * (S / 4) + q];
            k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float vs = fabsf(fmaf(z[t][d], k[2 * d], -k[2 * d + 1]));
                acc += vs;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            }
            const float v = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (v > best[t]) { best[t] = v; arg[t] = j; }
        }
    }
    if (T % 2 == 0) {
        typedef long long ll2 __attribute__((ext_vector_type(2)));
#pragma unroll
        for (int t = 0; t < T; t += 2) {
            ll2 v = {(long long)arg[t], (long long)arg[t + 1]};
            __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + base + t));
        }
    } else {
        for (int t = 0; t < T; ++t) out[base + t] = arg[t];
    }
}

// C: constants wave-uniform from global memory through the scalar cache; one SGPR operand per VALU instruction, so the
// affine map is split: u = z - b ; w = |u| A.  Table layout per class [b0 A0 b1 A1 ... cst2 pad], A > 0.
template <int D, int T, int UNR>
__global__ __launch_bounds__(256) void decode_c(const float* __restrict__ z_in, const float* __restrict__ stab, int C, long ntok,
                                                int64_t* __restrict__ out) {
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    const long base = ((long)blockIdx.x * 256 + threadIdx.x) * T;
    if (base >= ntok) return;
    float z[T][D];
    {
        float buf[T * D];
        const vf4* src = reinterpret_cast<const vf4*>(z_in + base * D);
#pragma unroll
        for (int q = 0; q < T * D / 4; ++q) {
            const vf4 v = __builtin_nontemporal_load(src + q);
            buf[4 * q] = v.x; buf[4 * q + 1] = v.y; buf[4 * q + 2] = v.z; buf[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int d = 0; d < D; ++d) z[t][d] = buf[t * D + d];
    }
    float best[T];
    int arg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
#pragma unroll UNR
    for (int j = 0; j < C; ++j) {
        const float* k = stab + (size_t)j * S;          // uniform address -> s_load
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float u = z[t][d] - k[2 * d];
                const float w = fabsf(u) * k[2 * d + 1];
                acc += w;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-w), prod);
            }
            const float v = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (v > best[t]) { best[t] = v; arg[t] = j; }
        }
    }
    typedef long long ll2 __attribute__((ext_vector_type(2)));
#pragma unroll
    for (int t = 0; t < T; t += 2) {
        ll2 v = {(long long)arg[t], (long long)arg[t + 1]};
        __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + base + t));
    }
}


// D: persistent waves (grid = a few workgroups per CU), T tokens per lane and step scored together against 16-byte LDS
// constants, the NEXT step's latents in flight while the current step is scored, and the derived table built from the raw
// [C, 2D] table AFTER the first loads have been issued (its tanhf / expf hide behind their latency).
template <int D, int T>
__global__ __launch_bounds__(256) void decode_d(const float* __restrict__ z_in, const float* __restrict__ raw, int C, long ntok,
                                                int64_t* __restrict__ out, float sigma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    float* tab = reinterpret_cast<float*>(smem);
    const float4* tab4 = reinterpret_cast<const float4*>(smem);
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const long tiles = (ntok + 64 * T - 1) / (64 * T);
    float zn[T][D];
    auto load = [&](long tile, float (&zz)[T][D]) {
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const long tok = min(tile * (64 * T) + t * 64 + lane, ntok - 1);
#pragma unroll
            for (int d = 0; d < D; ++d) zz[t][d] = __builtin_nontemporal_load(z_in + tok * D + d);
        }
    };
    if (wave < tiles) load(wave, zn);
    {
        const float k = kLog2e / sigma;
        for (int i = threadIdx.x; i < C * D; i += 256) {
            const int c = i / D, d = i - c * D;
            const float ts = tanhf(raw[(size_t)c * 2 * D + D + d]);
            tab[c * S + 2 * d] = expf(-ts) * k;
            tab[c * S + 2 * d + 1] = raw[(size_t)c * 2 * D + d] * k;
        }
        for (int c = threadIdx.x; c < C; c += 256) {
            float ssum = 0.f;
            for (int d = 0; d < D; ++d) ssum += tanhf(raw[(size_t)c * 2 * D + D + d]);
            tab[c * S + 2 * D] = ((-logf((float)C) - ssum) - (float)D * logf(sigma)) * kLog2e;
        }
        __syncthreads();
    }
    for (long tile = wave; tile < tiles; tile += nwaves) {
        float z[T][D];
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int d = 0; d < D; ++d) z[t][d] = zn[t][d];
        if (tile + nwaves < tiles) load(tile + nwaves, zn);
        float best[T];
        int arg[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
        for (int j = 0; j < C; ++j) {
            float k[S];
#pragma unroll
            for (int q = 0; q < S / 4; ++q) {
                const float4 v = tab4[j * (S / 4) + q];
                k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float acc = 0.f, prod = 1.f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float vs = fabsf(fmaf(z[t][d], k[2 * d], -k[2 * d + 1]));
                    acc += vs;
                    prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
                }
                const float v = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
                if (v > best[t]) { best[t] = v; arg[t] = j; }
            }
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const long tok = tile * (64 * T) + t * 64 + lane;
            if (tok < ntok) __builtin_nontemporal_store((long long)arg[t], reinterpret_cast<long long*>(out) + tok);
        }
    }
}

// E: like D, but a lane's T = 2 tokens are CONSECUTIVE (48 B at D = 6: three 16-byte nontemporal loads, one 16-byte
// store of the two indices), the raw table values a thread needs are loaded BEFORE its latents (vmcnt retires in order:
// loads issued behind the latents could not be waited for alone), every tanhf is evaluated once (the per-class sum reads
// the tanh values back from LDS), and BUILD = false copies a ready-made derived table instead (what a cached table costs).
template <int D, bool PREFETCH, bool BUILD>
__global__ __launch_bounds__(256) void decode_e(const float* __restrict__ z_in, const float* __restrict__ raw, const float* __restrict__ vtab,
                                                int C, long ntok, int64_t* __restrict__ out, float sigma) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 2;
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    constexpr int NV = T * D / 4;
    static_assert(T * D % 4 == 0, "");
    float* tab = reinterpret_cast<float*>(smem);
    const float4* tab4 = reinterpret_cast<const float4*>(smem);
    float* ts_sh = tab + C * S;
    const int lane = threadIdx.x & 63;
    const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * 4;
    const long tiles = ntok / (64 * T);                       // micro: ntok is a multiple of 128
    // 1. this thread's raw table entries first
    float r_b = 0.f, r_s = 0.f;
    const int i0 = threadIdx.x;
    if (BUILD && i0 < C * D) {
        const int c = i0 / D, d = i0 - c * D;
        r_b = raw[(size_t)c * 2 * D + d];
        r_s = raw[(size_t)c * 2 * D + D + d];
    }
    // 2. first tile's latents
    vf4 zn[NV];
    auto load = [&](long tile, vf4 (&zz)[NV]) {
        const vf4* src = reinterpret_cast<const vf4*>(z_in + (tile * (64 * T) + (long)lane * T) * D);
#pragma unroll
        for (int q = 0; q < NV; ++q) zz[q] = __builtin_nontemporal_load(src + q);
    };
    if (wave < tiles) load(wave, zn);
    // 3. derived table
    if (BUILD) {
        const float k = kLog2e / sigma;
        for (int i = i0; i < C * D; i += 256) {
            const int c = i / D, d = i - c * D;
            if (i != i0) {
                r_b = raw[(size_t)c * 2 * D + d];
                r_s = raw[(size_t)c * 2 * D + D + d];
            }
            const float ts = tanhf(r_s);
            tab[c * S + 2 * d] = expf(-ts) * k;
            tab[c * S + 2 * d + 1] = r_b * k;
            ts_sh[i] = ts;
        }
        __syncthreads();
        for (int c = threadIdx.x; c < C; c += 256) {
            float ssum = 0.f;
            for (int d = 0; d < D; ++d) ssum += ts_sh[c * D + d];
            tab[c * S + 2 * D] = ((-logf((float)C) - ssum) - (float)D * logf(sigma)) * kLog2e;
        }
    } else {
        for (int i = threadIdx.x; i < C * S / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(vtab)[i];
    }
    __syncthreads();
    for (long tile = wave; tile < tiles; tile += nwaves) {
        float z[T][D];
#pragma unroll
        for (int q = 0; q < NV; ++q) {
            const int e = 4 * q;
            z[e / D][e % D] = zn[q].x; z[(e + 1) / D][(e + 1) % D] = zn[q].y; z[(e + 2) / D][(e + 2) % D] = zn[q].z; z[(e + 3) / D][(e + 3) % D] = zn[q].w;
        }
        if (PREFETCH && tile + nwaves < tiles) load(tile + nwaves, zn);
        float best[T];
        int arg[T];
#pragma unroll
        for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
        for (int j = 0; j < C; ++j) {
            float k[S];
#pragma unroll
            for (int q = 0; q < S / 4; ++q) {
                const float4 v = tab4[j * (S / 4) + q];
                k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
            }
#pragma unroll
            for (int t = 0; t < T; ++t) {
                float acc = 0.f, prod = 1.f;
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float vs = fabsf(fmaf(z[t][d], k[2 * d], -k[2 * d + 1]));
                    acc += vs;
                    prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
                }
                const float v = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
                if (v > best[t]) { best[t] = v; arg[t] = j; }
            }
        }
        typedef long long ll2 __attribute__((ext_vector_type(2)));
        ll2 v = {(long long)arg[0], (long long)arg[1]};
        __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + tile * (64 * T) + (long)lane * T));
        if (!PREFETCH && tile + nwaves < tiles) load(tile + nwaves, zn);
    }
}

// F: E's loop with one tile per wave (2048 workgroups, everything resident) and the loads of the k-th resident workgroup of
// a CU delayed by k * delta: with every wave asking for its latents at t = 0 the memory system serves them all at the same
// rate and nobody computes before the whole 25 MB has arrived; staggered, the first waves of every SIMD start early.
// SLOT 0: k = blockIdx.x / 256 (dispatch order); SLOT 1: k = the hardware wave slot (HW_ID bits 3:0).
template <int D, int SLOT>
__global__ __launch_bounds__(256) void decode_f(const float* __restrict__ z_in, const float* __restrict__ vtab, int C, long ntok,
                                                int64_t* __restrict__ out, int sleep_units) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 2;
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    constexpr int NV = T * D / 4;
    const float4* tab4 = reinterpret_cast<const float4*>(smem);
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    for (int i = threadIdx.x; i < C * S / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(vtab)[i];
    int k;
    if (SLOT == 1) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        k = (int)(hw & 15u);
    } else {
        k = (int)(blockIdx.x >> 8);
    }
    for (int i = 0; i < k * sleep_units; ++i) __builtin_amdgcn_s_sleep(8);       // 8 x 64 cycles per unit
    if (tile * (64 * T) >= ntok) return;
    vf4 zn[NV];
    const vf4* src = reinterpret_cast<const vf4*>(z_in + (tile * (64 * T) + (long)lane * T) * D);
#pragma unroll
    for (int q = 0; q < NV; ++q) zn[q] = __builtin_nontemporal_load(src + q);
    __syncthreads();
    float z[T][D];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int e = 4 * q;
        z[e / D][e % D] = zn[q].x; z[(e + 1) / D][(e + 1) % D] = zn[q].y; z[(e + 2) / D][(e + 2) % D] = zn[q].z; z[(e + 3) / D][(e + 3) % D] = zn[q].w;
    }
    float best[T];
    int arg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
    for (int j = 0; j < C; ++j) {
        float kk[S];
#pragma unroll
        for (int q = 0; q < S / 4; ++q) {
            const float4 v = tab4[j * (S / 4) + q];
            kk[4 * q] = v.x; kk[4 * q + 1] = v.y; kk[4 * q + 2] = v.z; kk[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float vs = fabsf(fmaf(z[t][d], kk[2 * d], -kk[2 * d + 1]));
                acc += vs;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            }
            const float v = kk[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (v > best[t]) { best[t] = v; arg[t] = j; }
        }
    }
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 v = {(long long)arg[0], (long long)arg[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + tile * (64 * T) + (long)lane * T));
}

// T: variant B (T = 2, one tile per wave) with wall-clock stamps (s_memrealtime, 100 MHz) per wave: entry, table ready,
// latents arrived, class loop done, store issued.  One launch, read back and summarised on the host: where the time of a
// launch goes that is not the class loop.
template <int D>
__global__ __launch_bounds__(256) void decode_t(const float* __restrict__ z_in, const float* __restrict__ vtab, int C, long ntok,
                                                int64_t* __restrict__ out, unsigned long long* __restrict__ stamps) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 2;
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    constexpr int NV = T * D / 4;
    const unsigned long long t_entry = __builtin_amdgcn_s_memrealtime();
    const float4* tab4 = reinterpret_cast<const float4*>(smem);
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    vf4 zn[NV];
    const vf4* src = reinterpret_cast<const vf4*>(z_in + (tile * (64 * T) + (long)lane * T) * D);
#pragma unroll
    for (int q = 0; q < NV; ++q) zn[q] = __builtin_nontemporal_load(src + q);
    for (int i = threadIdx.x; i < C * S / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(vtab)[i];
    __syncthreads();
    const unsigned long long t_table = __builtin_amdgcn_s_memrealtime();
    float z[T][D];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int e = 4 * q;
        z[e / D][e % D] = zn[q].x; z[(e + 1) / D][(e + 1) % D] = zn[q].y; z[(e + 2) / D][(e + 2) % D] = zn[q].z; z[(e + 3) / D][(e + 3) % D] = zn[q].w;
    }
    float keep = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) keep += z[0][d] + z[1][d];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("" : "+v"(keep));
    const unsigned long long t_data = __builtin_amdgcn_s_memrealtime();
    float best[T];
    int arg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
    for (int j = 0; j < C; ++j) {
        float kk[S];
#pragma unroll
        for (int q = 0; q < S / 4; ++q) {
            const float4 v = tab4[j * (S / 4) + q];
            kk[4 * q] = v.x; kk[4 * q + 1] = v.y; kk[4 * q + 2] = v.z; kk[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float vs = fabsf(fmaf(z[t][d], kk[2 * d], -kk[2 * d + 1]));
                acc += vs;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            }
            const float v = kk[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (v > best[t]) { best[t] = v; arg[t] = j; }
        }
    }
    asm volatile("" : "+v"(arg[0]), "+v"(arg[1]));
    const unsigned long long t_loop = __builtin_amdgcn_s_memrealtime();
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 v = {(long long)arg[0], (long long)arg[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + tile * (64 * T) + (long)lane * T));
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const unsigned long long t_end = __builtin_amdgcn_s_memrealtime();
    if (lane == 0) {
        unsigned hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        unsigned long long* o = stamps + tile * 6;
        o[0] = t_entry; o[1] = t_table; o[2] = t_data; o[3] = t_loop; o[4] = t_end; o[5] = hw;
    }
}

// G: variant F without the stagger, with a priority schedule: the SIMD arbiter serves the OLDEST ready wave first, so eight
// equal waves finish one after another and the last of them runs alone (the timeline of variant T: ends spread over 5..19 us
// of a 19.6 us launch).  Here a wave lowers its own priority as it advances through the classes (3 -> 0 by quarters), so
// laggards overtake and all waves of a SIMD reach the end together.
template <int D, int MODE>
__global__ __launch_bounds__(256) void decode_g(const float* __restrict__ z_in, const float* __restrict__ vtab, int C, long ntok,
                                                int64_t* __restrict__ out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int T = 2;
    constexpr int S = (2 * D + 1 + 3) / 4 * 4;
    constexpr int NV = T * D / 4;
    const float4* tab4 = reinterpret_cast<const float4*>(smem);
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile * (64 * T) >= ntok) return;
    vf4 zn[NV];
    const vf4* src = reinterpret_cast<const vf4*>(z_in + (tile * (64 * T) + (long)lane * T) * D);
#pragma unroll
    for (int q = 0; q < NV; ++q) zn[q] = __builtin_nontemporal_load(src + q);
    for (int i = threadIdx.x; i < C * S / 4; i += 256) reinterpret_cast<float4*>(smem)[i] = reinterpret_cast<const float4*>(vtab)[i];
    __syncthreads();
    float z[T][D];
#pragma unroll
    for (int q = 0; q < NV; ++q) {
        const int e = 4 * q;
        z[e / D][e % D] = zn[q].x; z[(e + 1) / D][(e + 1) % D] = zn[q].y; z[(e + 2) / D][(e + 2) % D] = zn[q].z; z[(e + 3) / D][(e + 3) % D] = zn[q].w;
    }
    float best[T];
    int arg[T];
#pragma unroll
    for (int t = 0; t < T; ++t) { best[t] = -INFINITY; arg[t] = 0; }
    const int q1 = (C + 3) / 4, q2 = (C + 1) / 2, q3 = (3 * C + 3) / 4;
    if (MODE == 1) __builtin_amdgcn_s_setprio(3);
    for (int j = 0; j < C; ++j) {
        if (MODE == 1) {
            if (j == q1) __builtin_amdgcn_s_setprio(2);
            if (j == q2) __builtin_amdgcn_s_setprio(1);
            if (j == q3) __builtin_amdgcn_s_setprio(0);
        }
        if (MODE == 2) {                                     // the other way round: whoever is ahead stays ahead
            if (j == q1) __builtin_amdgcn_s_setprio(1);
            if (j == q2) __builtin_amdgcn_s_setprio(2);
            if (j == q3) __builtin_amdgcn_s_setprio(3);
        }
        float kk[S];
#pragma unroll
        for (int q = 0; q < S / 4; ++q) {
            const float4 v = tab4[j * (S / 4) + q];
            kk[4 * q] = v.x; kk[4 * q + 1] = v.y; kk[4 * q + 2] = v.z; kk[4 * q + 3] = v.w;
        }
#pragma unroll
        for (int t = 0; t < T; ++t) {
            float acc = 0.f, prod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float vs = fabsf(fmaf(z[t][d], kk[2 * d], -kk[2 * d + 1]));
                acc += vs;
                prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            }
            const float v = kk[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            if (v > best[t]) { best[t] = v; arg[t] = j; }
        }
    }
    typedef long long ll2 __attribute__((ext_vector_type(2)));
    ll2 v = {(long long)arg[0], (long long)arg[1]};
    __builtin_nontemporal_store(v, reinterpret_cast<ll2*>(out + tile * (64 * T) + (long)lane * T));
}

struct Bufs {
    float *z, *ta, *tb, *tc, *raw;
    int64_t *o_ref, *o;
    long ntok;
    int C, D;
};

template <typename F>
static float time_launch(F&& launch, int reps = 20, int blocks = 5) {
    std::vector<hipEvent_t> ev(blocks + 1);
    for (auto& e : ev) CK(hipEventCreate(&e));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(ev[0]));
    for (int b = 0; b < blocks; ++b) {
        for (int r = 0; r < reps; ++r) launch();
        CK(hipEventRecord(ev[b + 1]));
    }
    CK(hipDeviceSynchronize());
    std::vector<float> t;
    for (int b = 1; b < blocks; ++b) {
        float ms;
        CK(hipEventElapsedTime(&ms, ev[b], ev[b + 1]));
        t.push_back(ms / reps * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}
static long mismatches(const Bufs& b) {
    std::vector<int64_t> r(b.ntok), o(b.ntok);
    CK(hipMemcpy(r.data(), b.o_ref, b.ntok * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(o.data(), b.o, b.ntok * 8, hipMemcpyDeviceToHost));
    long n = 0;
    for (long i = 0; i < b.ntok; ++i) n += r[i] != o[i];
    return n;
}

template <int D>
static void run_decode(int C, long ntok) {
    std::mt19937 rng(1234);
    std::normal_distribution<float> nd(0.f, 1.f);
    const float sigma = 1.f / 1.81f;
    const float kk = kLog2e / sigma;
    constexpr int SA = 6 * D + 3, S = (2 * D + 1 + 3) / 4 * 4;
    std::vector<float> ta((size_t)C * SA, 0.f), tb((size_t)C * S, 0.f), tc((size_t)C * S, 0.f), raw((size_t)C * 2 * D, 0.f);
    for (int c = 0; c < C; ++c) {
        float sum_ts = 0.f;
        for (int d = 0; d < D; ++d) {
            const float bias = 2.0f * nd(rng), rs = 0.5f * nd(rng), ts = tanhf(rs);
            raw[c * 2 * D + d] = bias; raw[c * 2 * D + D + d] = rs;
            const float ems = expf(-ts);
            ta[c * SA + 4 * D + 2 * d] = ems * kk;
            ta[c * SA + 4 * D + 2 * d + 1] = bias * kk;
            tb[c * S + 2 * d] = ems * kk;
            tb[c * S + 2 * d + 1] = bias * kk;
            tc[c * S + 2 * d] = bias;
            tc[c * S + 2 * d + 1] = ems * kk;
            sum_ts += ts;
        }
        const float cst = ((-logf((float)C) - sum_ts) - (float)D * logf(sigma)) * kLog2e;
        ta[c * SA + 6 * D + 1] = cst;
        tb[c * S + 2 * D] = cst;
        tc[c * S + 2 * D] = cst;
    }
    std::vector<float> z((size_t)ntok * D);
    for (auto& v : z) v = 2.5f * nd(rng);
    Bufs b;
    b.ntok = ntok; b.C = C; b.D = D;
    CK(hipMalloc(&b.z, z.size() * 4)); CK(hipMalloc(&b.ta, ta.size() * 4)); CK(hipMalloc(&b.tb, tb.size() * 4)); CK(hipMalloc(&b.tc, tc.size() * 4)); CK(hipMalloc(&b.raw, raw.size() * 4));
    CK(hipMemcpy(b.raw, raw.data(), raw.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&b.o_ref, ntok * 8)); CK(hipMalloc(&b.o, ntok * 8));
    CK(hipMemcpy(b.z, z.data(), z.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.ta, ta.data(), ta.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.tb, tb.data(), tb.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(b.tc, tc.data(), tc.size() * 4, hipMemcpyHostToDevice));
    printf("\ndecode variants: %ld tokens, D = %d, C = %d (us per launch, median of 4 blocks of 20; mismatches vs variant A)\n", ntok, D, C);
    {
        const int grid = (int)std::min<long>((ntok + 255) / 256, 2048);
        const size_t lds = (size_t)C * SA * 4;
        float t = time_launch([&] { decode_a<D><<<grid, 256, lds>>>(b.z, b.ta, C, ntok, b.o_ref); });
        printf("  A  shipped loop, 1 token/lane, grid 2048          %7.2f us\n", t);
        const int grid2 = (int)((ntok + 255) / 256);
        t = time_launch([&] { decode_a<D><<<grid2, 256, lds>>>(b.z, b.ta, C, ntok, b.o); });
        printf("  A' shipped loop, 1 token/lane, one pass per lane  %7.2f us  mism %ld\n", t, mismatches(b));
    }
#define RUN_B(T, U)                                                                                                          \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const int grid = (int)((ntok / T + 255) / 256);                                                                      \
        const size_t lds = (size_t)C * S * 4;                                                                                \
        float t = time_launch([&] { decode_b<D, T, U><<<grid, 256, lds>>>(b.z, b.tb, C, ntok, b.o); });                      \
        printf("  B  %d tokens/lane, LDS b128 constants, unroll %d     %7.2f us  mism %ld\n", T, U, t, mismatches(b));        \
    }
#define RUN_C(T, U)                                                                                                          \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const int grid = (int)((ntok / T + 255) / 256);                                                                      \
        float t = time_launch([&] { decode_c<D, T, U><<<grid, 256, 0>>>(b.z, b.tc, C, ntok, b.o); });                        \
        printf("  C  %d tokens/lane, scalar-cache constants, unroll %d %7.2f us  mism %ld\n", T, U, t, mismatches(b));        \
    }
#define RUN_D(T, G)                                                                                                          \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const long tiles = (ntok + 64 * T - 1) / (64 * T);                                                                   \
        const int grid = (int)std::min<long>((tiles + 3) / 4, G);                                                            \
        const size_t lds = (size_t)C * S * 4;                                                                                \
        float t = time_launch([&] { decode_d<D, T><<<grid, 256, lds>>>(b.z, b.raw, C, ntok, b.o, sigma); });                 \
        printf("  D  persistent %4d WGs, %d tokens/lane/step, prefetch   %7.2f us  mism %ld\n", grid, T, t, mismatches(b));    \
    }
#define RUN_E(PF, BLD, G)                                                                                                    \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const long tiles = ntok / 128;                                                                                       \
        const int grid = (int)std::min<long>((tiles + 3) / 4, G);                                                            \
        const size_t lds = (size_t)C * S * 4 + (size_t)C * D * 4;                                                            \
        float t = time_launch([&] { decode_e<D, PF, BLD><<<grid, 256, lds>>>(b.z, b.raw, b.tb, C, ntok, b.o, sigma); });     \
        printf("  E  persistent %4d WGs, 2 consecutive tokens/lane, prefetch %d, table %s %7.2f us  mism %ld\n", grid, (int)PF, BLD ? "built " : "copied", t, mismatches(b));    \
    }
#define RUN_F(SLOT, U)                                                                                                       \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const int grid = (int)((ntok / 128 + 3) / 4);                                                                        \
        const size_t lds = (size_t)C * S * 4;                                                                                \
        float t = time_launch([&] { decode_f<D, SLOT><<<grid, 256, lds>>>(b.z, b.tb, C, ntok, b.o, U); });                   \
        printf("  F  one tile per wave, loads staggered by %s x %d x 512 cycles  %7.2f us  mism %ld\n", SLOT ? "wave slot" : "blockIdx/256", U, t, mismatches(b));    \
    }
    if (C > 0 && D == 6) {
        const long tiles = ntok / 128;
        unsigned long long* d_st;
        CK(hipMalloc(&d_st, tiles * 6 * 8));
        const int grid = (int)((tiles + 3) / 4);
        const size_t lds = (size_t)C * S * 4;
        for (int rep = 0; rep < 3; ++rep) decode_t<D><<<grid, 256, lds>>>(b.z, b.tb, C, ntok, b.o, d_st);
        CK(hipDeviceSynchronize());
        std::vector<unsigned long long> st(tiles * 6);
        CK(hipMemcpy(st.data(), d_st, st.size() * 8, hipMemcpyDeviceToHost));
        unsigned long long t0 = ~0ull, t1 = 0;
        for (long w = 0; w < tiles; ++w) { t0 = std::min(t0, st[w * 6]); t1 = std::max(t1, st[w * 6 + 4]); }
        auto pct = [&](int a, int bcol, double q) {
            std::vector<double> v(tiles);
            for (long w = 0; w < tiles; ++w) v[w] = (double)(st[w * 6 + a] - (bcol < 0 ? t0 : st[w * 6 + bcol])) * 0.01;
            std::sort(v.begin(), v.end());
            return v[(size_t)(q * (tiles - 1))];
        };
        printf("  T  timeline of ONE launch of variant B (T = 2, %ld waves), us (10 ns ticks): span first entry -> last store done %.2f\n", tiles, (double)(t1 - t0) * 0.01);
        const char* nm[] = {"entry since first entry", "table ready since own entry", "latents arrived since own entry", "class loop (data -> loop end)", "store drained since loop end", "own entry -> own end"};
        const int ca[] = {0, 1, 2, 3, 4, 4}, cb[] = {-1, 0, 0, 2, 3, 0};
        for (int i = 0; i < 6; ++i)
            printf("     %-34s min %6.2f  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", nm[i], pct(ca[i], cb[i], 0.0), pct(ca[i], cb[i], 0.1), pct(ca[i], cb[i], 0.5), pct(ca[i], cb[i], 0.9), pct(ca[i], cb[i], 1.0));
        printf("     end of own work since first entry:  p10 %6.2f  p50 %6.2f  p90 %6.2f  max %6.2f\n", pct(4, -1, 0.1), pct(4, -1, 0.5), pct(4, -1, 0.9), pct(4, -1, 1.0));
        // waves per (xcc?, cu, simd) slot census: HW_ID bits: wave 3:0, simd 5:4, cu 11:8, sh 12, se 15:13
        CK(hipFree(d_st));
    }
#define RUN_G(M)                                                                                                             \
    {                                                                                                                        \
        CK(hipMemset(b.o, 0xff, ntok * 8));                                                                                  \
        const int grid = (int)((ntok / 128 + 3) / 4);                                                                        \
        const size_t lds = (size_t)C * S * 4;                                                                                \
        float t = time_launch([&] { decode_g<D, M><<<grid, 256, lds>>>(b.z, b.tb, C, ntok, b.o); });                         \
        printf("  G  one tile per wave, priority schedule %d (0 none, 1 laggards first, 2 leaders first)  %7.2f us  mism %ld\n", M, t, mismatches(b)); \
    }
    RUN_G(0) RUN_G(1) RUN_G(2) RUN_G(0) RUN_G(1) RUN_G(2)
    RUN_F(0, 0) RUN_F(0, 2)
    RUN_E(true, true, 1536) RUN_E(true, true, 2048) RUN_E(false, true, 2048) RUN_E(false, false, 2048)
    RUN_B(2, 1) RUN_B(2, 2) RUN_B(4, 1) RUN_B(4, 2) RUN_B(8, 1)
    if (getenv("MICRO_SCALAR")) { RUN_C(2, 1) RUN_C(2, 2) RUN_C(4, 1) RUN_C(4, 2) RUN_C(8, 1) }
    CK(hipFree(b.z)); CK(hipFree(b.ta)); CK(hipFree(b.tb)); CK(hipFree(b.tc)); CK(hipFree(b.raw)); CK(hipFree(b.o_ref)); CK(hipFree(b.o));
}

int main(int argc, char** argv) {
    const bool only_calib = argc > 1 && !strcmp(argv[1], "calib");
    const bool only_decode = argc > 1 && !strcmp(argv[1], "decode");
    hipDeviceProp_t p;
    CK(hipGetDeviceProperties(&p, 0));
    printf("device %s, %d CUs, clock %d MHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate / 1000);
    if (!only_decode) {
        float* d_out;
        long long* d_cyc;
        CK(hipMalloc(&d_out, (size_t)256 * 8 * 256 * 4));
        CK(hipMalloc(&d_cyc, (size_t)256 * 8 * 4 * 8));
        const int iters = 2000;
        for (int wps : {1, 2, 4, 8}) {
            run_calib<0>(wps, d_out, d_cyc, iters);
            run_calib<1>(wps, d_out, d_cyc, iters);
            run_calib<2>(wps, d_out, d_cyc, iters);
            run_calib<8>(wps, d_out, d_cyc, iters);
            run_calib<3>(wps, d_out, d_cyc, iters);
            run_calib<4>(wps, d_out, d_cyc, iters);
            run_calib<5>(wps, d_out, d_cyc, iters);
            run_calib<6>(wps, d_out, d_cyc, iters);
            run_calib<7>(wps, d_out, d_cyc, iters);
        }
        CK(hipFree(d_out)); CK(hipFree(d_cyc));
    }
    if (!only_calib) {
        run_decode<6>(16, 16384L * 64);
        run_decode<6>(0, 16384L * 64);
        run_decode<6>(51, 16384L * 64);
        run_decode<6>(3, 16384L * 64);
        run_decode<4>(16, 16384L * 16);
    }
    return 0;
}
