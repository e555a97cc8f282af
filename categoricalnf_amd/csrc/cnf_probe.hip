// Diagnostic only: a streaming kernel with the affine coupling's traffic mix (reads 4 + 8 bytes, writes 4 bytes per
// element, 16-byte accesses, nontemporal stores) and next to no arithmetic.  bench.py times it beside the coupling
// kernel so that `roofline` can be read against what THIS device sustains for this read/write mix, not only against
// the 8 TB/s data-sheet figure ("also report a measured stream-copy ceiling from the same run").
// Not part of the reference's interface; nothing in the layers calls it.
#include "cnf_common.h"
#include "cnf_f64_math.h"

namespace cnf {
namespace {

typedef float pf4 __attribute__((ext_vector_type(4)));

template <int U>
__global__ __launch_bounds__(256) void stream_mix_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                         float* __restrict__ out, long nchunks) {
    // a block owns 256*U consecutive 16-byte chunks of `a` / `out` and the matching 32-byte pairs of `b`
    const long base = (long)xcd_block(blockIdx.x, gridDim.x) * (256 * U) + threadIdx.x;     // same workgroup -> XCD layout as the kernels
    pf4 va[U], vb0[U], vb1[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long c = base + (long)u * 256;
        if (c < nchunks) {
            va[u] = __builtin_nontemporal_load(reinterpret_cast<const pf4*>(a + 4 * c));     // read once: 16.3 -> 15.8 us
            vb0[u] = *reinterpret_cast<const pf4*>(b + 8 * c);
            vb1[u] = *reinterpret_cast<const pf4*>(b + 8 * c + 4);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long c = base + (long)u * 256;
        if (c < nchunks) {
            pf4 r;
            r.x = va[u].x + vb0[u].x * vb0[u].y;
            r.y = va[u].y + vb0[u].z * vb0[u].w;
            r.z = va[u].z + vb1[u].x * vb1[u].y;
            r.w = va[u].w + vb1[u].z * vb1[u].w;
            __builtin_nontemporal_store(r, reinterpret_cast<pf4*>(out + 4 * c));
        }
    }
}

// The affine coupling BACKWARD's mix: a [n] + b [2n] + c [n] read (16 B per element), o1 [n] + o2 [2n] written (12 B).
// HINT bit 0: nontemporal loads of a and b (the saved forward tensors), bit 1: of c (the upstream gradient), bit 2:
// nontemporal stores.
template <int U, int HINT>
__global__ __launch_bounds__(256) void stream_mix_bwd_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                             const float* __restrict__ c, float* __restrict__ o1,
                                                             float* __restrict__ o2, long nchunks) {
    const long base = (long)blockIdx.x * (256 * U) + threadIdx.x;
    auto ld = [](const float* p, bool nt) {
        return nt ? __builtin_nontemporal_load(reinterpret_cast<const pf4*>(p)) : *reinterpret_cast<const pf4*>(p);
    };
    auto st = [](float* p, pf4 v) {
        if (HINT & 4) __builtin_nontemporal_store(v, reinterpret_cast<pf4*>(p));
        else *reinterpret_cast<pf4*>(p) = v;
    };
    pf4 va[U], vb0[U], vb1[U], vc[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long k = base + (long)u * 256;
        if (k < nchunks) {
            va[u] = ld(a + 4 * k, HINT & 1);
            vb0[u] = ld(b + 8 * k, HINT & 1);
            vb1[u] = ld(b + 8 * k + 4, HINT & 1);
            vc[u] = ld(c + 4 * k, HINT & 2);
        }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const long k = base + (long)u * 256;
        if (k < nchunks) {
            st(o1 + 4 * k, va[u] * vc[u]);
            st(o2 + 8 * k, vb0[u] + vc[u]);
            st(o2 + 8 * k + 4, vb1[u] * vc[u]);
        }
    }
}

// Experiment (round 4): the affine coupling FORWARD in the token-owner wave-tile form of the backward kernels (coalesced spans
// transposed through LDS, per-channel constants in registers), D = 6, channel mask, scaling factor, hardware transcendentals, rows
// of 64 tokens (32 lanes per row: the row's log-det is a 32-lane butterfly).  Timing partner of cnf_affine_coupling at S*.
typedef float pr_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void probe_lds_order() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
template <int NTH>
__global__ __launch_bounds__(256) void affine_fwd_tile_probe_kernel(const float* z, const float* nn, const float* sf, const float* mask,
                                                                     const float* ldj_in, float* z_out, float* ldj_out, long ntok, int N) {
    constexpr int D = 6, TP = 2, NV = 3, NC = 6, kW = 64;
    __shared__ pr_f4 strip_all[4][3][kW * NV];
    pr_f4* sz = strip_all[threadIdx.x >> 6][0];
    pr_f4* sc0 = strip_all[threadIdx.x >> 6][1];
    pr_f4* sc1 = strip_all[threadIdx.x >> 6][2];
    const int lane = threadIdx.x & 63;
    float keep[D], keepf[D], x2[D], x3[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float m = mask[d], f = expf(sf[d]), fc = fmaxf(f, 1.f);
        keep[d] = 1.f - m;
        keepf[d] = (1.f - m) * f;
        x2[d] = -2.f * keepf[d];
        x3[d] = 2.8853900817779268f / fc;
    }
    const long ntiles = ntok / TP / kW;
    const long wave_id = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
    const int lpr = N / TP;      // lanes per row
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const pr_f4* srcz = reinterpret_cast<const pr_f4*>(z + tile * (kW * TP * D));
        const pr_f4* srcc = reinterpret_cast<const pr_f4*>(nn + tile * (kW * TP * 2 * D));
        pr_f4 qz[NV], qc[NC];
#pragma unroll
        for (int v = 0; v < NV; ++v) qz[v] = __builtin_nontemporal_load(srcz + v * kW + lane);
#pragma unroll
        for (int v = 0; v < NC; ++v) qc[v] = NTH ? __builtin_nontemporal_load(srcc + v * kW + lane) : srcc[v * kW + lane];
        const long row = (tile * kW + lane) * TP / N;
        const float base = ldj_in ? ldj_in[row] : 0.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) sz[v * kW + lane] = qz[v];
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int i = v * kW + lane, grp = i / NC, j = i - grp * NC;
            (j < NV ? sc0 : sc1)[grp * NV + (j < NV ? j : j - NV)] = qc[v];
        }
        probe_lds_order();
        float zi[TP * D], cv[2 * TP * D], out[TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const pr_f4 r = sz[lane * NV + v];
            zi[4 * v] = r.x; zi[4 * v + 1] = r.y; zi[4 * v + 2] = r.z; zi[4 * v + 3] = r.w;
            const pr_f4 c0 = sc0[lane * NV + v];
            cv[4 * v] = c0.x; cv[4 * v + 1] = c0.y; cv[4 * v + 2] = c0.z; cv[4 * v + 3] = c0.w;
            const pr_f4 c1 = sc1[lane * NV + v];
            cv[4 * (NV + v)] = c1.x; cv[4 * (NV + v) + 1] = c1.y; cv[4 * (NV + v) + 2] = c1.z; cv[4 * (NV + v) + 3] = c1.w;
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < TP; ++k)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float sr = cv[k * 2 * D + 2 * d], tr = cv[k * 2 * D + 2 * d + 1];
                const float s = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(sr * x3[d]) + 1.f), x2[d], keepf[d]);
                out[k * D + d] = (zi[k * D + d] + tr * keep[d]) * __builtin_amdgcn_exp2f(s * 1.4426950408889634f);
                acc += s;
            }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const pr_f4 r = {out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
            sz[lane * NV + v] = r;
        }
        for (int m = lpr >> 1; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, kW);
        if ((lane & (lpr - 1)) == 0) ldj_out[row] = base + acc;
        probe_lds_order();
        pr_f4* dz = reinterpret_cast<pr_f4*>(z_out + tile * (kW * TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) __builtin_nontemporal_store(sz[v * kW + lane], dz + v * kW + lane);
        probe_lds_order();
    }
}

// which: 0 log64_pos, 1 rcp64, 2 log1p64_unit; 3 library exp, 4 library log, 5 library 1 / x, 6 library log1p.
// reps > 1 (timing): the function is applied reps times to values derived from the input, results folded together
template <int WHICH>
__device__ __forceinline__ double f64_math_apply(double x) {
    if (WHICH == 0) return log64_pos(x);
    if (WHICH == 1) return rcp64(x);
    if (WHICH == 2) return log1p64_unit(x);
    if (WHICH == 3) return exp(x);
    if (WHICH == 4) return log(x);
    if (WHICH == 5) return 1.0 / x;
    return log1p(x);
}
template <int WHICH>
__global__ __launch_bounds__(256) void f64_math_kernel(const double* in, double* out, long n, int reps) {
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = in[i];
    if (reps <= 1) {
        out[i] = f64_math_apply<WHICH>(x);
        return;
    }
    double acc = 0.0, v = x;
    for (int r = 0; r < reps; ++r) {
        acc += f64_math_apply<WHICH>(v);
        v = x + acc * 1e-300;           // a dependence the compiler cannot remove, numerically void
    }
    out[i] = acc;
}
}  // namespace
}  // namespace cnf

extern "C" int cnf_probe_f64_math(int which, const double* in, double* out, long n, int reps, void* stream) {
    using namespace cnf;
    CNF_REQUIRE(in && out && n > 0 && which >= 0 && which <= 6, "cnf_probe_f64_math: which in 0..6");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)((n + 255) / 256)), block(256);
    switch (which) {
        case 0: CNF_LAUNCH((f64_math_kernel<0>), grid, block, 0, st, in, out, n, reps); break;
        case 1: CNF_LAUNCH((f64_math_kernel<1>), grid, block, 0, st, in, out, n, reps); break;
        case 2: CNF_LAUNCH((f64_math_kernel<2>), grid, block, 0, st, in, out, n, reps); break;
        case 3: CNF_LAUNCH((f64_math_kernel<3>), grid, block, 0, st, in, out, n, reps); break;
        case 4: CNF_LAUNCH((f64_math_kernel<4>), grid, block, 0, st, in, out, n, reps); break;
        case 5: CNF_LAUNCH((f64_math_kernel<5>), grid, block, 0, st, in, out, n, reps); break;
        default: CNF_LAUNCH((f64_math_kernel<6>), grid, block, 0, st, in, out, n, reps); break;
    }
    return launch_status("cnf_probe_f64_math");
}

extern "C" int cnf_stream_probe(const float* a, const float* b, float* out, long n, int chunks_per_lane, void* stream) {
    using namespace cnf;
    CNF_REQUIRE(a && b && out && n > 0 && n % 4 == 0, "cnf_stream_probe: n must be a positive multiple of 4");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long nchunks = n / 4;
    const int U = chunks_per_lane;
    const long blocks = (nchunks + 256L * U - 1) / (256L * U);
    switch (U) {
        case 1: CNF_LAUNCH((stream_mix_kernel<1>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, nchunks); break;
        case 2: CNF_LAUNCH((stream_mix_kernel<2>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, nchunks); break;
        case 4: CNF_LAUNCH((stream_mix_kernel<4>), dim3((unsigned)blocks), dim3(256), 0, st, a, b, out, nchunks); break;
        default: CNF_REQUIRE(false, "cnf_stream_probe: chunks_per_lane must be 1, 2 or 4");
    }
    return launch_status("cnf_stream_probe");
}

extern "C" int cnf_stream_probe_bwd(const float* a, const float* b, const float* c, float* o1, float* o2, long n,
                                    int chunks_per_lane, int hint, void* stream) {
    using namespace cnf;
    CNF_REQUIRE(a && b && c && o1 && o2 && n > 0 && n % 4 == 0, "cnf_stream_probe_bwd: n must be a positive multiple of 4");
    CNF_REQUIRE((chunks_per_lane == 1 || chunks_per_lane == 2) && hint >= 0 && hint < 8, "cnf_stream_probe_bwd: chunks_per_lane in {1,2}, hint in 0..7");
    hipStream_t st = static_cast<hipStream_t>(stream);
    const long nchunks = n / 4;
    const int U = chunks_per_lane;
    const dim3 grid((unsigned)((nchunks + 256L * U - 1) / (256L * U))), block(256);
#define PROBE_BWD(UU, H) CNF_LAUNCH((stream_mix_bwd_kernel<UU, H>), grid, block, 0, st, a, b, c, o1, o2, nchunks)
#define PROBE_BWD_H(UU)                                                                                  \
    switch (hint) {                                                                                      \
        case 0: PROBE_BWD(UU, 0); break; case 1: PROBE_BWD(UU, 1); break; case 2: PROBE_BWD(UU, 2); break; \
        case 3: PROBE_BWD(UU, 3); break; case 4: PROBE_BWD(UU, 4); break; case 5: PROBE_BWD(UU, 5); break; \
        case 6: PROBE_BWD(UU, 6); break; default: PROBE_BWD(UU, 7); break;                                 \
    }
    if (U == 1) { PROBE_BWD_H(1) } else { PROBE_BWD_H(2) }
#undef PROBE_BWD_H
#undef PROBE_BWD
    return launch_status("cnf_stream_probe_bwd");
}

extern "C" int cnf_probe_affine_fwd_tile(const float* z, const float* nn, const float* sf, const float* mask, const float* ldj_in,
                                         float* z_out, float* ldj_out, int B, int N, int groups_per_wave, int nt_hint, void* stream) {
    using namespace cnf;
    CNF_REQUIRE(z && nn && sf && mask && z_out && ldj_out && B > 0 && (N == 64 || N == 32 || N == 16 || N == 128), "cnf_probe_affine_fwd_tile: D = 6 rows of 16..128 tokens");
    const long ntok = (long)B * N, tiles = ntok / 128;
    CNF_REQUIRE(ntok % 128 == 0, "cnf_probe_affine_fwd_tile: whole wave tiles only");
    const int G = groups_per_wave > 0 ? groups_per_wave : 2;
    const dim3 grid((unsigned)std::max<long>((tiles + 4L * G - 1) / (4L * G), 1)), block(256);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (nt_hint) CNF_LAUNCH((affine_fwd_tile_probe_kernel<1>), grid, block, 0, st, z, nn, sf, mask, ldj_in, z_out, ldj_out, ntok, N);
    else CNF_LAUNCH((affine_fwd_tile_probe_kernel<0>), grid, block, 0, st, z, nn, sf, mask, ldj_in, z_out, ldj_out, ntok, N);
    return launch_status("cnf_probe_affine_fwd_tile");
}
