// Diagnostic only: a streaming kernel with the affine coupling's traffic mix (reads 4 + 8 bytes, writes 4 bytes per
// element, 16-byte accesses, nontemporal stores) and next to no arithmetic.  bench.py times it beside the coupling
// kernel so that `roofline` can be read against what THIS device sustains for this read/write mix, not only against
// the 8 TB/s data-sheet figure ("also report a measured stream-copy ceiling from the same run").
// Not part of the reference's interface; nothing in the layers calls it.
#include "cnf_common.h"

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

}  // namespace
}  // namespace cnf

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
