// Shared by the encoder's translation units: cnf_encoder.hip (forward / decode, LDS-resident and class-tiled) and
// cnf_encoder_bwd_tiled.hip (the class-tiled backward).  They are separate files because they are compiled with
// different flags (Makefile: the forward / decode loops lose 3 of 26.5 plain instructions per class without the SLP
// vectoriser, the backward kernels lose time without it).
#pragma once
#include "cnf_common.h"

#include <algorithm>

namespace cnf {

struct EncArgs {
    const int64_t* categ;
    const float* eps;
    const float* z_in;       // decode
    const float* table;      // [C, 2D]
    const float* prior;      // [C]
    const float* pad;        // [B*N] or null
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    float* cpl;              // class_prob_log [B*N] or null
    int64_t* categ_out;
    int* flags;
    int B, N, D, C;
    float beta, sigma, log_sigma;
    // sampler fusion (cnf_encoder_forward_sampled): `eps` holds the UNIFORM draw, the kernel turns it into logistic noise
    // itself (and writes the noise to eps_out if that is set, for the backward)
    int eps_is_u;
    float u_squeeze;
    float* eps_out;
    // ActNorm + 1x1 convolution of the first flow step applied to the latents before they are written
    // (cnf_encoder_forward_actconv); null e_w = no epilogue
    const float* e_bias;
    const float* e_scales;
    const float* e_w;
    const float* e_sldj;
    const float* e_length;
};

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;
constexpr int kEncMaxD = 16;

// LogisticDistribution.sample given the uniform draw (distributions.py:139-145,117-127) in math mode 1: the expression of
// logistic_icdf<true> in cnf_affine.hip (cnf_logistic_from_uniform), mu = 0 — kept literally the same so that the fused
// sampler gives the bits of the two-kernel path (tests/test_gpu_encoder_kernels.py pins that).
__device__ __forceinline__ float enc_noise_from_uniform(float u, float sigma, float squeeze) {
    const float uf = (u * (1.f - squeeze)) + squeeze / 2.f;
    const float v = (__builtin_amdgcn_logf(uf) - __builtin_amdgcn_logf(1.f - uf)) * 0.6931471805599453f;
    return v * sigma + 0.f;
}
// a token's D noise values from token-major [T, D] storage, whichever form the caller handed over: the D loads go out back
// to back, THEN one wave-uniform branch samples (a branch per value would put a memory round trip between the loads)
// MODE: 0 = the caller's noise as it is (no branch at all: the hot LDS-resident kernel is instantiated both ways — with the
// run-time test in it, even untaken, its forward ran 1.5 us slower at the benchmark shape), 1 = sample, 2 = run-time test
// WRITE: false where another launch of the same forward stores the noise (the class-split sweeps: the merge phase does it once)
template <int MODE, int DM, bool WRITE = true>
__device__ __forceinline__ void enc_token_noise(const EncArgs& a, size_t tok, int D, float (&e)[DM]) {
#pragma unroll
    for (int d = 0; d < DM; ++d)
        if (d < D) e[d] = a.eps[tok * D + d];
    if (MODE == 1 || (MODE == 2 && a.eps_is_u)) {
#pragma unroll
        for (int d = 0; d < DM; ++d)
            if (d < D) e[d] = enc_noise_from_uniform(e[d], a.sigma, a.u_squeeze);
        if (WRITE && a.eps_out) {
#pragma unroll
            for (int d = 0; d < DM; ++d)
                if (d < D) a.eps_out[tok * D + d] = e[d];
        }
    }
}

// ---- class chunks of the tiled kernels (see cnf_encoder.hip: "large vocabularies") ----------------------------------
__device__ __forceinline__ int chunk_stride(int D) { return 2 * D + 2; }     // [A0 C0 ... A(D-1) C(D-1) | cst2 | E = 2^cst2]
__device__ __forceinline__ void build_class_chunk(const EncArgs& a, float* tab, int j0, int cc, int D) {
    const int stride = chunk_stride(D);
    const float k = kLog2e / a.sigma;
    for (int i = threadIdx.x; i < cc * D; i += blockDim.x) {
        const int c = i / D, d = i - c * D;
        const float* row = a.table + (size_t)(j0 + c) * 2 * D;
        const float ts = tanhf(row[D + d]);
        tab[c * stride + 2 * d] = expf(-ts) * k;
        tab[c * stride + 2 * d + 1] = row[d] * k;
    }
    for (int c = threadIdx.x; c < cc; c += blockDim.x) {
        const float* row = a.table + (size_t)(j0 + c) * 2 * D;
        float ssum = 0.f;
        for (int d = 0; d < D; ++d) ssum += tanhf(row[D + d]);
        const float cst2 = ((a.prior[j0 + c] - ssum) - (float)D * a.log_sigma) * kLog2e;
        tab[c * stride + 2 * D] = cst2;
        tab[c * stride + 2 * D + 1] = __builtin_amdgcn_exp2f(cst2);
    }
}

// The density-sum posterior (cnf_encoder.hip: class_density) is taken when its total is finite AND the token's own density
// 2^lp2 is above 2^-90: below that, densities of other classes of the same size may have been flushed to zero in the
// products, which would bias the posterior without overflowing anything.  Otherwise: the log-domain sweep.
__device__ __forceinline__ bool density_sum_ok(float tot, float lp2) { return tot < 3e38f && lp2 > -90.f; }

}  // namespace cnf

#define DISPATCH_D(D, CALL)                               \
    switch (D) {                                          \
        case 1: { constexpr int DT = 1; CALL; } break;    \
        case 2: { constexpr int DT = 2; CALL; } break;    \
        case 3: { constexpr int DT = 3; CALL; } break;    \
        case 4: { constexpr int DT = 4; CALL; } break;    \
        case 6: { constexpr int DT = 6; CALL; } break;    \
        case 8: { constexpr int DT = 8; CALL; } break;    \
        default: { constexpr int DT = 0; CALL; } break;   \
    }

static inline int tiled_chunk_classes(int D) { return std::max(1, (int)(32768 / ((2 * D + 2) * sizeof(float)))); }

// class splits of the token-lane kernels: a function of C only (see encoder_tiled_kernel), no empty split
static inline int tiled_class_splits(int C) {
    if (C <= 1024) return 1;
    const int ks = std::min(32, (C + 511) / 512);
    const int per = (C + ks - 1) / ks;
    return (C + per - 1) / per;
}
