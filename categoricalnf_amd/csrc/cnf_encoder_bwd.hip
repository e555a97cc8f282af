// Backward of the mixture-model categorical encoder forward (linear_encoding.py:59-106,153-174) with
// respect to the class table [C, 2D] = [bias | scales_raw] (which PyTorch differentiates further into the
// class embedding and the predictor).  One lane per token; the table gradient is accumulated per
// workgroup in LDS and reduced over workgroups in a fixed order.
//
// With ts = tanh(scales_raw), z = (eps + b_c) e^{ts_c}, v_c = init_lp - sum ts_c + prior_c,
// v_j = sum_d logp(z e^{-ts_j} - b_j) - sum ts_j + prior_j (j != c), q = softmax(v):
//   ldj_tok = (beta (v_c - logsumexp v) - init_lp + sum ts_c) pad,   z_out = z pad.
#include "cnf_common.h"

#include <algorithm>

namespace cnf {

constexpr int kEncBwdMaxP = 2048;
constexpr int kEncBwdGrid = 1024;
constexpr int kEncBwdMaxD = 16;

struct EncBwdArgs {
    const int64_t* categ;
    const float* eps;
    const float* table;
    const float* prior;
    const float* pad;
    const float* g_zout;
    const float* g_ldj;
    float* partials;
    long ntok;
    int N, D, C;
    float beta, sigma, log_sigma;
};

__device__ __forceinline__ float lg_logp(float x, float sigma, float log_sigma) {
    const float v = fabsf(x / sigma);
    return -((v + 2.f * __logf(1.f + __expf(-v))) + log_sigma);
}
__device__ __forceinline__ float lg_dlogp(float x, float sigma) { return -tanhf(0.5f * x / sigma) / sigma; }

__global__ __launch_bounds__(kBlock) void encoder_fwd_bwd_kernel(EncBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* acc = reinterpret_cast<float*>(smem);                 // [C, 2D]: d bias | d tanh(scale)
    float* tab = acc + a.C * 2 * a.D;                              // [C, 3D]: bias | ts | e^{-ts}
    const int D = a.D, C = a.C;
    for (int i = threadIdx.x; i < C * 2 * D; i += kBlock) acc[i] = 0.f;
    for (int i = threadIdx.x; i < C * D; i += kBlock) {
        const int c = i / D, d = i - c * D;
        const float ts = tanhf(a.table[(size_t)c * 2 * D + D + d]);
        tab[c * 3 * D + d] = a.table[(size_t)c * 2 * D + d];
        tab[c * 3 * D + D + d] = ts;
        tab[c * 3 * D + 2 * D + d] = expf(-ts);
    }
    __syncthreads();
    for (long tok = (long)blockIdx.x * kBlock + threadIdx.x; tok < a.ntok; tok += (long)gridDim.x * kBlock) {
        const long long craw = a.categ[tok];      // range-checked (and reported) by the forward kernel; clamped here
        const int c = (int)(craw < 0 ? 0 : (craw >= a.C ? a.C - 1 : craw));
        const float pv = a.pad ? a.pad[tok] : 1.f;
        const float G = (a.g_ldj ? a.g_ldj[tok / a.N] : 0.f) * pv;     // d loss / d ldj_tok (before padding)
        const float* tc = tab + c * 3 * D;
        float z[kEncBwdMaxD], gz[kEncBwdMaxD];
        float init_lp = 0.f, ldj_f = 0.f;
        for (int d = 0; d < D; ++d) {
            const float e = a.eps[tok * D + d];
            init_lp += lg_logp(e, a.sigma, a.log_sigma);
            const float ts = tc[D + d];
            z[d] = (e + tc[d]) * expf(ts);
            ldj_f += ts;
            gz[d] = (a.g_zout ? a.g_zout[tok * D + d] : 0.f) * pv;
        }
        const float v_c = (init_lp - ldj_f) + a.prior[c];
        // pass 1: log-sum-exp of the class scores
        float mx = v_c, s = 1.f;
        for (int j = 0; j < C; ++j) {
            if (j == c) continue;
            const float* tj = tab + j * 3 * D;
            float lp = 0.f, sts = 0.f;
            for (int d = 0; d < D; ++d) {
                lp += lg_logp(z[d] * tj[2 * D + d] - tj[d], a.sigma, a.log_sigma);
                sts += tj[D + d];
            }
            const float v = (lp - sts) + a.prior[j];
            if (v > mx) { s = s * expf(mx - v) + 1.f; mx = v; } else s += expf(v - mx);
        }
        const float lse = mx + logf(s);
        const float Gb = G * a.beta;
        // pass 2: gradients through the other classes' reverse flows
        for (int j = 0; j < C; ++j) {
            if (j == c) continue;
            const float* tj = tab + j * 3 * D;
            float lp = 0.f, sts = 0.f;
            for (int d = 0; d < D; ++d) {
                lp += lg_logp(z[d] * tj[2 * D + d] - tj[d], a.sigma, a.log_sigma);
                sts += tj[D + d];
            }
            const float q = expf(((lp - sts) + a.prior[j]) - lse);
            const float gv = -Gb * q;                                   // d loss / d v_j
            if (gv == 0.f) continue;
            for (int d = 0; d < D; ++d) {
                const float ems = tj[2 * D + d];
                const float l = lg_dlogp(z[d] * ems - tj[d], a.sigma);
                gz[d] += gv * l * ems;
                atomicAdd(&acc[j * 2 * D + d], gv * (-l));
                atomicAdd(&acc[j * 2 * D + D + d], gv * (l * (-z[d] * ems) - 1.f));
            }
        }
        // own class: z = (eps + b_c) e^{ts_c}; ldj_f enters directly (+G) and through v_c (weight beta (1 - q_c))
        const float q_c = expf(v_c - lse);
        const float g_ldjf = G - Gb * (1.f - q_c);
        for (int d = 0; d < D; ++d) {
            atomicAdd(&acc[c * 2 * D + d], gz[d] * expf(tc[D + d]));
            atomicAdd(&acc[c * 2 * D + D + d], gz[d] * z[d] + g_ldjf);
        }
    }
    __syncthreads();
    // chain through tanh and store this workgroup's partial table gradient
    for (int i = threadIdx.x; i < C * 2 * D; i += kBlock) {
        const int c = i / (2 * D), r = i - c * 2 * D;
        float v = acc[i];
        if (r >= D) {
            const float ts = tab[c * 3 * D + D + (r - D)];
            v *= (1.f - ts * ts);
        }
        a.partials[(size_t)blockIdx.x * C * 2 * D + i] = v;
    }
}

__global__ __launch_bounds__(kBlock) void enc_reduce_partials_kernel(const float* partials, int nrows, int P, float* out) {
    const int p = blockIdx.x;
    double accd = 0.0;
    for (int r = threadIdx.x; r < nrows; r += kBlock) accd += (double)partials[(size_t)r * P + p];
    __shared__ double sh[kWavesPerBlock];
    accd = wave_sum(accd);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = accd;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += sh[w];
        out[p] = (float)t;
    }
}

}  // namespace cnf

using namespace cnf;

extern "C" {

int cnf_encoder_forward_bwd(const int64_t* categ, const float* eps, const float* table,
                            const float* category_prior, const float* pad, float beta,
                            const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                            int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream) {
    CNF_REQUIRE(categ && eps && table && category_prior && g_table && workspace, "cnf_encoder_forward_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && C > 0 && D <= kEncBwdMaxD, "cnf_encoder_forward_bwd: bad shape");
    const int P = C * 2 * D;
    if (P > kEncBwdMaxP) { set_error("cnf_encoder_forward_bwd: table with %d entries exceeds %d", P, kEncBwdMaxP); return CNF_ERR_UNSUPPORTED; }
    EncBwdArgs a{categ, eps, table, category_prior, pad, g_zout, g_ldj, workspace, (long)B * N, N, D, C, beta, sigma, log_sigma};
    const size_t smem = sizeof(float) * ((size_t)C * 2 * D + (size_t)C * 3 * D);
    const int grid = (int)std::min<long>(std::max<long>((a.ntok + kBlock - 1) / kBlock, 1), kEncBwdGrid);
    hipStream_t st = (hipStream_t)stream;
    CNF_LAUNCH(encoder_fwd_bwd_kernel, dim3(grid), dim3(kBlock), smem, st, a);
    CNF_LAUNCH(enc_reduce_partials_kernel, dim3(P), dim3(kBlock), 0, st, workspace, grid, P, g_table);
    return launch_status("cnf_encoder_forward_bwd");
}

}  // extern "C"
