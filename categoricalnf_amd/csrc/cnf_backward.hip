// Backward (vector-Jacobian) kernels of the fp32 flow layers: affine coupling (both directions),
// ExtActNorm, ActNorm, 1x1 convolution, logistic prior / NLL and the sigmoid flow.
//
// The reference has no backward code: it differentiates its eager op chains with autograd
// (SURVEY.md §3.3).  These kernels compute the same gradients in one pass per layer:
//   inputs   saved forward tensors + upstream grads g_zout [B,N,D], g_ldj [B]
//   outputs  g_z, g_nn (element-wise, streamed) and the parameter gradients, which are reductions
//            over the whole batch: every workgroup accumulates them in LDS, writes one row of
//            `partials[gridDim.x][P]`, and reduce_partials_kernel sums the rows in fp64 in a fixed
//            order (deterministic; no global atomics).
#include "cnf_common.h"

#include <algorithm>

namespace cnf {

constexpr int kBwdMaxP = 1024;      // parameter-gradient entries per kernel
constexpr int kBwdGrid = 1024;      // workgroups (grid-stride) = rows of the partials buffer

// cnf_mixture_bwd.hip: column sums of [nrows, P] partials, columns [0, split) -> out_a, the rest -> out_b
__global__ void mix_reduce_partials_kernel(const float* partials, int nrows, int P, float* out_a, float* out_b, int split);

__global__ __launch_bounds__(kBlock) void reduce_partials_kernel(const float* partials, int nrows, int P,
                                                                 float* out) {
    // one workgroup per parameter entry
    const int p = blockIdx.x;
    double acc = 0.0;
    for (int r = threadIdx.x; r < nrows; r += kBlock) acc += (double)partials[(size_t)r * P + p];
    __shared__ double sh[kWavesPerBlock];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kWavesPerBlock; ++w) t += sh[w];
        out[p] = (float)t;
    }
}

__device__ __forceinline__ void flush_partials(const float* lds, int P, float* partials) {
    __syncthreads();
    for (int i = threadIdx.x; i < P; i += kBlock) partials[(size_t)blockIdx.x * P + i] = lds[i];
}

// ---- affine coupling ------------------------------------------------------------------------------
struct AffBwdArgs {
    const float* z_out;     // forward OUTPUT of the direction being differentiated
    const float* nn;
    const float* sf;        // nullable
    const float* mask;
    const float* g_zout;    // nullable (= zeros)
    const float* g_ldj;     // nullable
    float* g_z;
    float* g_nn;
    float* partials;        // [gridDim.x, D] (d scaling_factor), used when sf != null
    long total;             // B*N*D
    int N, D, L, mr, mc, reverse;
};

__global__ __launch_bounds__(kBlock) void affine_bwd_kernel(AffBwdArgs a) {
    __shared__ float gsf[kBwdMaxP];
    for (int i = threadIdx.x; i < a.D; i += kBlock) gsf[i] = 0.f;
    __syncthreads();
    const bool has_sf = a.sf != nullptr;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long)gridDim.x * kBlock) {
        const long b = e / a.L;
        const int er = (int)(e - b * a.L);
        const int n = er / a.D, d = er - n * a.D;
        const float keep = 1.f - mask_at(a.mask, a.mr, a.mc, n, d);
        const float2 p = *reinterpret_cast<const float2*>(a.nn + 2 * e);
        const float gzo = a.g_zout ? a.g_zout[e] : 0.f;
        const float gl = a.g_ldj ? a.g_ldj[b] : 0.f;
        const float zo = a.z_out[e];
        float f = 1.f, fc = 1.f, th = 0.f, u = 0.f, s;
        if (has_sf) {
            f = expf(a.sf[d]);
            fc = fmaxf(f, 1.f);
            u = p.x / fc;
            th = tanhf(u);
            s = th * f * keep;
        } else {
            s = p.x * keep;
        }
        const float t = p.y * keep;
        float gz, gt, gs;
        if (!a.reverse) {                       // z' = (z + t) e^s ; ldj += s
            const float es = expf(s);
            gz = gzo * es;
            gt = gz;
            gs = gzo * zo + gl;
        } else {                                // z' = z e^-s - t ; ldj -= s
            gz = gzo * expf(-s);
            gt = -gzo;
            gs = -gzo * (zo + t) - gl;
        }
        float g_sr;
        if (has_sf) {
            const float sech2 = 1.f - th * th;
            g_sr = gs * keep * f * sech2 / fc;
            // d s / d scaling_factor: f = e^sf, clamp(min=1) passes the gradient for f >= 1 (torch semantics)
            const float ds = (f >= 1.f) ? f * (th - u * sech2) : f * th;
            if (keep != 0.f) atomicAdd(&gsf[d], gs * keep * ds);
        } else {
            g_sr = gs * keep;
        }
        a.g_z[e] = gz;
        *reinterpret_cast<float2*>(a.g_nn + 2 * e) = make_float2(g_sr, gt * keep);
    }
    if (has_sf) flush_partials(gsf, a.D, a.partials);
}

// ---- ExtActNorm (activation_normalization.py:116-144) ------------------------------------------------
struct ExtBwdArgs {
    const float* z_out;
    const float* nn;        // [B,N,2D] = [bias | scales_raw]
    const float* pad;       // [B*N] nullable (weights the ldj only)
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* g_nn;
    long total;
    int N, D, L, reverse;
};
__global__ __launch_bounds__(kBlock) void ext_actnorm_bwd_kernel(ExtBwdArgs a) {
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long)gridDim.x * kBlock) {
        const long b = e / a.L;
        const long tok = e / a.D;
        const int d = (int)(e - tok * a.D);
        const float bias = a.nn[tok * 2 * a.D + d];
        const float th = tanhf(a.nn[tok * 2 * a.D + a.D + d]);
        const float gzo = a.g_zout ? a.g_zout[e] : 0.f;
        const float gl = (a.g_ldj ? a.g_ldj[b] : 0.f) * (a.pad ? a.pad[tok] : 1.f);
        const float zo = a.z_out[e];
        float gz, gb, gs;
        if (!a.reverse) {                       // z' = (z + bias) e^s ; ldj += s pad
            gz = gzo * expf(th);
            gb = gz;
            gs = gzo * zo + gl;
        } else {                                // z' = z e^-s - bias ; ldj -= s pad
            gz = gzo * expf(-th);
            gb = -gzo;
            gs = -gzo * (zo + bias) - gl;
        }
        a.g_z[e] = gz;
        a.g_nn[tok * 2 * a.D + d] = gb;
        a.g_nn[tok * 2 * a.D + a.D + d] = gs * (1.f - th * th);
    }
}

// ---- ActNorm (activation_normalization.py:24-48) ---------------------------------------------------
struct ActBwdArgs {
    const float* z_out;
    const float* bias;
    const float* scales;
    const float* pad;
    const float* length;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* partials;        // [gridDim.x, 2D]: d bias | d scales
    long total;
    int B, N, D, L, reverse;
};
__global__ __launch_bounds__(kBlock) void actnorm_bwd_kernel(ActBwdArgs a) {
    __shared__ float acc[kBwdMaxP];
    for (int i = threadIdx.x; i < 2 * a.D; i += kBlock) acc[i] = 0.f;
    __syncthreads();
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long)gridDim.x * kBlock) {
        const long tok = e / a.D;
        const int d = (int)(e - tok * a.D);
        const float pv = a.pad ? a.pad[tok] : 1.f;
        const float gzo = (a.g_zout ? a.g_zout[e] : 0.f) * pv;     // z' = (...) * pad
        const float sc = a.scales[d], bi = a.bias[d];
        float gz, gb, gs;
        if (!a.reverse) {                       // y = (z + b) e^sc
            const float es = expf(sc);
            gz = gzo * es;
            gb = gz;
            // (z + b) e^sc = z'/pad where pad = 1; padded positions have gzo = 0 anyway
            gs = gzo * (pv != 0.f ? a.z_out[e] / pv : 0.f);
        } else {                                // y = z e^-sc - b
            const float ems = expf(-sc);
            gz = gzo * ems;
            gb = -gzo;
            gs = -gzo * ((pv != 0.f ? a.z_out[e] / pv : 0.f) + bi);
        }
        a.g_z[e] = gz;
        atomicAdd(&acc[d], gb);
        atomicAdd(&acc[a.D + d], gs);
    }
    // log-det term: ldj += (+-sum_d scales) * len_b  ->  d scales[d] += +-sum_b g_ldj[b] len_b
    if (a.g_ldj) {
        float part = 0.f;
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
            float len;
            if (a.length) len = a.length[b];
            else if (a.pad) {
                len = 0.f;
                for (int n = 0; n < a.N; ++n) len += a.pad[b * a.N + n];
            } else len = (float)a.N;
            part += a.g_ldj[b] * len;
        }
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0 && part != 0.f) {
            const float sgn = a.reverse ? -1.f : 1.f;
            for (int d = 0; d < a.D; ++d) atomicAdd(&acc[a.D + d], sgn * part);
        }
    }
    flush_partials(acc, 2 * a.D, a.partials);
}

// ---- invertible 1x1 convolution (permutation_layers.py:106-136) -------------------------------------------
struct ConvBwdArgs {
    const float* x;
    const float* w;         // the matrix that was applied (W or W^-1), [D,D]
    const float* pad;
    const float* length;
    const float* g_zout;
    const float* g_ldj;
    float* g_x;
    float* partials;        // [gridDim.x, D*D + 1]: dW | d sldj
    long ntok;
    int B, N, D, reverse;
};
__global__ __launch_bounds__(kBlock) void invconv_bwd_kernel(ConvBwdArgs a) {
    __shared__ float acc[kBwdMaxP];
    const int D = a.D, P = D * D + 1;
    for (int i = threadIdx.x; i < P; i += kBlock) acc[i] = 0.f;
    __syncthreads();
    // one lane per (token, input channel i): g_x[t,i] = sum_j g[t,j] W[i,j];  dW[i,j] += x[t,i] g[t,j]
    const long total = a.ntok * D;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const long t = e / D;
        const int i = (int)(e - t * D);
        const float pv = a.pad ? a.pad[t] : 1.f;
        const float xi = a.x[e];
        float gx = 0.f;
        for (int j = 0; j < D; ++j) {
            const float g = (a.g_zout ? a.g_zout[t * D + j] : 0.f) * pv;
            gx = fmaf(g, a.w[i * D + j], gx);
            if (g != 0.f) atomicAdd(&acc[i * D + j], xi * g);
        }
        a.g_x[e] = gx;
    }
    if (a.g_ldj) {
        float part = 0.f;
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock)
            part += a.g_ldj[b] * (a.length ? a.length[b] : (float)a.N);
        part = wave_sum(part);
        if ((threadIdx.x & 63) == 0 && part != 0.f) atomicAdd(&acc[D * D], a.reverse ? -part : part);
    }
    flush_partials(acc, P, a.partials);
}

// ---- logistic log-prob / NLL (distributions.py:129-163; set_modeling/task.py:96-118) ------------------------
// d/dx [-(softplus(v) + softplus(-v) + log sigma)] = -tanh(v/2) / sigma,  v = (x - mu)/sigma
__global__ __launch_bounds__(kBlock) void logistic_log_prob_bwd_kernel(const float* x, const float* g_out,
                                                                       float* g_x, long n, float mu,
                                                                       float sigma) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float v = (x[i] - mu) / sigma;
        g_x[i] = g_out[i] * (-tanhf(0.5f * v) / sigma);
    }
}

struct NllBwdArgs {
    const float* z;
    const float* pad;
    const float* length;
    const float* g_nll;     // [B]
    float* g_z;
    float* g_ldj;           // [B] nullable
    long total;
    int B, N, D, L;
    float sigma;
};
// nll_b = (-ldj_b - sum logp(z) pad) / len_b
__global__ __launch_bounds__(kBlock) void prior_nll_bwd_kernel(NllBwdArgs a) {
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long)gridDim.x * kBlock) {
        const long b = e / a.L;
        const long tok = e / a.D;
        const float len = a.length ? a.length[b] : (float)a.N;
        const float pv = a.pad ? a.pad[tok] : 1.f;
        const float v = a.z[e] / a.sigma;
        a.g_z[e] = a.g_nll[b] / len * pv * (tanhf(0.5f * v) / a.sigma);
    }
    if (a.g_ldj)
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock)
            a.g_ldj[b] = -a.g_nll[b] / (a.length ? a.length[b] : (float)a.N);
}

// ---- sigmoid / logit flow (sigmoid_layer.py:24-47) -----------------------------------------------------------
__global__ __launch_bounds__(kBlock) void sigmoid_flow_bwd_kernel(const float* z_in, const float* g_zout,
                                                                  const float* g_ldj, float* g_z, long total,
                                                                  int L, int reverse, float alpha) {
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const float gl = g_ldj ? g_ldj[e / L] : 0.f;
        const float gzo = g_zout ? g_zout[e] : 0.f;
        const float x = z_in[e];
        float g;
        if (!reverse) {                         // y = sigmoid(x), ldj += -x - 2 softplus(-x)
            const float s = 1.f / (1.f + expf(-x));
            g = gzo * s * (1.f - s) + gl * (1.f - 2.f * s);
        } else {                                // u = x(1-a)+a/2 ; y = log u - log(1-u) ; ldj += -log u - log(1-u) + c
            const float u = x * (1.f - alpha) + alpha * 0.5f;
            const float inv = 1.f / (u * (1.f - u));
            g = (gzo * inv + gl * (2.f * u - 1.f) * inv) * (1.f - alpha);
        }
        g_z[e] = g;
    }
}

// ---- static-API split forms of the affine coupling (coupling_layer.py:76-98) ---------------------------------
__global__ __launch_bounds__(kBlock) void affine_params_bwd_kernel(const float* nn, const float* sf, const float* mask,
                                                                   int mr, int mc, const float* g_s, const float* g_t,
                                                                   float* g_nn, float* partials, long total, int N, int D) {
    __shared__ float gsf[kBwdMaxP];
    for (int i = threadIdx.x; i < D; i += kBlock) gsf[i] = 0.f;
    __syncthreads();
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const int d = (int)(e % D);
        const int n = (int)((e / D) % N);
        const float keep = 1.f - mask_at(mask, mr, mc, n, d);
        const float gs = (g_s ? g_s[e] : 0.f) * keep;
        const float gt = (g_t ? g_t[e] : 0.f) * keep;
        float g_sr = gs;
        if (sf) {
            const float f = expf(sf[d]);
            const float fc = fmaxf(f, 1.f);
            const float u = nn[2 * e] / fc;
            const float th = tanhf(u);
            const float sech2 = 1.f - th * th;
            g_sr = gs * f * sech2 / fc;
            if (gs != 0.f) atomicAdd(&gsf[d], gs * ((f >= 1.f) ? f * (th - u * sech2) : f * th));
        }
        *reinterpret_cast<float2*>(g_nn + 2 * e) = make_float2(g_sr, gt);
    }
    if (sf) flush_partials(gsf, D, partials);
}

__global__ __launch_bounds__(kBlock) void affine_transform_bwd_kernel(const float* z_out, const float* s, const float* t,
                                                                      const float* g_zout, const float* g_ldj,
                                                                      float* g_z, float* g_s, float* g_t, long total,
                                                                      int L, int reverse) {
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const float gzo = g_zout ? g_zout[e] : 0.f;
        const float gl = g_ldj ? g_ldj[e / L] : 0.f;
        if (!reverse) {
            const float gz = gzo * expf(s[e]);
            g_z[e] = gz;
            g_t[e] = gz;
            g_s[e] = gzo * z_out[e] + gl;
        } else {
            g_z[e] = gzo * expf(-s[e]);
            g_t[e] = -gzo;
            g_s[e] = -gzo * (z_out[e] + t[e]) - gl;
        }
    }
}

static inline int bwd_grid(long n) {
    return (int)std::min<long>(std::max<long>((n + kBlock - 1) / kBlock, 1), kBwdGrid);
}

static int reduce_partials(const float* partials, int rows, int P, float* out, hipStream_t st) {
    CNF_LAUNCH(reduce_partials_kernel, dim3(P), dim3(kBlock), 0, st, partials, rows, P, out);
    return CNF_OK;
}

}  // namespace cnf

using namespace cnf;

extern "C" {

/* floats the caller must provide as `workspace` to the backward entry points that return parameter
 * gradients: kBwdGrid rows of `param_count` partial sums + one reduced row */
int64_t cnf_bwd_workspace_floats(int param_count) { return (int64_t)(kBwdGrid + 1) * param_count; }

int cnf_affine_coupling_bwd(const float* z_out, const float* nn_out, const float* scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const float* g_zout, const float* g_ldj,
                            float* g_z, float* g_nn, float* g_scaling_factor, float* workspace,
                            int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && nn_out && g_z && g_nn, "cnf_affine_coupling_bwd: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && D <= kBwdMaxP, "cnf_affine_coupling_bwd: bad shape");
    CNF_REQUIRE(!scaling_factor || (g_scaling_factor && workspace), "cnf_affine_coupling_bwd: scaling_factor needs g_scaling_factor and workspace");
    if (B == 0) return CNF_OK;
    if (!mask) { mask_rows = 1; mask_cols = D; }
    if (mask_rows > N) mask_rows = N;
    AffBwdArgs a{z_out, nn_out, scaling_factor, mask, g_zout, g_ldj, g_z, g_nn, workspace, (long)B * N * D,
                 N, D, N * D, mask_rows, mask_cols, reverse};
    const int grid = bwd_grid(a.total);
    CNF_LAUNCH(affine_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, a);
    if (scaling_factor) reduce_partials(workspace, grid, D, g_scaling_factor, (hipStream_t)stream);
    return launch_status("cnf_affine_coupling_bwd");
}

int cnf_affine_params_bwd(const float* nn_out, const float* scaling_factor, const float* mask, int mask_rows, int mask_cols,
                          const float* g_s, const float* g_t, float* g_nn, float* g_scaling_factor, float* workspace,
                          int B, int N, int D, cnf_stream_t stream) {
    CNF_REQUIRE(nn_out && g_nn, "cnf_affine_params_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && D <= kBwdMaxP, "cnf_affine_params_bwd: bad shape");
    CNF_REQUIRE(!scaling_factor || (g_scaling_factor && workspace), "cnf_affine_params_bwd: scaling_factor needs g_scaling_factor and workspace");
    if (mask && mask_rows > N) mask_rows = N;
    const long total = (long)B * N * D;
    const int grid = bwd_grid(total);
    CNF_LAUNCH(affine_params_bwd_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream, nn_out, scaling_factor, mask,
                       mask_rows, mask_cols, g_s, g_t, g_nn, workspace, total, N, D);
    if (scaling_factor) reduce_partials(workspace, grid, D, g_scaling_factor, (hipStream_t)stream);
    return launch_status("cnf_affine_params_bwd");
}

int cnf_affine_transform_bwd(const float* z_out, const float* s, const float* t, const float* g_zout, const float* g_ldj,
                             float* g_z, float* g_s, float* g_t, int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && s && t && g_z && g_s && g_t, "cnf_affine_transform_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_affine_transform_bwd: bad shape");
    const long total = (long)B * N * D;
    CNF_LAUNCH(affine_transform_bwd_kernel, dim3(bwd_grid(total)), dim3(kBlock), 0, (hipStream_t)stream, z_out, s, t,
                       g_zout, g_ldj, g_z, g_s, g_t, total, N * D, reverse);
    return launch_status("cnf_affine_transform_bwd");
}

int cnf_ext_actnorm_bwd(const float* z_out, const float* nn_out, const float* pad,
                        const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                        int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && nn_out && g_z && g_nn, "cnf_ext_actnorm_bwd: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_ext_actnorm_bwd: bad shape");
    if (B == 0) return CNF_OK;
    ExtBwdArgs a{z_out, nn_out, pad, g_zout, g_ldj, g_z, g_nn, (long)B * N * D, N, D, N * D, reverse};
    CNF_LAUNCH(ext_actnorm_bwd_kernel, dim3(bwd_grid(a.total)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return launch_status("cnf_ext_actnorm_bwd");
}

int cnf_actnorm_bwd(const float* z_out, const float* bias, const float* scales,
                    const float* pad, const float* length, const float* g_zout, const float* g_ldj,
                    float* g_z, float* g_bias, float* g_scales, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && bias && scales && g_z && g_bias && g_scales && workspace, "cnf_actnorm_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && 2 * D <= kBwdMaxP, "cnf_actnorm_bwd: bad shape");
    ActBwdArgs a{z_out, bias, scales, pad, length, g_zout, g_ldj, g_z, workspace, (long)B * N * D, B, N, D, N * D, reverse};
    const int grid = bwd_grid(a.total);
    hipStream_t st = (hipStream_t)stream;
    CNF_LAUNCH(actnorm_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, a);
    // partial rows are [d bias (D) | d scales (D)]
    // straight into the two gradient tensors (two device-to-device copies cost ~10 us each)
    CNF_LAUNCH(mix_reduce_partials_kernel, dim3(2 * D), dim3(kBlock), 0, st, workspace, grid, 2 * D, g_bias, g_scales, D);
    return launch_status("cnf_actnorm_bwd");
}

int cnf_invconv_bwd(const float* x, const float* weight, const float* pad, const float* length,
                    const float* g_zout, const float* g_ldj,
                    float* g_x, float* g_weight, float* g_sldj, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(x && weight && g_x && g_weight && g_sldj && workspace, "cnf_invconv_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && D * D + 1 <= kBwdMaxP, "cnf_invconv_bwd: bad shape");
    ConvBwdArgs a{x, weight, pad, length, g_zout, g_ldj, g_x, workspace, (long)B * N, B, N, D, reverse};
    const int P = D * D + 1;
    const int grid = bwd_grid(a.ntok * D);
    hipStream_t st = (hipStream_t)stream;
    CNF_LAUNCH(invconv_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, a);
    CNF_LAUNCH(mix_reduce_partials_kernel, dim3(P), dim3(kBlock), 0, st, workspace, grid, P, g_weight, g_sldj, D * D);
    return launch_status("cnf_invconv_bwd");
}

int cnf_logistic_log_prob_bwd(const float* x, const float* g_logp, float* g_x, int64_t n, float mu, float sigma,
                              cnf_stream_t stream) {
    CNF_REQUIRE(x && g_logp && g_x && n >= 0, "cnf_logistic_log_prob_bwd: bad argument");
    if (n == 0) return CNF_OK;
    CNF_LAUNCH(logistic_log_prob_bwd_kernel, dim3(bwd_grid(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       x, g_logp, g_x, (long)n, mu, sigma);
    return launch_status("cnf_logistic_log_prob_bwd");
}

int cnf_prior_nll_bwd(const float* z, const float* pad, const float* length, const float* g_nll,
                      float* g_z, float* g_ldj, int B, int N, int D, float sigma, cnf_stream_t stream) {
    CNF_REQUIRE(z && g_nll && g_z, "cnf_prior_nll_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_prior_nll_bwd: bad shape");
    NllBwdArgs a{z, pad, length, g_nll, g_z, g_ldj, (long)B * N * D, B, N, D, N * D, sigma};
    CNF_LAUNCH(prior_nll_bwd_kernel, dim3(bwd_grid(a.total)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return launch_status("cnf_prior_nll_bwd");
}

int cnf_sigmoid_flow_bwd(const float* z_in, const float* g_zout, const float* g_ldj, float* g_z,
                         int B, int L, int reverse, float alpha, cnf_stream_t stream) {
    CNF_REQUIRE(z_in && g_z, "cnf_sigmoid_flow_bwd: null tensor");
    CNF_REQUIRE(B > 0 && L > 0, "cnf_sigmoid_flow_bwd: bad shape");
    const long total = (long)B * L;
    CNF_LAUNCH(sigmoid_flow_bwd_kernel, dim3(bwd_grid(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       z_in, g_zout, g_ldj, g_z, total, L, reverse, alpha);
    return launch_status("cnf_sigmoid_flow_bwd");
}

}  // extern "C"
