// Backward of the logistic-mixture CDF coupling FORWARD transform (mixture_cdf_layer.py:95-123,145-180),
// fp64 like the forward kernel.  The reference never differentiates the inverse (its encoders use affine
// couplings for that reason, linear_encoding.py:233), so only the forward direction has a backward.
//
// Per transformed element (x, parameters t, log_s, log_pi[K], mu[K], ls[K]):
//   u = sum_k pi_k sigma_k,  pdf = sum_k pi_k p_k,  p_k = sigma_k (1 - sigma_k) e^{-ls_k},  z_k = (x - mu_k) e^{-ls_k}
//   y = log u - log(1-u),  z' = (y + t) e^{log_s},  ldj = log_s - log u - log(1-u) + log pdf (+ reg)
// Upstream: g_zout (d loss / d z'), g_ldj (per sample).  Outputs: g_z, g_nn (raw subnet output, through
// the fp32 tanh bounds and the channel mask), g_scaling_factor [D], g_mixture_scaling_factor [D,K].
#include "cnf_common.h"

#include <algorithm>

namespace cnf {

constexpr int kMixBwdMaxP = 1024;
constexpr int kMixBwdGrid = 1024;
constexpr double kLn10b = 2.302585092994045684;

// A workgroup's parameter-gradient sums: the lanes meet in LDS words, many lanes per word and in no fixed order — so the words
// are 64-bit FIXED-POINT sums (2^-32 units, integer atomics): integer addition is associative, the sum does not depend on the
// order, and the gradients of this file are bit-reproducible like the rest of the library's (rounds 1-4 added fp32 here with
// atomicAdd: run-to-run differences in the last bits).  Range +-2^31 per workgroup and parameter, resolution 2.3e-10: a
// workgroup sees fewer than 2^21 elements (B N D < 2^31 over 1024 workgroups), so terms below 2^10 cannot wrap its word; a
// larger term, +-inf and NaN (a diverged element) go to the word's fp64 escape twin (cnf_common.h: fix_pair_add) and the
// gradient comes out as the reference's floating-point sum would — NaN for a NaN term, not a finite wrong number.
constexpr double kGradTermMax = 1024.0;
__device__ __forceinline__ void fix_add(long long* word, double* big, double v) {
    fix_pair_add(reinterpret_cast<unsigned long long*>(word), big, v, kGradTermMax);
}
__device__ __forceinline__ float fix_value(long long word, double big) { return (float)fix_pair_value(word, big); }

struct MixBwdArgs {
    const float* z;
    const float* nn;
    const float* sf;
    const float* msf;
    const float* mask;
    const float* pad;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* g_nn;            // pre-zeroed by the host side (zero_fill_async below)
    float* partials;        // [gridDim.x, D + D*K]
    // split (static API) form: fp64 parameter tensors in, fp64 gradients out
    const double* z64;
    const double* p_t;
    const double* p_log_s;
    const double* p_log_pi;
    const double* p_mu;
    const double* p_ls;
    const double* g_zout64;
    const double* g_ldj64;
    double* g_z64;
    double* g_t;
    double* g_log_s;
    double* g_log_pi;
    double* g_mu;
    double* g_ls;
    long total;             // B*N*D
    int N, D, K, P, L, mr, mc;
    int pad_in_transform, pad_output, use_reg;
    double reg_max, reg_factor;
};

__device__ __forceinline__ float bound_f(float raw, const float* fac_ptr) {      // tanh bound in fp32 (:156-162)
    if (!fac_ptr) return raw;
    const float f = expf(fac_ptr[0]);
    return tanhf(raw / fmaxf(f, 1.f)) * f;
}
// d bound / d raw and d bound / d factor-parameter (f = e^sf; clamp(min=1) passes the gradient for f >= 1)
__device__ __forceinline__ void bound_grads(float raw, const float* fac_ptr, double& d_raw, double& d_sf) {
    if (!fac_ptr) { d_raw = 1.0; d_sf = 0.0; return; }
    const float f = expf(fac_ptr[0]);
    const float fc = fmaxf(f, 1.f);
    const float u = raw / fc;
    const float th = tanhf(u);
    const float sech2 = 1.f - th * th;
    d_raw = (double)(f * sech2 / fc);
    d_sf = (double)((f >= 1.f) ? f * (th - u * sech2) : f * th);
}

template <bool SPLIT>
__global__ __launch_bounds__(kBlock) void mixture_fwd_bwd_kernel(MixBwdArgs a) {
    __shared__ long long acc[kMixBwdMaxP];
    __shared__ double accbig[kMixBwdMaxP];
    const int PP = a.D + a.D * a.K;
    for (int i = threadIdx.x; i < PP; i += kBlock) {
        acc[i] = 0;
        accbig[i] = 0.0;
    }
    __syncthreads();
    const int K = a.K;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < a.total; e += (long)gridDim.x * kBlock) {
        const long b = e / a.L;
        const int er = (int)(e - b * a.L);
        const int n = er / a.D, d = er - n * a.D;
        const long tok = e / a.D;
        const float m = mask_at(a.mask, a.mr, a.mc, n, d);
        const float pv = a.pad ? a.pad[tok] : 1.f;
        const float change = (1.f - m) * (a.pad_in_transform ? pv : 1.f);
        const float outscale = a.pad_output ? pv : 1.f;
        const double gzo = SPLIT ? (a.g_zout64 ? a.g_zout64[e] : 0.0)
                                 : (double)((a.g_zout ? a.g_zout[e] : 0.f) * outscale);
        if (change == 0.f) {                    // copied through: z' = z * outscale, no parameter dependence
            if (SPLIT) a.g_z64[e] = gzo;
            else a.g_z[e] = (float)gzo;
            continue;
        }
        const double gl = SPLIT ? (a.g_ldj64 ? a.g_ldj64[b] : 0.0) : (a.g_ldj ? (double)a.g_ldj[b] : 0.0);
        const float* row = SPLIT ? nullptr : a.nn + (size_t)e * a.P;
        float* grow = SPLIT ? nullptr : a.g_nn + (size_t)e * a.P;
        const float* sf_d = (!SPLIT && a.sf) ? a.sf + d : nullptr;
        const double x = SPLIT ? a.z64[e] : (double)a.z[e];
        const double t = SPLIT ? a.p_t[e] : (double)row[0];
        const double log_s = SPLIT ? a.p_log_s[e] : (double)bound_f(row[1], sf_d);
        auto lp_at = [&](int k) -> double { return SPLIT ? a.p_log_pi[(size_t)e * K + k] : (double)row[2 + k]; };
        auto mu_at = [&](int k) -> double { return SPLIT ? a.p_mu[(size_t)e * K + k] : (double)row[2 + K + k]; };
        auto ls_at = [&](int k) -> double {
            return SPLIT ? a.p_ls[(size_t)e * K + k]
                         : (double)bound_f(row[2 + 2 * K + k], a.msf ? a.msf + (size_t)d * K + k : nullptr);
        };
        // pass 1: softmax normaliser and the mixture sums
        double mx = -INFINITY;
        for (int k = 0; k < K; ++k) mx = fmax(mx, lp_at(k));
        double se = 0.0, cdf = 0.0, pdf = 0.0, dpdf = 0.0;
        for (int k = 0; k < K; ++k) {
            const double w = exp(lp_at(k) - mx);
            const double ls = ls_at(k);
            const double inv_s = exp(-ls);
            const double zk = (x - mu_at(k)) * inv_s;
            const double ee = exp(-fabs(zk));
            const double r = 1.0 / (1.0 + ee);
            const double sig = zk >= 0.0 ? r : ee * r;
            const double pk = ee * r * r * inv_s;                // sigma (1 - sigma) / s
            se += w;
            cdf += w * sig;
            pdf += w * pk;
            dpdf += w * pk * (1.0 - 2.0 * sig) * inv_s;           // d p_k / d x
        }
        const double u = cdf / se;
        const double pdf_n = pdf / se;
        const double a_s = exp(log_s);
        // torch.clamp keeps a NaN (fmax returns its other operand): a diverged element gives NaN gradients, as autograd does
        const double uc = u != u ? u : fmax(u, 1e-22), u1c = u != u ? u : fmax(1.0 - u, 1e-22);
        const double lu = log(uc), l1u = log(u1c);
        const double dlu = u > 1e-22 ? 1.0 / u : (u != u ? u : 0.0);               // d safe_log(u) / du
        const double dl1u = (1.0 - u) > 1e-22 ? -1.0 / (1.0 - u) : (u != u ? u : 0.0);   // d safe_log(1-u) / du
        const double zt = ((lu - l1u) + t) * a_s;
        double g_u = gzo * a_s * (dlu - dl1u) + gl * (-dlu - dl1u);
        if (a.use_reg) {
            const double r1 = lu / kLn10b, r2 = l1u / kLn10b;
            double dreg = 0.0;
            if (r1 <= -a.reg_max) dreg += dlu / kLn10b;
            if (r2 <= -a.reg_max) dreg += dl1u / kLn10b;
            g_u += gl * a.reg_factor * dreg;
        }
        const double inv_pdf = pdf_n > 1e-290 ? 1.0 / pdf_n : 0.0;
        const double g_x = g_u * pdf_n + gl * (dpdf / se) * inv_pdf;
        const double g_logs = gzo * zt + gl;
        if (SPLIT) {
            a.g_z64[e] = g_x;
            a.g_t[e] = gzo * a_s;
            a.g_log_s[e] = g_logs;
        } else {
            a.g_z[e] = (float)g_x;
            grow[0] = (float)(gzo * a_s);
            double d_raw, d_sf;
            bound_grads(row[1], sf_d, d_raw, d_sf);
            grow[1] = (float)(g_logs * d_raw);
            if (sf_d) fix_add(&acc[d], &accbig[d], g_logs * d_sf);
        }
        // pass 2: per-mixture parameter gradients
        for (int k = 0; k < K; ++k) {
            const double pi = exp(lp_at(k) - mx) / se;
            const float* msf_k = (!SPLIT && a.msf) ? a.msf + (size_t)d * K + k : nullptr;
            const double ls = ls_at(k);
            const double inv_s = exp(-ls);
            const double zk = (x - mu_at(k)) * inv_s;
            const double ee = exp(-fabs(zk));
            const double r = 1.0 / (1.0 + ee);
            const double sig = zk >= 0.0 ? r : ee * r;
            const double s1s = ee * r * r;                         // sigma (1 - sigma)
            const double pk = s1s * inv_s;
            const double resp = pi * pk * inv_pdf;                 // pi_k p_k / pdf
            const double g_lp = g_u * pi * (sig - u) + gl * (resp - pi);                       // logits
            const double g_mu = g_u * (-pi * pk) + gl * (-resp * (1.0 - 2.0 * sig) * inv_s);   // mean
            const double g_ls = g_u * (-pi * zk * s1s) + gl * (resp * (-1.0 - zk * (1.0 - 2.0 * sig)));   // log-scale
            if (SPLIT) {
                a.g_log_pi[(size_t)e * K + k] = g_lp;
                a.g_mu[(size_t)e * K + k] = g_mu;
                a.g_ls[(size_t)e * K + k] = g_ls;
            } else {
                grow[2 + k] = (float)g_lp;
                grow[2 + K + k] = (float)g_mu;
                double d_raw, d_sf;                                  // through the tanh bound
                bound_grads(row[2 + 2 * K + k], msf_k, d_raw, d_sf);
                grow[2 + 2 * K + k] = (float)(g_ls * d_raw);
                if (msf_k) fix_add(&acc[a.D + d * K + k], &accbig[a.D + d * K + k], g_ls * d_sf);
            }
        }
    }
    __syncthreads();
    if (!SPLIT)
        for (int i = threadIdx.x; i < PP; i += kBlock) a.partials[(size_t)blockIdx.x * PP + i] = fix_value(acc[i], accbig[i]);
}

// d(get_mixt_params) (:145-180): five fp64 upstream gradients -> g_nn (fp32) through mask and tanh bounds
struct MixParamsBwdArgs {
    const float* nn;
    const float* sf;
    const float* msf;
    const float* mask;
    const double* g_t;
    const double* g_log_s;
    const double* g_log_pi;
    const double* g_mu;
    const double* g_ls;
    float* g_nn;
    float* partials;
    long nelem;
    int N, D, K, mr, mc;
};
__global__ __launch_bounds__(kBlock) void mixture_params_bwd_kernel(MixParamsBwdArgs a) {
    __shared__ long long acc[kMixBwdMaxP];
    __shared__ double accbig[kMixBwdMaxP];
    const int K = a.K, P = 2 + 3 * K, PP = a.D + a.D * K;
    for (int i = threadIdx.x; i < PP; i += kBlock) {
        acc[i] = 0;
        accbig[i] = 0.0;
    }
    __syncthreads();
    const long total = a.nelem * P;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long elem = i / P;
        const int j = (int)(i - elem * P);
        const int d = (int)(elem % a.D);
        const int n = (int)((elem / a.D) % a.N);
        const double keep = (double)(1.f - mask_at(a.mask, a.mr, a.mc, n, d));
        double g, d_raw = 1.0, d_sf = 0.0;
        int slot = -1;
        if (j == 0) g = a.g_t ? a.g_t[elem] : 0.0;
        else if (j == 1) {
            g = a.g_log_s ? a.g_log_s[elem] : 0.0;
            bound_grads(a.nn[i], a.sf ? a.sf + d : nullptr, d_raw, d_sf);
            if (a.sf) slot = d;
        } else if (j < 2 + K) g = a.g_log_pi ? a.g_log_pi[elem * K + (j - 2)] : 0.0;
        else if (j < 2 + 2 * K) g = a.g_mu ? a.g_mu[elem * K + (j - 2 - K)] : 0.0;
        else {
            const int k = j - 2 - 2 * K;
            g = a.g_ls ? a.g_ls[elem * K + k] : 0.0;
            bound_grads(a.nn[i], a.msf ? a.msf + (size_t)d * K + k : nullptr, d_raw, d_sf);
            if (a.msf) slot = a.D + d * K + k;
        }
        g *= keep;
        a.g_nn[i] = (float)(g * d_raw);
        if (slot >= 0 && g != 0.0) fix_add(&acc[slot], &accbig[slot], g * d_sf);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PP; i += kBlock) a.partials[(size_t)blockIdx.x * PP + i] = fix_value(acc[i], accbig[i]);
}

// partials [nrows, P] -> column sums in fp64, fixed order; columns [0, split) go to out_a, the rest to out_b (either may
// be null): the parameter-gradient tensors are written directly (two device-to-device copies cost ~10 us each)
__global__ __launch_bounds__(kBlock) void mix_reduce_partials_kernel(const float* partials, int nrows, int P, float* out_a,
                                                                     float* out_b, int split) {
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
        if (p < split) {
            if (out_a) out_a[p] = (float)t;
        } else if (out_b) {
            out_b[p - split] = (float)t;
        }
    }
}

}  // namespace cnf

using namespace cnf;

extern "C" {

int cnf_mixture_coupling_bwd(const float* z, const float* nn_out,
                             const float* scaling_factor, const float* mixture_scaling_factor,
                             const float* mask, int mask_rows, int mask_cols,
                             const float* pad, int pad_in_transform, int pad_output,
                             const float* g_zout, const float* g_ldj,
                             float* g_z, float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor,
                             float* workspace,
                             int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                             cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && g_z && g_nn && workspace, "cnf_mixture_coupling_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && K > 0, "cnf_mixture_coupling_bwd: bad shape");
    CNF_REQUIRE(D + D * K <= kMixBwdMaxP, "cnf_mixture_coupling_bwd: D*(K+1)=%d exceeds %d", D + D * K, kMixBwdMaxP);
    CNF_REQUIRE(!scaling_factor || g_scaling_factor, "cnf_mixture_coupling_bwd: g_scaling_factor missing");
    CNF_REQUIRE(!mixture_scaling_factor || g_mixture_scaling_factor, "cnf_mixture_coupling_bwd: g_mixture_scaling_factor missing");
    if (!mask) { mask_rows = 1; mask_cols = D; }
    if (mask_rows > N) mask_rows = N;
    hipStream_t st = (hipStream_t)stream;
    MixBwdArgs a = {};
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor; a.mask = mask; a.pad = pad;
    a.g_zout = g_zout; a.g_ldj = g_ldj; a.g_z = g_z; a.g_nn = g_nn; a.partials = workspace;
    a.total = (long)B * N * D; a.N = N; a.D = D; a.K = K; a.P = 2 + 3 * K; a.L = N * D; a.mr = mask_rows; a.mc = mask_cols;
    a.pad_in_transform = pad ? pad_in_transform : 0;
    a.pad_output = pad ? pad_output : 0;
    a.use_reg = (reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    // parameters of untransformed elements get no gradient
    zero_fill_async(g_nn, sizeof(float) * (size_t)a.total * a.P, st);
    const int grid = (int)std::min<long>(std::max<long>((a.total + kBlock - 1) / kBlock, 1), kMixBwdGrid);
    CNF_LAUNCH((mixture_fwd_bwd_kernel<false>), dim3(grid), dim3(kBlock), 0, st, a);
    const int PP = D + D * K;
    CNF_LAUNCH(mix_reduce_partials_kernel, dim3(PP), dim3(kBlock), 0, st, workspace, grid, PP,
               scaling_factor ? g_scaling_factor : nullptr, mixture_scaling_factor ? g_mixture_scaling_factor : nullptr, D);
    return launch_status("cnf_mixture_coupling_bwd");
}

int cnf_mixture_transform_bwd(const double* z, const double* t, const double* log_s, const double* log_pi,
                              const double* mixt_t, const double* mixt_log_s,
                              const float* mask, int mask_rows, int mask_cols, const float* pad,
                              const double* g_zout, const double* g_ldj,
                              double* g_z, double* g_t, double* g_log_s, double* g_log_pi, double* g_mixt_t,
                              double* g_mixt_log_s,
                              int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                              cnf_stream_t stream) {
    CNF_REQUIRE(z && t && log_s && log_pi && mixt_t && mixt_log_s && g_z && g_t && g_log_s && g_log_pi && g_mixt_t && g_mixt_log_s,
                "cnf_mixture_transform_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && K > 0 && D + D * K <= kMixBwdMaxP, "cnf_mixture_transform_bwd: bad shape");
    if (!mask) { mask_rows = 1; mask_cols = D; }
    if (mask_rows > N) mask_rows = N;
    hipStream_t st = (hipStream_t)stream;
    MixBwdArgs a = {};
    a.z64 = z; a.p_t = t; a.p_log_s = log_s; a.p_log_pi = log_pi; a.p_mu = mixt_t; a.p_ls = mixt_log_s;
    a.mask = mask; a.pad = pad; a.g_zout64 = g_zout; a.g_ldj64 = g_ldj;
    a.g_z64 = g_z; a.g_t = g_t; a.g_log_s = g_log_s; a.g_log_pi = g_log_pi; a.g_mu = g_mixt_t; a.g_ls = g_mixt_log_s;
    a.total = (long)B * N * D; a.N = N; a.D = D; a.K = K; a.P = 2 + 3 * K; a.L = N * D; a.mr = mask_rows; a.mc = mask_cols;
    a.pad_in_transform = pad ? 1 : 0;
    a.pad_output = 0;
    a.use_reg = (reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    const size_t n = (size_t)a.total;
    zero_fill_async(g_t, sizeof(double) * n, st);
    zero_fill_async(g_log_s, sizeof(double) * n, st);
    zero_fill_async(g_log_pi, sizeof(double) * n * K, st);
    zero_fill_async(g_mixt_t, sizeof(double) * n * K, st);
    zero_fill_async(g_mixt_log_s, sizeof(double) * n * K, st);
    const int grid = (int)std::min<long>(std::max<long>((a.total + kBlock - 1) / kBlock, 1), kMixBwdGrid);
    CNF_LAUNCH((mixture_fwd_bwd_kernel<true>), dim3(grid), dim3(kBlock), 0, st, a);
    return launch_status("cnf_mixture_transform_bwd");
}

int cnf_mixture_params_bwd(const float* nn_out, const float* scaling_factor, const float* mixture_scaling_factor,
                           const float* mask, int mask_rows, int mask_cols,
                           const double* g_t, const double* g_log_s, const double* g_log_pi, const double* g_mixt_t,
                           const double* g_mixt_log_s,
                           float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor, float* workspace,
                           int B, int N, int D, int K, cnf_stream_t stream) {
    CNF_REQUIRE(nn_out && g_nn && workspace, "cnf_mixture_params_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && K > 0 && D + D * K <= kMixBwdMaxP, "cnf_mixture_params_bwd: bad shape");
    if (mask && mask_rows > N) mask_rows = N;
    hipStream_t st = (hipStream_t)stream;
    MixParamsBwdArgs a{nn_out, scaling_factor, mixture_scaling_factor, mask, g_t, g_log_s, g_log_pi, g_mixt_t, g_mixt_log_s,
                       g_nn, workspace, (long)B * N * D, N, D, K, mask_rows, mask_cols};
    const long total = a.nelem * (2 + 3 * K);
    const int grid = (int)std::min<long>(std::max<long>((total + kBlock - 1) / kBlock, 1), kMixBwdGrid);
    CNF_LAUNCH(mixture_params_bwd_kernel, dim3(grid), dim3(kBlock), 0, st, a);
    const int PP = D + D * K;
    CNF_LAUNCH(mix_reduce_partials_kernel, dim3(PP), dim3(kBlock), 0, st, workspace, grid, PP,
               scaling_factor ? g_scaling_factor : nullptr, mixture_scaling_factor ? g_mixture_scaling_factor : nullptr, D);
    return launch_status("cnf_mixture_params_bwd");
}

}  // extern "C"
