// Logistic-mixture CDF coupling (Flow++ style) — forward, bisection inverse, log-det — in fp64,
// following mixture_cdf_layer.py:95-180, 197-276 of the reference.
//
// Work item = one TRANSFORMED element (b, n, d): masked channels are never evaluated (the
// reference evaluates them on all-zero parameters and then overwrites the result, :137-138), they
// are copied through.  One wave owns whole samples; the per-sample log-det is summed in fp64
// through the wave's private LDS strip, in a fixed order.
//
// Arithmetic: the reference works in log space and exponentiates immediately
// (`mixture_log_cdf(...).exp()`, :105).  Here the mixture CDF and PDF are accumulated directly,
//     u   = sum_k w_k sigma(z_k) / sum_k w_k,        w_k = exp(log_pi_k - max log_pi)
//     pdf = sum_k w_k e^{-ls_k} sigma(z_k)(1 - sigma(z_k)) / sum_k w_k,
// which is the same quantity to fp64 rounding (3 exp + 1 divide per mixture instead of ~6
// transcendentals); if the PDF sum underflows, the element falls back to the log-space form.
#include "cnf_mixture.h"

#include <algorithm>
#include <atomic>

namespace cnf {

// Per-element parameter access.  FUSED: read the fp32 subnet row and apply the tanh bounds in fp32
// exactly like get_mixt_params (:156-162) before widening; SPLIT: read the fp64 tensors.
template <bool SPLIT>
struct ParamRow {
    // plain values only: holding a reference to the kernel-argument block would force a stack copy
    const float* row;     // fused: start of this channel's P-block of nn_out
    const float* sf_d;    // fused: &scaling_factor[d] or null
    const float* msf_d;   // fused: &mixture_scaling_factor[d*K] or null
    const double* pt;     // split: &t[elem], &log_s[elem], &log_pi[elem*K] ...
    const double* pls;
    const double* ppi;
    const double* pmu;
    const double* pmls;
    int K;
    __device__ __forceinline__ ParamRow(const MixArgs& a, size_t elem, int d) {
        K = a.K;
        if (SPLIT) {
            pt = a.p_t + elem; pls = a.p_log_s + elem;
            ppi = a.p_log_pi + elem * K; pmu = a.p_mu + elem * K; pmls = a.p_ls + elem * K;
            row = nullptr; sf_d = nullptr; msf_d = nullptr;
        } else {
            row = a.nn + elem * a.P;
            sf_d = a.sf ? a.sf + d : nullptr;
            msf_d = a.msf ? a.msf + (size_t)d * K : nullptr;
            pt = pls = ppi = pmu = pmls = nullptr;
        }
    }
    __device__ __forceinline__ double t() const { return SPLIT ? pt[0] : (double)row[0]; }
    __device__ __forceinline__ double log_s() const {
        if (SPLIT) return pls[0];
        float v = row[1];
        if (sf_d) {
            const float f = expf(sf_d[0]);
            v = tanhf(v / fmaxf(f, 1.f)) * f;
        }
        return (double)v;
    }
    __device__ __forceinline__ double log_pi(int k) const { return SPLIT ? ppi[k] : (double)row[2 + k]; }
    __device__ __forceinline__ double mu(int k) const { return SPLIT ? pmu[k] : (double)row[2 + K + k]; }
    __device__ __forceinline__ double ls(int k) const {
        if (SPLIT) return pmls[k];
        float v = row[2 + 2 * K + k];
        if (msf_d) {
            const float f = expf(msf_d[k]);
            v = tanhf(v / fmaxf(f, 1.f)) * f;
        }
        return (double)v;
    }
};

// log-space mixture log-pdf (:217-223) — only used when the direct sum underflows
template <bool SPLIT>
__device__ __noinline__ double log_pdf_logspace(const ParamRow<SPLIT> p, int K, double x, double lse_pi) {
    double m = -INFINITY;
    for (int k = 0; k < K; ++k) {
        const double ls = p.ls(k);
        const double zk = (x - p.mu(k)) * exp(-ls);
        const double tk = p.log_pi(k) - lse_pi + zk - ls - 2.0 * softplus64(zk);
        m = (tk != tk || m != m) ? (double)NAN : fmax(m, tk);       // torch.max keeps a NaN
    }
    if (isinf(m)) return m;       // torch.logsumexp: every term -inf (a latent of -inf) gives -inf, not exp(-inf + inf)
    double s = 0.0;
    for (int k = 0; k < K; ++k) {
        const double ls = p.ls(k);
        const double zk = (x - p.mu(k)) * exp(-ls);
        s += exp(p.log_pi(k) - lse_pi + zk - ls - 2.0 * softplus64(zk) - m);
    }
    return m + log(s);
}

struct MixEval {
    double u;        // mixture CDF at x
    double log_pdf;  // mixture log-PDF at x
};

// Parameters of one transformed element.  With K a template constant (KT > 0) they are loaded once,
// in one batch of independent loads, and stay in registers; KT == 0 reads them lazily (any K).
template <bool SPLIT, int KT>
struct ElemParams {
    static constexpr int KK = KT > 0 ? KT : 1;
    ParamRow<SPLIT> row;
    double t_, log_s_;
    double lp[KK], mu_[KK], ls_[KK];
    int K;
    __device__ __forceinline__ ElemParams(const MixArgs& a, size_t elem, int d) : row(a, elem, d) {
        K = KT > 0 ? KT : a.K;
        t_ = row.t();
        log_s_ = row.log_s();
        if (KT > 0) {
#pragma unroll
            for (int k = 0; k < KK; ++k) {
                lp[k] = row.log_pi(k);
                mu_[k] = row.mu(k);
                ls_[k] = row.ls(k);
            }
        }
    }
    __device__ __forceinline__ double log_pi(int k) const { return KT > 0 ? lp[k] : row.log_pi(k); }
    __device__ __forceinline__ double mu(int k) const { return KT > 0 ? mu_[k] : row.mu(k); }
    __device__ __forceinline__ double ls(int k) const { return KT > 0 ? ls_[k] : row.ls(k); }
};

// one pass over the K mixtures at point x
template <bool SPLIT, int KT>
__device__ __forceinline__ MixEval eval_mixture(const ElemParams<SPLIT, KT>& p, double x) {
    const int K = p.K;
    double mx = -INFINITY, se = 0.0, cdf = 0.0, pdf = 0.0;
    if (KT > 0) {
#pragma unroll
        for (int k = 0; k < K; ++k) mx = fmax(mx, p.log_pi(k));
#pragma unroll
        for (int k = 0; k < K; ++k) {
            const double w = exp(p.log_pi(k) - mx);
            const double inv_s = exp(-p.ls(k));
            const double zk = (x - p.mu(k)) * inv_s;
            const double e = exp(-fabs(zk));
            const double r = 1.0 / (1.0 + e);
            const double sig = zk >= 0.0 ? r : e * r;
            se += w;
            cdf += w * sig;
            pdf += w * inv_s * (e * r * r);
        }
    } else {
        // run-time K: parameters come straight from memory; fetch them eight mixtures at a time so that 24
        // independent loads are in flight instead of one dependent load per use (444 us -> at K = 51)
        constexpr int CH = 8;
        for (int k0 = 0; k0 < K; k0 += CH) {
            double lpv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) lpv[j] = p.log_pi(min(k0 + j, K - 1));
#pragma unroll
            for (int j = 0; j < CH; ++j) mx = fmax(mx, lpv[j]);
        }
        for (int k0 = 0; k0 < K; k0 += CH) {
            double lpv[CH], muv[CH], lsv[CH];
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const int k = min(k0 + j, K - 1);
                lpv[j] = p.log_pi(k);
                muv[j] = p.mu(k);
                lsv[j] = p.ls(k);
            }
#pragma unroll
            for (int j = 0; j < CH; ++j) {
                const double w = (k0 + j < K) ? exp(lpv[j] - mx) : 0.0;
                const double inv_s = exp(-lsv[j]);
                const double zk = (x - muv[j]) * inv_s;
                const double e = exp(-fabs(zk));
                const double r = 1.0 / (1.0 + e);
                const double sig = zk >= 0.0 ? r : e * r;
                se += w;
                cdf += w * sig;
                pdf += w * inv_s * (e * r * r);
            }
        }
    }
    MixEval o;
    o.u = cdf / se;
    if (pdf > 1e-290) o.log_pdf = log(pdf / se);
    else o.log_pdf = log_pdf_logspace<SPLIT>(p.row, K, x, mx + log(se));
    return o;
}

// Forward of one transformed element in fp64 (run_with_params :100-123): out = (logit(u) + t) e^{log_s},
// contrib = its log-det term, reg = its regularisation term.
template <bool SPLIT, int KT>
__device__ __forceinline__ void forward_elem_f64(const MixArgs& a, const ElemParams<SPLIT, KT>& p, double x,
                                                 double& out, double& contrib, double& reg) {
    const MixEval ev = eval_mixture<SPLIT, KT>(p, x);
    const double u = ev.u;
    const double lu = safe_log(u), l1u = safe_log(1.0 - u);
    reg = 0.0;
    if (a.use_reg) {
        const double r1 = lu / kLn10, r2 = l1u / kLn10;
        reg = (fmin(r1, -a.reg_max) + a.reg_max) + (fmin(r2, -a.reg_max) + a.reg_max);
    }
    // y = -safe_log(1/u - 1) (:273) = log u - log(1-u) wherever safe_log(u) is not clamped; below u = 1e-22
    // the reference's form keeps falling (down to -inf at u == 0) while lu stays at log(1e-22)
    // (a NaN u stays a NaN: torch.clamp keeps it, the fmax of safe_log does not)
    const double y = u >= 1e-22 ? lu - l1u : (u != u ? u : -safe_log(1.0 / u - 1.0));
    const double mixt_ldj = -lu - l1u;
    out = (y + p.t_) * exp(p.log_s_);
    contrib = p.log_s_ + mixt_ldj + ev.log_pdf + reg * a.reg_factor;
}
// SPLIT selects the parameter source and the I/O precision (fp64 tensors for the static API);
// KT = compile-time number of mixtures (0 = run-time K); NEWTON = safeguarded Newton inverse.
#ifdef CNF_MIX64_WAVES
#define CNF_MIX64_ATTR __attribute__((amdgpu_waves_per_eu(CNF_MIX64_WAVES, 8)))
#else
#define CNF_MIX64_ATTR
#endif
template <bool SPLIT, bool REVERSE, int KT, bool NEWTON>
__global__ __launch_bounds__(kBlock) CNF_MIX64_ATTR void mixture_kernel(MixArgs a, RowTiling tl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    double* part = reinterpret_cast<double*>(smem) + (size_t)wave * kMaxTileChunks;
    double* cst = reinterpret_cast<double*>(smem) + (size_t)W * kMaxTileChunks;   // [3K][blockDim], KT == 0 only
    __shared__ int s_act[kMaxAct];   // kernel arguments cannot be indexed per lane
    if (threadIdx.x < kMaxAct && ((a.act_bits >> threadIdx.x) & 1ull))
        s_act[__popcll(a.act_bits & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
    __syncthreads();
    // block-per-row mode (few long rows): all waves of the workgroup share row blockIdx.x
    const int nlanes = tl.bpr ? (int)blockDim.x : kWave;       // lanes striding over one tile
    const int lane_t = tl.bpr ? (int)threadIdx.x : lane;
    const long tile = tl.bpr ? (long)blockIdx.x : (long)blockIdx.x * W + wave;
    if (tile >= tl.ntiles) return;
    const int row0 = (int)(tile * tl.rw);
    const int nrows = min(tl.rw, tl.B - row0);
    bool bad = false, range = false;

    // ---- phase 1: copy the elements that are not transformed (masked channel or padded token)
    {
        const long nel = (long)nrows * a.L;
        const size_t base = (size_t)row0 * a.L;
        for (long e = lane_t; e < nel; e += nlanes) {
            const int r = (int)(e / a.L);
            const int er = (int)(e - (long)r * a.L);
            const int n = (int)fdiv((uint32_t)er, a.div_d);
            const int d = er - n * a.D;
            const float m = mask_at(a.mask, a.mr, a.mc, n, d);
            const float pv = a.pad ? a.pad[(size_t)(row0 + r) * a.N + n] : 1.f;
            const float change = (1.f - m) * (a.pad_in_transform ? pv : 1.f);
            if (change == 0.f) {
                if (SPLIT) {
                    a.z_out64[base + e] = a.z64[base + e];
                    if (a.reg_elem64) a.reg_elem64[base + e] = 0.0;
                } else {
                    const float zv = a.z[base + e];
                    a.z_out[base + e] = a.pad_output ? zv * pv : zv;
                }
            }
        }
    }

    // ---- phase 2: transformed elements, one item per lane
    const int ipr = tl.L;                 // items per row = N * DA
    const int nitems = nrows * ipr;
    double acc1 = 0.0;
    for (int c = lane_t; c < nitems; c += nlanes) {
        const int r = tl.rw == 1 ? 0 : (int)fdiv((uint32_t)c, tl.div_cpr);
        const int it = c - r * ipr;
        const int n = (int)fdiv((uint32_t)it, a.div_da);
        const int d = s_act[it - n * a.DA];
        const int row = row0 + r;
        double contrib = 0.0;
        bool active = true;
        if (a.per_item_mask) active = mask_at(a.mask, a.mr, a.mc, n, d) == 0.f;
        const float pv = a.pad ? a.pad[(size_t)row * a.N + n] : 1.f;
        if (a.pad_in_transform && pv == 0.f) active = false;
        if (active) {
            const size_t elem = ((size_t)row * a.N + n) * a.D + d;
            const ElemParams<SPLIT, KT> p(a, elem, d);
            const int K = p.K;
            const double x = SPLIT ? a.z64[elem] : (double)a.z[elem];
            const double t = p.t_, log_s = p.log_s_;
            double out, reg = 0.0;
            if (!REVERSE) {
                forward_elem_f64<SPLIT, KT>(a, p, x, out, contrib, reg);
            } else {
                const double v = x * exp(-log_s) - t;
                double u = 1.0 / (1.0 + exp(-v));
                const double mixt_ldj = softplus64(v) + softplus64(-v);
                u = fmin(fmax(u, 1e-5), 1.0 - 1e-5);
                if (!(u > 0.0 && u < 1.0)) range = true;
                // per-mixture constants: registers when K is a template constant, else LDS [3k + c][tid]
                constexpr int KK = KT > 0 ? KT : 1;
                double wr[KK], isr[KK], mur[KK];
                double* my = cst + threadIdx.x;
                const int S = blockDim.x;
                double mx = -INFINITY;
#pragma unroll
                for (int k = 0; k < K; ++k) mx = fmax(mx, p.log_pi(k));
                double se = 0.0, spread = 0.0, lb = INFINITY, ub = -INFINITY;
                // NEWTON: every component's own u-quantile q_k = mu_k + s_k logit(u); the mixture quantile
                // lies in [min q_k, max q_k] (all component CDFs are <= u at min q_k and >= u at max q_k)
                const double logit_u = NEWTON ? log(u) - log(1.0 - u) : 0.0;
                double qsum = 0.0;
#pragma unroll
                for (int k = 0; k < K; ++k) {
                    const double w = exp(p.log_pi(k) - mx);
                    const double ls = p.ls(k);
                    const double sk = exp(ls);
                    const double mk = p.mu(k);
                    se += w;
                    spread += sk;
                    if (NEWTON) {
                        const double qk = mk + sk * logit_u;
                        lb = fmin(lb, qk);
                        ub = fmax(ub, qk);
                        qsum += w * qk;
                    }
                    if (KT > 0) {
                        wr[k] = w; isr[k] = exp(-ls); mur[k] = mk;
                    } else if (a.cst_lds) {
                        my[(3 * k + 0) * S] = w;
                        my[(3 * k + 1) * S] = exp(-ls);
                        my[(3 * k + 2) * S] = mk;
                    }
                }
                if (!NEWTON) {
                    // the reference's bracket (:252-254): mu_k -+ 20 * sum_k s_k, min / max over k
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        const double mu = KT > 0 ? mur[k] : p.mu(k);
                        lb = fmin(lb, mu - 20.0 * spread);
                        ub = fmax(ub, mu + 20.0 * spread);
                    }
                }
                const double target = u * se;          // compare un-normalised sums
                // start: x = 0 like the reference (:251), or the weight-averaged component quantile
                double xb = NEWTON ? fmin(fmax(qsum / se, lb), ub) : 0.0;
                double dx_prev = ub - lb;              // previous step length (Newton acceptance test)
                for (int iter = 0; iter < 100; ++iter) {
                    double cdf = 0.0, dens = 0.0;
#pragma unroll
                    for (int k = 0; k < K; ++k) {
                        double wk, ik, mk;
                        if (KT > 0) {
                            wk = wr[k]; ik = isr[k]; mk = mur[k];
                        } else if (a.cst_lds) {
                            wk = my[(3 * k + 0) * S]; ik = my[(3 * k + 1) * S]; mk = my[(3 * k + 2) * S];
                        } else {
                            // K too large for the LDS table: rebuild the constants from the (cache-resident)
                            // parameter row in every iteration, same expressions as the table holds
                            wk = exp(p.log_pi(k) - mx); ik = exp(-p.ls(k)); mk = p.mu(k);
                        }
                        const double zk = (xb - mk) * ik;
                        const double e = exp(-fabs(zk));
                        const double rr = 1.0 / (1.0 + e);
                        cdf += wk * (zk >= 0.0 ? rr : e * rr);
                        if (NEWTON) dens += wk * ik * (e * rr * rr);
                    }
                    // the reference's bisection step and bracket update (:244-248)
                    double nx;
                    if (cdf > target) {
                        nx = (xb + lb) / 2.0;
                        ub = xb;
                    } else {
                        nx = (xb + ub) / 2.0;
                        lb = xb;
                    }
                    if (NEWTON) {
                        // safeguarded Newton (rtsafe): take the Newton step from the same evaluation when it
                        // stays inside the bracket and at least halves the previous step, else bisect
                        const double f = cdf - target;
                        if (dens > 0.0 && fabs(2.0 * f) <= fabs(dx_prev * dens)) {
                            const double xn = xb - f / dens;
                            if (xn >= lb && xn <= ub) nx = xn;   // inclusive: at f == 0 the step is 0 and xn sits on the bracket
                        }
                    }
                    const double diff = fabs(nx - xb);
                    dx_prev = diff;
                    xb = nx;
                    if (!(diff > 1e-10)) break;
                }
                out = xb;
                const MixEval ev = eval_mixture<SPLIT, KT>(p, xb);
                contrib = log_s + mixt_ldj + ev.log_pdf;
            }
            if (SPLIT) {
                a.z_out64[elem] = out;
                if (a.reg_elem64) a.reg_elem64[elem] = reg;
                bad |= isnan(out);
            } else {
                float of = (float)out;
                if (a.pad_output) of = of * pv;
                a.z_out[elem] = of;
                bad |= isnan(of);
                if (a.use_reg && a.reg_out && reg != 0.0) atomicAdd(&a.reg_out[row], (float)reg);
            }
        }
        if (tl.rw == 1) acc1 += contrib;
        else part[c] = contrib;
    }

    // ---- per-sample log-det
    auto finish = [&](int row, double sum) {
        const double s = REVERSE ? -sum : sum;
        if (SPLIT) {
            a.ldj_out64[row] = s;
            if (isnan(s)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        } else {
            const float v = (a.ldj_in ? a.ldj_in[row] : 0.f) + (float)s;
            a.ldj_out[row] = v;
            if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        }
    };
    if (tl.bpr) {
        acc1 = wave_sum(acc1);
        if (lane == 0) part[0] = acc1;                 // this wave's strip, slot 0
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < W; ++w) t += reinterpret_cast<double*>(smem)[(size_t)w * kMaxTileChunks];
            finish(row0, t);
        }
    } else if (tl.rw == 1) {
        acc1 = wave_sum(acc1);
        if (lane == 0) finish(row0, acc1);
    } else {
        wave_lds_sync();
        const int g = kWave / tl.p2;
        const int sub = lane & (g - 1);
        for (int r0 = 0; r0 < nrows; r0 += tl.p2) {
            const int r = r0 + lane / g;
            double acc = 0.0;
            if (r < nrows)
                for (int i = sub; i < ipr; i += g) acc += part[r * ipr + i];
            acc = group_sum(acc, g);
            if (sub == 0 && r < nrows) finish(row0 + r, acc);
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    if (range) raise_flag(a.flags, CNF_FLAG_RANGE);
}


// ---- fast forward (math mode 1): fp32 arithmetic on LDS-staged parameter rows -------------------------------
//
// The fp64 kernel above is bound by software fp64 exp/log (2 waves/SIMD at 200+ VGPRs) and by its parameter
// fetch: every lane walks its own 4(2+3K)-byte row with scalar loads, so one wave-level load touches 64 cache
// lines.  This kernel
//   * stages the rows of the 64 items a wave works on through its private LDS strip with coalesced loads (the
//     rows of neighbouring items are contiguous or a fixed stride apart; every 128-byte line is fetched once),
//     then each lane reads its row back at an odd stride (conflict-free);
//   * evaluates the mixture in fp32 with the hardware exp2 / log2 / rcp, carrying BOTH tails as sums of
//     positive terms,  u = sum w sigma(z_k) / sum w  and  1 - u = sum w sigma(-z_k) / sum w,  so that log u and
//     log(1 - u) keep fp32 RELATIVE accuracy (the reference needs fp64 only because it forms 1 - u by
//     subtraction);
//   * hands elements with u or 1 - u below 1e-9 (|logit| > 20.7), or an underflowing PDF sum, to the fp64
//     element routine, which reproduces the reference's clamps (safe_log 1e-22) and rounding there.
// Differences to the fp64 kernel on the fast branch: <= 3e-6 absolute in z_out / per-element log-det
// (tests/test_gpu_parity.py::test_mixture_fast_vs_exact).  The per-sample log-det is still summed in fp64.
template <int KT, bool REVERSE, int G = 1>
__global__ __launch_bounds__(kBlock) void mixture_f32_kernel(MixArgs a, RowTiling tl, int PS, int strip) {
    static_assert(G == 1 || KT == 0, "several lanes per item only with a run-time K");
    constexpr int IPP = kWave / G;            // items per wave pass
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int W = blockDim.x >> 6;
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int K = KT > 0 ? KT : a.K;
    // LDS: [W][strip] fp64 row partials | [W][64 * PS] staged rows | [D] + [D*K] bound tables
    double* part = reinterpret_cast<double*>(smem) + (size_t)wave * strip;
    float* stage_all = reinterpret_cast<float*>(reinterpret_cast<double*>(smem) + (size_t)W * strip);
    float* stage = stage_all + (size_t)wave * IPP * PS;
    BoundTab* sf_tab = reinterpret_cast<BoundTab*>(stage_all + (size_t)W * IPP * PS);
    BoundTab* msf_tab = sf_tab + a.D;
    __shared__ int s_act[kMaxAct];
    if (threadIdx.x < kMaxAct && ((a.act_bits >> threadIdx.x) & 1ull))
        s_act[__popcll(a.act_bits & ((1ull << threadIdx.x) - 1ull))] = threadIdx.x;
    for (int i = threadIdx.x; i < a.D; i += blockDim.x)
        if (a.sf) sf_tab[i] = make_bound(a.sf[i]);
    for (int i = threadIdx.x; i < a.D * K; i += blockDim.x)
        if (a.msf) msf_tab[i] = make_bound(a.msf[i]);
    __syncthreads();
    const int nlanes = tl.bpr ? (int)blockDim.x : kWave;
    const int lane_t = tl.bpr ? (int)threadIdx.x : lane;
    const long tile = tl.bpr ? (long)blockIdx.x : (long)blockIdx.x * W + wave;
    // a wave without a tile still has to reach no later barrier: none follow
    if (tile >= tl.ntiles) return;
    const int row0 = __builtin_amdgcn_readfirstlane((int)(tile * tl.rw));      // wave-uniform by construction
    const int nrows = min(tl.rw, tl.B - row0);
    const size_t tile_elem0 = (size_t)row0 * a.N * a.D;
    const float* tile_nn = a.nn + tile_elem0 * (size_t)a.P;
    bool bad = false, range = false;

    // ---- phase 1: copy the elements that are not transformed (masked channel or padded token)
    {
        const long nel = (long)nrows * a.L;
        const size_t base = (size_t)row0 * a.L;
        for (long e = lane_t; e < nel; e += nlanes) {
            const int r = (int)(e / a.L);
            const int er = (int)(e - (long)r * a.L);
            const int n = (int)fdiv((uint32_t)er, a.div_d);
            const int d = er - n * a.D;
            const float m = mask_at(a.mask, a.mr, a.mc, n, d);
            const float pv = a.pad ? a.pad[(size_t)(row0 + r) * a.N + n] : 1.f;
            const float change = (1.f - m) * (a.pad_in_transform ? pv : 1.f);
            if (change == 0.f) {
                const float zv = a.z[base + e];
                a.z_out[base + e] = a.pad_output ? zv * pv : zv;
            }
        }
    }

    // ---- phase 2: transformed elements, 64 consecutive items per wave pass
    const int ipr = tl.L;
    const int nitems = nrows * ipr;
    const int P = a.P;
    const bool vec2 = (P & 1) == 0;                 // rows are 8-byte aligned -> stage in 8-byte units
    const int upi = vec2 ? P >> 1 : P;              // units per item
    const FastDiv div_upi = a.div_upi;
    double acc1 = 0.0;
    const int sub = lane & (G - 1), li = lane / G;      // lane's share of the mixtures, lane's item in the pass
    for (int c0 = tl.bpr ? wave * IPP : 0; c0 < nitems; c0 += nlanes / G) {     // wave-uniform
        const int c = c0 + li;
        const bool valid = c < nitems;
        const int cc = valid ? c : nitems - 1;
        const int r = tl.rw == 1 ? 0 : (int)fdiv((uint32_t)cc, tl.div_cpr);
        const int it = cc - r * ipr;
        const int n = (int)fdiv((uint32_t)it, a.div_da);
        const int d = s_act[it - n * a.DA];
        const int row = row0 + r;
        bool active = valid;
        if (a.per_item_mask) active = active && mask_at(a.mask, a.mr, a.mc, n, d) == 0.f;
        const float pv = a.pad ? a.pad[(size_t)row * a.N + n] : 1.f;
        if (a.pad_in_transform && pv == 0.f) active = false;
        const size_t elem = ((size_t)row * a.N + n) * a.D + d;
        const float x = active ? a.z[elem] : 0.f;
        // -- stage the rows of the pass: unit q belongs to item q / upi, whose row start (a 32-bit offset from
        // the tile's first parameter row, which is wave-uniform -> scalar base + 32-bit vector offset addressing)
        // comes from that item's lane by shuffle
        const unsigned rowoff = active ? (unsigned)((elem - tile_elem0) * (size_t)P) : ~0u;
        const int total = IPP * upi;
        constexpr int UR = 8;           // units in flight per lane and round
        for (int q0 = 0; q0 < total; q0 += kWave * UR) {
            float2 v[UR];
            unsigned ok = 0;
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                const int q = min(q0 + u * kWave + lane, total - 1);
                const int item = (int)fdiv((uint32_t)q, div_upi);
                const int off = q - item * upi;
                const unsigned b = __shfl(rowoff, item * G, kWave);
                // unconditional loads (an inactive item reads the tile's first row): no branch per unit
                const unsigned bs = b == ~0u ? 0u : b;
                if (b != ~0u && q0 + u * kWave + lane < total) ok |= 1u << u;
                if (vec2) v[u] = *reinterpret_cast<const float2*>(tile_nn + (bs + 2u * (unsigned)off));
                else v[u] = make_float2(tile_nn[bs + (unsigned)off], 0.f);
            }
#pragma unroll
            for (int u = 0; u < UR; ++u) {
                if ((ok >> u) & 1u) {
                    const int q = q0 + u * kWave + lane;
                    const int item = (int)fdiv((uint32_t)q, div_upi);
                    const int off = q - item * upi;
                    float* dst = stage + item * PS + (vec2 ? 2 * off : off);
                    dst[0] = v[u].x;
                    if (vec2) dst[1] = v[u].y;
                }
            }
        }
        wave_lds_sync();
        double contrib = 0.0;
        if (active && REVERSE) {
            // ---- inverse (:125-134, :235-264) in fp32: safeguarded Newton on the two-sided CDF.  u = sigmoid(v)
            // is clamped to [1e-5, 1 - 1e-5] by the reference, so the root lies where fp32 sums of positive
            // terms are accurate to ~1e-6 relative in BOTH tails: solve cdf(x) = u se for u <= 1/2 and
            // ccdf(x) = (1 - u) se otherwise.  A relative error eps of the sum moves x by ~eps * s_k.
            float* my = stage + li * PS;
            const float t = my[0];
            float log_s = my[1];
            if (a.sf) log_s = apply_bound(log_s, sf_tab[d]);
            const float v = x * __builtin_amdgcn_exp2f(-log_s * kLog2eF) - t;
            const float ev = __builtin_amdgcn_exp2f(-fabsf(v) * kLog2eF);
            const float rv = __builtin_amdgcn_rcpf(1.f + ev);
            const float mixt_ldj = fabsf(v) + 2.f * kLn2F * __builtin_amdgcn_logf(1.f + ev);
            float usmall = fmaxf(ev * rv, 1e-5f), ubig = fminf(rv, 1.f - 1e-5f);     // (:130) clamp
            const bool upper = v >= 0.f;                    // u > 1/2
            const float u = upper ? ubig : usmall, uc = upper ? usmall : ubig;
            if (!(u > 0.f && u < 1.f)) range = true;
            const float logit_u = (__builtin_amdgcn_logf(u) - __builtin_amdgcn_logf(uc)) * kLn2F;
            const BoundTab* mt = msf_tab + d * K;
            constexpr int KK = KT > 0 ? KT : 1;
            float wr[KK], isr[KK], mur[KK];
            float mx = -INFINITY;
            if (KT > 0) {
#pragma unroll
                for (int k = 0; k < KK; ++k) mx = fmaxf(mx, my[2 + k]);
            } else {
                for (int k = sub; k < K; k += G) mx = fmaxf(mx, my[2 + k]);
                mx = gmax<G>(mx);
            }
            float se = 0.f, spread = 0.f, lb = INFINITY, ub = -INFINITY, qsum = 0.f;
            auto setup = [&](int k) {
                const float lsk = my[2 + 2 * K + k];
                const float ls = a.msf ? apply_bound(lsk, mt[k]) : lsk;
                const float w = __builtin_amdgcn_exp2f((my[2 + k] - mx) * kLog2eF);
                const float sk = __builtin_amdgcn_exp2f(ls * kLog2eF);
                const float ik = __builtin_amdgcn_rcpf(sk);
                const float mk = my[2 + K + k];
                // every component's own u-quantile: the mixture quantile lies between their min and max
                const float qk = fmaf(sk, logit_u, mk);
                se += w;
                spread += sk;
                lb = fminf(lb, qk);
                ub = fmaxf(ub, qk);
                qsum = fmaf(w, qk, qsum);
                if (KT > 0) {
                    wr[k < KK ? k : 0] = w; isr[k < KK ? k : 0] = ik; mur[k < KK ? k : 0] = mk;
                } else {
                    my[2 + k] = w;              // run-time K: the constants replace the raw row in LDS
                    my[2 + 2 * K + k] = ik;
                }
            };
            if (KT > 0) {
#pragma unroll
                for (int k = 0; k < KK; ++k) setup(k);
            } else {
                for (int k = sub; k < K; k += G) setup(k);
                se = gsum<G>(se); spread = gsum<G>(spread); qsum = gsum<G>(qsum);
                lb = gmin<G>(lb); ub = gmax<G>(ub);
            }
            const float target = (upper ? uc : u) * se;
            const float tol_scale = 1e-7f * spread;
            float xb = fminf(fmaxf(qsum * __builtin_amdgcn_rcpf(se), lb), ub);
            float dx_prev = ub - lb, dens = 0.f, diff = INFINITY;
            auto eval = [&](float xq, float& f_out, float& dens_out) {
                float cdf = 0.f, ccdf = 0.f, dn = 0.f;
                auto one = [&](float wk, float ik, float mk) {
                    const float zk = (xq - mk) * ik;
                    const float e = __builtin_amdgcn_exp2f(-fabsf(zk) * kLog2eF);
                    const float rr = __builtin_amdgcn_rcpf(1.f + e);
                    const float er = e * rr;
                    const bool pos = zk >= 0.f;
                    cdf = fmaf(wk, pos ? rr : er, cdf);
                    ccdf = fmaf(wk, pos ? er : rr, ccdf);
                    dn = fmaf(wk * ik, er * rr, dn);
                };
                if (KT > 0) {
#pragma unroll
                    for (int k = 0; k < KK; ++k) one(wr[k], isr[k], mur[k]);
                } else {
                    for (int k = sub; k < K; k += G) one(my[2 + k], my[2 + 2 * K + k], my[2 + K + k]);
                    cdf = gsum<G>(cdf); ccdf = gsum<G>(ccdf); dn = gsum<G>(dn);
                }
                f_out = upper ? target - ccdf : cdf - target;       // increasing in x either way
                dens_out = dn;
            };
            for (int iter = 0; iter < 64; ++iter) {
                float f;
                eval(xb, f, dens);
                float nx;
                if (f > 0.f) {
                    nx = 0.5f * (xb + lb);
                    ub = xb;
                } else {
                    nx = 0.5f * (xb + ub);
                    lb = xb;
                }
                // rtsafe: Newton step when it stays in the bracket and at least halves the previous step
                if (dens > 0.f && fabsf(2.f * f) <= fabsf(dx_prev * dens)) {
                    const float xn = xb - f * __builtin_amdgcn_rcpf(dens);
                    if (xn >= lb && xn <= ub) nx = xn;
                }
                diff = fabsf(nx - xb);
                dx_prev = diff;
                xb = nx;
                if (!(diff > fmaf(1e-7f, fabsf(xb), tol_scale))) break;
            }
            if (diff > 1e-5f * (fabsf(xb) + spread)) {      // left the loop far from converged: density at the final point
                float f;
                eval(xb, f, dens);
            }
            const float lpdf = (__builtin_amdgcn_logf(dens) - __builtin_amdgcn_logf(se)) * kLn2F;
            float of = xb;
            if (a.pad_output) of = of * pv;
            if (sub == 0) {
                a.z_out[elem] = of;
                contrib = (double)(log_s + mixt_ldj + lpdf);
            }
            bad |= isnan(of);
        }
        if (active && !REVERSE) {
            const float* my = stage + li * PS;
            const float t = my[0];
            float log_s = my[1];
            if (a.sf) log_s = apply_bound(log_s, sf_tab[d]);
            constexpr int KK = KT > 0 ? KT : 1;
            float lp[KK], mu[KK], lsr[KK];
            float mx = -INFINITY;
            if (KT > 0) {
#pragma unroll
                for (int k = 0; k < KK; ++k) {
                    lp[k] = my[2 + k];
                    mu[k] = my[2 + KK + k];
                    lsr[k] = my[2 + 2 * KK + k];
                }
#pragma unroll
                for (int k = 0; k < KK; ++k) mx = fmaxf(mx, lp[k]);
            } else {
                for (int k = sub; k < K; k += G) mx = fmaxf(mx, my[2 + k]);
                mx = gmax<G>(mx);
            }
            const BoundTab* mt = msf_tab + d * K;
            float se = 0.f, cdf = 0.f, ccdf = 0.f, pdf = 0.f;
            auto one = [&](float lpk, float muk, float lsk, int k) {
                const float ls = a.msf ? apply_bound(lsk, mt[k]) : lsk;
                const float inv_s = __builtin_amdgcn_exp2f(-ls * kLog2eF);
                const float w = __builtin_amdgcn_exp2f((lpk - mx) * kLog2eF);
                const float zk = (x - muk) * inv_s;
                const float e = __builtin_amdgcn_exp2f(-fabsf(zk) * kLog2eF);
                const float rr = __builtin_amdgcn_rcpf(1.f + e);
                const float er = e * rr;
                const bool pos = zk >= 0.f;
                se += w;
                cdf = fmaf(w, pos ? rr : er, cdf);
                ccdf = fmaf(w, pos ? er : rr, ccdf);
                pdf = fmaf(w * inv_s, er * rr, pdf);
            };
            if (KT > 0) {
#pragma unroll
                for (int k = 0; k < KK; ++k) one(lp[k], mu[k], lsr[k], k);
            } else {
                for (int k = sub; k < K; k += G) one(my[2 + k], my[2 + K + k], my[2 + 2 * K + k], k);
                se = gsum<G>(se); cdf = gsum<G>(cdf); ccdf = gsum<G>(ccdf); pdf = gsum<G>(pdf);
            }
            float of;
            double reg = 0.0;
            const float inv_se = __builtin_amdgcn_rcpf(se);
            const float u = cdf * inv_se, uc = ccdf * inv_se;
            if (u > 1e-9f && uc > 1e-9f && pdf > 1e-30f) {
                const float l2se = __builtin_amdgcn_logf(se);
                const float lu = (__builtin_amdgcn_logf(cdf) - l2se) * kLn2F;
                const float l1u = (__builtin_amdgcn_logf(ccdf) - l2se) * kLn2F;
                const float lpdf = (__builtin_amdgcn_logf(pdf) - l2se) * kLn2F;
                float regf = 0.f;
                if (a.use_reg) {
                    const float rmax = (float)a.reg_max;
                    const float r1 = lu * 0.43429448190325176f, r2 = l1u * 0.43429448190325176f;
                    regf = (fminf(r1, -rmax) + rmax) + (fminf(r2, -rmax) + rmax);
                }
                of = ((lu - l1u) + t) * __builtin_amdgcn_exp2f(log_s * kLog2eF);
                contrib = (double)(log_s + (-lu - l1u) + lpdf + regf * (float)a.reg_factor);
                reg = (double)regf;
            } else {
                // rare: a tail or an underflow.  The reference's fp64 arithmetic (u, then 1 - u by subtraction,
                // safe_log clamps; :100-123, :217-233) on the staged row, one mixture at a time — rolled loops
                // keep this branch's registers below the fast path's.
                const double xd = (double)x;
                double sed = 0.0, cdfd = 0.0, pdfd = 0.0;
#pragma clang loop unroll(disable)
                for (int k = 0; k < K; ++k) {
                    const float lsf = a.msf ? apply_bound(my[2 + 2 * K + k], mt[k]) : my[2 + 2 * K + k];
                    const double wd = exp((double)my[2 + k] - (double)mx);
                    const double isd = exp(-(double)lsf);
                    const double zd = (xd - (double)my[2 + K + k]) * isd;
                    const double ed = exp(-fabs(zd));
                    const double rd = 1.0 / (1.0 + ed);
                    sed += wd;
                    cdfd += wd * (zd >= 0.0 ? rd : ed * rd);
                    pdfd += wd * isd * (ed * rd * rd);
                }
                const double ud = cdfd / sed;
                double lpdfd;
                if (pdfd > 1e-290) {
                    lpdfd = log(pdfd / sed);
                } else {
                    // log-space form (:217-223)
                    const double lse_pi = (double)mx + log(sed);
                    double m = -INFINITY;
#pragma clang loop unroll(disable)
                    for (int k = 0; k < K; ++k) {
                        const float lsf = a.msf ? apply_bound(my[2 + 2 * K + k], mt[k]) : my[2 + 2 * K + k];
                        const double zd = (xd - (double)my[2 + K + k]) * exp(-(double)lsf);
                        const double tk = (double)my[2 + k] - lse_pi + zd - (double)lsf - 2.0 * softplus64(zd);
                        m = (tk != tk || m != m) ? (double)NAN : fmax(m, tk);
                    }
                    double ssum = 0.0;
#pragma clang loop unroll(disable)
                    for (int k = 0; k < K; ++k) {
                        const float lsf = a.msf ? apply_bound(my[2 + 2 * K + k], mt[k]) : my[2 + 2 * K + k];
                        const double zd = (xd - (double)my[2 + K + k]) * exp(-(double)lsf);
                        ssum += exp((double)my[2 + k] - lse_pi + zd - (double)lsf - 2.0 * softplus64(zd) - m);
                    }
                    lpdfd = isinf(m) ? m : m + log(ssum);
                }
                const double lud = safe_log(ud), l1ud = safe_log(1.0 - ud);
                if (a.use_reg) {
                    const double r1 = lud / kLn10, r2 = l1ud / kLn10;
                    reg = (fmin(r1, -a.reg_max) + a.reg_max) + (fmin(r2, -a.reg_max) + a.reg_max);
                }
                const double yd = ud >= 1e-22 ? lud - l1ud : -safe_log(1.0 / ud - 1.0);
                of = (float)((yd + (double)t) * exp((double)log_s));
                contrib = (double)log_s + (-lud - l1ud) + lpdfd + reg * a.reg_factor;
            }
            if (a.pad_output) of = of * pv;
            bad |= isnan(of);
            if (sub == 0) {
                a.z_out[elem] = of;
                if (a.use_reg && a.reg_out && reg != 0.0) atomicAdd(&a.reg_out[row], (float)reg);
            } else {
                contrib = 0.0;      // the item's log-det term is counted once
            }
        }
        if (valid && sub == 0) {
            if (tl.rw == 1) acc1 += contrib;
            else part[c] = contrib;
        }
        wave_lds_sync();      // the strip is overwritten by the next pass
    }

    auto finish = [&](int row, double sum) {
        const float v = (a.ldj_in ? a.ldj_in[row] : 0.f) + (float)(REVERSE ? -sum : sum);
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    };
    if (tl.bpr) {
        acc1 = wave_sum(acc1);
        if (lane == 0) part[0] = acc1;
        __syncthreads();
        if (threadIdx.x == 0) {
            double tsum = 0.0;
            for (int w = 0; w < W; ++w) tsum += reinterpret_cast<double*>(smem)[(size_t)w * strip];
            finish(row0, tsum);
        }
    } else if (tl.rw == 1) {
        acc1 = wave_sum(acc1);
        if (lane == 0) finish(row0, acc1);
    } else {
        wave_lds_sync();
        const int g = kWave / tl.p2;
        const int sub = lane & (g - 1);
        for (int r0 = 0; r0 < nrows; r0 += tl.p2) {
            const int r = r0 + lane / g;
            double acc = 0.0;
            if (r < nrows)
                for (int i = sub; i < ipr; i += g) acc += part[r * ipr + i];
            acc = group_sum(acc, g);
            if (sub == 0 && r < nrows) finish(row0 + r, acc);
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    if (range) raise_flag(a.flags, CNF_FLAG_RANGE);
}

// get_mixt_params (:145-180): split + fp32 tanh bound + mask, widened to fp64
__global__ __launch_bounds__(kBlock) void mixture_params_kernel(const float* nn, const float* sf,
                                                                const float* msf, const float* mask,
                                                                int mr, int mc, double* t, double* log_s,
                                                                double* log_pi, double* mu, double* ls,
                                                                long nelem, int N, int D, int K) {
    const int P = 2 + 3 * K;
    const long total = nelem * P;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const long elem = i / P;
        const int j = (int)(i - elem * P);
        const int d = (int)(elem % D);
        const int n = (int)((elem / D) % N);
        const float keep = 1.f - mask_at(mask, mr, mc, n, d);
        float v = nn[i];
        if (j == 0) {
            t[elem] = (double)(mask ? v * keep : v);
        } else if (j == 1) {
            if (sf) {
                const float f = expf(sf[d]);
                v = tanhf(v / fmaxf(f, 1.f)) * f;
            }
            log_s[elem] = (double)(mask ? v * keep : v);
        } else if (j < 2 + K) {
            log_pi[elem * K + (j - 2)] = (double)(mask ? v * keep : v);
        } else if (j < 2 + 2 * K) {
            mu[elem * K + (j - 2 - K)] = (double)(mask ? v * keep : v);
        } else {
            const int k = j - 2 - 2 * K;
            if (msf) {
                const float f = expf(msf[d * K + k]);
                v = tanhf(v / fmaxf(f, 1.f)) * f;
            }
            ls[elem * K + k] = (double)(mask ? v * keep : v);
        }
    }
}

// fallback of cnf_mixture_coupling_nll: per-sample NLL -> the 64 fixed-point batch-sum words
__global__ void nll_to_acc_kernel(const float* nll, long long* acc, int B) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    if (row < B) nll_acc_add(acc, row & 63, nll[row]);
}

static std::atomic<int> g_mix_kernel{0};     // 0: token-pass kernel first; 1: round-1 fp32 kernel (A/B and tests)
static std::atomic<int> g_mix_lanes{0};      // lanes per item of the token-pass kernel with a run-time K: 0 = automatic, 1 / 2 / 4

// Which channels are transformed.  The caller knows the (tiny, constant) coupling mask on the host
// and passes the list of transformed channels; without it every item tests the mask itself.
static void fill_act(MixArgs& a, const int* act_host, int n_act) {
    a.per_item_mask = 0;
    a.act_bits = 0;
    if (a.mask && act_host && a.mr == 1 && a.mc == a.D && n_act >= 0 && n_act <= a.D) {
        a.DA = n_act;
        for (int i = 0; i < n_act; ++i) a.act_bits |= 1ull << act_host[i];
    } else {
        a.DA = a.D;
        a.act_bits = a.D >= 64 ? ~0ull : ((1ull << a.D) - 1ull);
        a.per_item_mask = a.mask ? 1 : 0;
    }
    if (a.DA == 0) {   // nothing is transformed: keep one (inactive) item per token
        a.DA = 1;
        a.act_bits = 1ull;
        a.per_item_mask = 1;
    }
}

}  // namespace cnf

using namespace cnf;

extern "C" {

static int launch_mixture(MixArgs& a, bool split, const int* act_host, int n_act, hipStream_t st,
                          const char* who) {
    CNF_REQUIRE(a.B >= 0 && a.N > 0 && a.D > 0 && a.K > 0, "%s: bad shape B=%d N=%d D=%d K=%d", who, a.B, a.N, a.D, a.K);
    if (a.B == 0) return CNF_OK;
    CNF_REQUIRE(a.D <= kMaxAct, "%s: D=%d exceeds %d", who, a.D, kMaxAct);
    CNF_REQUIRE((long)a.N * a.D < 65536, "%s: N*D=%ld exceeds 65535", who, (long)a.N * a.D);
    if (!a.mask) { a.mr = 1; a.mc = a.D; }
    CNF_REQUIRE(a.mr >= 1 && (a.mc == a.D || a.mc == 1), "%s: mask must be [rows,%d] or [rows,1]", who, a.D);
    if (a.mr > a.N) a.mr = a.N;
    a.P = 2 + 3 * a.K;
    a.L = a.N * a.D;
    a.div_d = make_fastdiv((uint32_t)a.D);
    fill_act(a, act_host, n_act);
    a.div_da = make_fastdiv((uint32_t)a.DA);
    a.nn_D = a.D; a.nn_c0 = 0;
    if (a.compact) {
        // nn_out = [B, N, DA * P]: the blocks of the transformed channels only (a channel mask whose transformed channels are one range)
        CNF_REQUIRE(!split && a.mask && !a.per_item_mask && a.act_bits != 0, "%s: the compact layout needs a channel mask and its host channel list", who);
        a.nn_D = a.DA;
        a.nn_c0 = __builtin_ctzll(a.act_bits);
    }
    const RowTiling tl = make_row_tiling(a.B, a.N * a.DA, /*force_vec=*/1);
    // block size: with a run-time K the inverse keeps 3K fp64 constants per thread in LDS (<= 64 KiB/block)
    const int kt = (a.K == 4 || a.K == 8 || a.K == 16) ? a.K : 0;
    int threads = kBlock;
    size_t cst_bytes = 0;
    if (a.reverse && kt == 0) {
        const size_t per_thread = (size_t)3 * a.K * sizeof(double);
        while (threads > kWave &&
               per_thread * threads + (size_t)(threads / kWave) * kMaxTileChunks * sizeof(double) > 65536)
            threads >>= 1;
        cst_bytes = per_thread * threads;
        a.cst_lds = 1;
        if (cst_bytes + (size_t)(threads / kWave) * kMaxTileChunks * sizeof(double) > 65536) {
            // K > 42: no LDS table, the kernel recomputes the per-mixture constants in its iterations
            a.cst_lds = 0;
            cst_bytes = 0;
            threads = kBlock;
        }
    }
    // fast math mode: fp32 kernels on LDS-staged parameter rows (forward; inverse when the Newton mode is selected).
    // First choice is the token-pass kernel on DMA-staged rows (cnf_mixture_tok.hip); shapes it is not built for
    // (transformed channels not a contiguous range, unaligned nn_out) take the round-1 kernel below.
    if (!split && math_mode() == 1 && (!a.reverse || inverse_mode() == 1) && g_mix_kernel != 1) {
        if (launch_mixture_tok(a, st, g_mix_lanes)) return launch_status(who);
    }
    // math mode 0 (fp64 like the reference) on the same token passes; the kernel below stays for the fp64-tensor API,
    // bisection, K > 64 and shapes the token-pass geometry declines
    if (!split && math_mode() == 0 && (!a.reverse || inverse_mode() == 1) && g_mix_kernel != 1 && !a.e_w && !a.nll_out) {
        if (launch_mixture_tok(a, st, 0, /*x64=*/true)) return launch_status(who);
    }
    if (a.compact && !a.e_w && !a.nll_out) {
        // only the token-pass kernels read the compact layout: the caller expands nn_out to the reference layout and calls again
        set_error("%s: shape / mode outside the compact-layout kernels (B=%d N=%d D=%d K=%d)", who, a.B, a.N, a.D, a.K);
        return CNF_ERR_UNSUPPORTED;
    }
    if (a.e_w) {
        // the token-pass kernel declined: the plain coupling, then the fused ActNorm + 1x1 convolution kernel in place
        MixArgs b = a;
        b.e_w = nullptr; b.e_bias = nullptr; b.e_scales = nullptr; b.e_sldj = nullptr; b.e_length = nullptr;
        const int rc = launch_mixture(b, split, act_host, n_act, st, who);
        if (rc != CNF_OK) return rc;
        return cnf_actnorm_invconv(a.z_out, a.e_bias, a.e_scales, a.e_w, a.e_sldj, a.pad, a.e_length, a.ldj_out, a.z_out, a.ldj_out,
                                   a.B, a.N, a.D, 0, a.flags, (cnf_stream_t)st);
    }
    if (a.nll_out) {
        // the token-pass kernel declined (or another math / inverse mode is selected): same results from the plain
        // coupling followed by the separate prior / NLL kernel (and one tiny launch for the fixed-point batch sum)
        MixArgs b = a;
        b.nll_out = nullptr; b.neglog_out = nullptr; b.nll_acc = nullptr; b.length = nullptr;
        const int rc = launch_mixture(b, split, act_host, n_act, st, who);
        if (rc != CNF_OK) return rc;
        const float sigma = 1.f / a.prior.inv_sigma;
        const int rc2 = cnf_prior_nll(a.z_out, a.pad, a.ldj_out, a.length, a.neglog_out, a.nll_out, nullptr, a.B, a.N, a.D,
                                      sigma, a.prior.log_sigma, (cnf_stream_t)st);
        if (rc2 != CNF_OK) return rc2;
        if (a.nll_acc) CNF_LAUNCH(nll_to_acc_kernel, dim3((a.B + kBlock - 1) / kBlock), dim3(kBlock), 0, st, a.nll_out, a.nll_acc, a.B);
        return launch_status(who);
    }
    if (!split && math_mode() == 1 && (!a.reverse || inverse_mode() == 1)) {
        // smaller tiles than the fp64 kernel: at ~100 VGPRs four waves per SIMD are resident, and a
        // config-sized batch (5e5 items) only fills them when a wave owns ~128 items
        const RowTiling tl = make_row_tiling(a.B, a.N * a.DA, /*force_vec=*/1, mixture_tile_items());
        const int PS = a.P | 1;                                   // odd row stride in LDS
        const int upi = (a.P & 1) ? a.P : a.P / 2;
        a.div_upi = make_fastdiv((uint32_t)upi);
        const int strip = (tl.bpr || tl.rw == 1) ? 1 : tl.rw * tl.L;
        const size_t tabs = (size_t)(a.D + a.D * a.K) * 3 * sizeof(float);
        const int th = kBlock;
        // 64 staged rows per wave while four waves fit into 64 KiB (K <= ~19); beyond that a wave stages 16 rows
        // and four lanes share an item (the language model's K = 51)
        auto need = [&](int ipp) { return (size_t)(th / kWave) * ((size_t)strip * sizeof(double) + (size_t)ipp * PS * sizeof(float)) + tabs; };
        const int G = need(kWave) <= 65536 ? 1 : 4;
        if (need(kWave / G) <= 65536 && (size_t)kWave * upi < 65536) {
            const int Wf = th / kWave;
            const dim3 gridf(tl.bpr ? (unsigned)tl.ntiles : (unsigned)((tl.ntiles + Wf - 1) / Wf)), blockf(th);
            const size_t lds = need(kWave / G);
#define CNF_MIXF(KT_, G_)                                                                                          \
    if (a.reverse) CNF_LAUNCH((mixture_f32_kernel<KT_, true, G_>), gridf, blockf, lds, st, a, tl, PS, strip); \
    else CNF_LAUNCH((mixture_f32_kernel<KT_, false, G_>), gridf, blockf, lds, st, a, tl, PS, strip)
            if (G == 4) { CNF_MIXF(0, 4); }
            else switch (kt) {
                case 4: CNF_MIXF(4, 1); break;
                case 8: CNF_MIXF(8, 1); break;
                case 16: CNF_MIXF(16, 1); break;
                default: CNF_MIXF(0, 1); break;
            }
#undef CNF_MIXF
            return launch_status(who);
        }
    }
    const int W = threads / kWave;
    const size_t smem = (size_t)W * kMaxTileChunks * sizeof(double) + cst_bytes;
    const dim3 grid(tl.bpr ? (unsigned)tl.ntiles : (unsigned)((tl.ntiles + W - 1) / W)), block(threads);
    const bool newton = inverse_mode() == 1;
#define CNF_MIX_LAUNCH(SPLIT_, REV_, KT_, NEWT_) \
    CNF_LAUNCH((mixture_kernel<SPLIT_, REV_, KT_, NEWT_>), grid, block, smem, st, a, tl)
#define CNF_MIX_K(SPLIT_, REV_, NEWT_)                                   \
    switch (kt) {                                                        \
        case 4: CNF_MIX_LAUNCH(SPLIT_, REV_, 4, NEWT_); break;           \
        case 8: CNF_MIX_LAUNCH(SPLIT_, REV_, 8, NEWT_); break;           \
        case 16: CNF_MIX_LAUNCH(SPLIT_, REV_, 16, NEWT_); break;         \
        default: CNF_MIX_LAUNCH(SPLIT_, REV_, 0, NEWT_); break;          \
    }
    if (split) {
        if (a.reverse) { if (newton) { CNF_MIX_K(true, true, true) } else { CNF_MIX_K(true, true, false) } }
        else { CNF_MIX_K(true, false, false) }
    } else {
        if (a.reverse) { if (newton) { CNF_MIX_K(false, true, true) } else { CNF_MIX_K(false, true, false) } }
        else { CNF_MIX_K(false, false, false) }
    }
#undef CNF_MIX_K
#undef CNF_MIX_LAUNCH
    return launch_status(who);
}

int cnf_mixture_coupling(const float* z, const float* nn_out,
                         const float* scaling_factor, const float* mixture_scaling_factor,
                         const float* mask, int mask_rows, int mask_cols,
                         const int* act_host, int n_act,
                         const float* pad, int pad_in_transform, int pad_output,
                         const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                         int B, int N, int D, int K, int reverse,
                         double reg_max, double reg_factor, int is_training,
                         int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out, "cnf_mixture_coupling: null tensor");
    MixArgs a = {};
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor;
    a.mask = mask; a.pad = pad; a.ldj_in = ldj_in; a.z_out = z_out; a.ldj_out = ldj_out;
    a.reg_out = reg_out; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask_rows; a.mc = mask_cols;
    a.reverse = reverse != 0;
    a.pad_in_transform = pad ? pad_in_transform : 0;
    a.pad_output = pad ? pad_output : 0;
    a.use_reg = (!reverse && reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    return launch_mixture(a, false, act_host, n_act, (hipStream_t)stream, "cnf_mixture_coupling");
}

static void split_workspace(MixArgs& a, void* workspace, int64_t workspace_bytes) {
    // [2B] int64 fixed-point row sums | [2B] fp64 escape words | [B] int32 tickets
    if (workspace && workspace_bytes >= (int64_t)a.B * 36 && (reinterpret_cast<uintptr_t>(workspace) & 7) == 0) {
        a.ws_acc = reinterpret_cast<long long*>(workspace);
        a.ws_big = reinterpret_cast<double*>(a.ws_acc + (size_t)2 * a.B);
        a.ws_cnt = reinterpret_cast<int*>(a.ws_acc + (size_t)4 * a.B);
    }
}

int64_t cnf_mixture_workspace_bytes(int B) { return B > 0 ? (int64_t)B * 36 : 0; }

void cnf_set_mixture_kernel(int which) {
    if (which == 0 || which == 1) g_mix_kernel = which;
}

void cnf_set_mixture_lanes(int lanes_per_item) {
    if (lanes_per_item == 0 || lanes_per_item == 1 || lanes_per_item == 2 || lanes_per_item == 4) g_mix_lanes = lanes_per_item;
}

void cnf_set_mixture_whole_tokens(int on) { set_mixture_whole_tokens(on); }
void cnf_set_mixture_nt_mb(int megabytes) { set_mixture_nt_mb(megabytes); }

void cnf_set_mixture_split(int waves) {
    if (waves >= 256 && waves <= 65536) set_mixture_split_waves(waves);
}

static int mixture_coupling_ws_impl(const char* who, int compact, const float* z, const float* nn_out,
                            const float* scaling_factor, const float* mixture_scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const int* act_host, int n_act,
                            const float* pad, int pad_in_transform, int pad_output,
                            const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                            int B, int N, int D, int K, int reverse,
                            double reg_max, double reg_factor, int is_training,
                            void* workspace, int64_t workspace_bytes,
                            int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out, "%s: null tensor", who);
    MixArgs a = {};
    a.compact = compact;
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor;
    a.mask = mask; a.pad = pad; a.ldj_in = ldj_in; a.z_out = z_out; a.ldj_out = ldj_out;
    a.reg_out = reg_out; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask_rows; a.mc = mask_cols;
    a.reverse = reverse != 0;
    a.pad_in_transform = pad ? pad_in_transform : 0;
    a.pad_output = pad ? pad_output : 0;
    a.use_reg = (!reverse && reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    split_workspace(a, workspace, workspace_bytes);
    return launch_mixture(a, false, act_host, n_act, (hipStream_t)stream, who);
}

int cnf_mixture_coupling_ws(const float* z, const float* nn_out,
                            const float* scaling_factor, const float* mixture_scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const int* act_host, int n_act,
                            const float* pad, int pad_in_transform, int pad_output,
                            const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                            int B, int N, int D, int K, int reverse,
                            double reg_max, double reg_factor, int is_training,
                            void* workspace, int64_t workspace_bytes,
                            int* flags, cnf_stream_t stream) {
    return mixture_coupling_ws_impl("cnf_mixture_coupling_ws", 0, z, nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows, mask_cols,
                                    act_host, n_act, pad, pad_in_transform, pad_output, ldj_in, z_out, ldj_out, reg_out, B, N, D, K, reverse,
                                    reg_max, reg_factor, is_training, workspace, workspace_bytes, flags, stream);
}

int cnf_mixture_coupling_compact(const float* z, const float* nn_compact,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad, int pad_in_transform, int pad_output,
                                 const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                 int B, int N, int D, int K, int reverse,
                                 double reg_max, double reg_factor, int is_training,
                                 void* workspace, int64_t workspace_bytes,
                                 int* flags, cnf_stream_t stream) {
    return mixture_coupling_ws_impl("cnf_mixture_coupling_compact", 1, z, nn_compact, scaling_factor, mixture_scaling_factor, mask, mask_rows,
                                    mask_cols, act_host, n_act, pad, pad_in_transform, pad_output, ldj_in, z_out, ldj_out, reg_out, B, N, D, K,
                                    reverse, reg_max, reg_factor, is_training, workspace, workspace_bytes, flags, stream);
}

static int mixture_coupling_nll_impl(const char* who, int compact, const float* z, const float* nn_out,
                             const float* scaling_factor, const float* mixture_scaling_factor,
                             const float* mask, int mask_rows, int mask_cols,
                             const int* act_host, int n_act,
                             const float* pad, int pad_in_transform, int pad_output,
                             const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                             const float* length, float* neglog_out, float* nll_out, int64_t* nll_acc,
                             int B, int N, int D, int K,
                             double reg_max, double reg_factor, int is_training,
                             float sigma, float log_sigma,
                             void* workspace, int64_t workspace_bytes,
                             int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out && nll_out, "%s: null tensor", who);
    MixArgs a = {};
    a.compact = compact;
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor;
    a.mask = mask; a.pad = pad; a.ldj_in = ldj_in; a.z_out = z_out; a.ldj_out = ldj_out;
    a.reg_out = reg_out; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask_rows; a.mc = mask_cols;
    a.reverse = 0;
    a.pad_in_transform = pad ? pad_in_transform : 0;
    a.pad_output = pad ? pad_output : 0;
    a.use_reg = (reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    a.length = length; a.neglog_out = neglog_out; a.nll_out = nll_out;
    a.nll_acc = reinterpret_cast<long long*>(nll_acc);
    a.prior = make_prior_const(sigma, log_sigma);
    split_workspace(a, workspace, workspace_bytes);
    return launch_mixture(a, false, act_host, n_act, (hipStream_t)stream, who);
}

int cnf_mixture_coupling_nll(const float* z, const float* nn_out,
                             const float* scaling_factor, const float* mixture_scaling_factor,
                             const float* mask, int mask_rows, int mask_cols,
                             const int* act_host, int n_act,
                             const float* pad, int pad_in_transform, int pad_output,
                             const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                             const float* length, float* neglog_out, float* nll_out, int64_t* nll_acc,
                             int B, int N, int D, int K,
                             double reg_max, double reg_factor, int is_training,
                             float sigma, float log_sigma,
                             void* workspace, int64_t workspace_bytes,
                             int* flags, cnf_stream_t stream) {
    return mixture_coupling_nll_impl("cnf_mixture_coupling_nll", 0, z, nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows, mask_cols,
                                     act_host, n_act, pad, pad_in_transform, pad_output, ldj_in, z_out, ldj_out, reg_out, length, neglog_out,
                                     nll_out, nll_acc, B, N, D, K, reg_max, reg_factor, is_training, sigma, log_sigma, workspace,
                                     workspace_bytes, flags, stream);
}

int cnf_mixture_coupling_compact_nll(const float* z, const float* nn_compact,
                                     const float* scaling_factor, const float* mixture_scaling_factor,
                                     const float* mask, int mask_rows, int mask_cols,
                                     const int* act_host, int n_act,
                                     const float* pad, int pad_in_transform, int pad_output,
                                     const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                     const float* length, float* neglog_out, float* nll_out, int64_t* nll_acc,
                                     int B, int N, int D, int K,
                                     double reg_max, double reg_factor, int is_training,
                                     float sigma, float log_sigma,
                                     void* workspace, int64_t workspace_bytes,
                                     int* flags, cnf_stream_t stream) {
    return mixture_coupling_nll_impl("cnf_mixture_coupling_compact_nll", 1, z, nn_compact, scaling_factor, mixture_scaling_factor, mask, mask_rows,
                                     mask_cols, act_host, n_act, pad, pad_in_transform, pad_output, ldj_in, z_out, ldj_out, reg_out, length,
                                     neglog_out, nll_out, nll_acc, B, N, D, K, reg_max, reg_factor, is_training, sigma, log_sigma, workspace,
                                     workspace_bytes, flags, stream);
}

static int mixture_coupling_actconv_impl(const char* who, int compact, const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad,
                                 const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                 const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                 const float* length,
                                 int B, int N, int D, int K,
                                 double reg_max, double reg_factor, int is_training,
                                 void* workspace, int64_t workspace_bytes,
                                 int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out && an_bias && an_scales && conv_weight && conv_sldj, "%s: null tensor", who);
    CNF_REQUIRE(D == 1 || D == 2 || D == 3 || D == 4 || D == 5 || D == 6 || D == 8,
                "%s: D=%d is outside the fused ActNorm + convolution kernels", who, D);
    MixArgs a = {};
    a.compact = compact;
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor;
    a.mask = mask; a.pad = pad; a.ldj_in = ldj_in; a.z_out = z_out; a.ldj_out = ldj_out;
    a.reg_out = reg_out; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask_rows; a.mc = mask_cols;
    a.reverse = 0;
    a.pad_in_transform = pad ? 1 : 0;
    a.pad_output = pad ? 1 : 0;
    a.use_reg = (reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    a.e_bias = an_bias; a.e_scales = an_scales; a.e_w = conv_weight; a.e_sldj = conv_sldj; a.e_length = length;
    split_workspace(a, workspace, workspace_bytes);
    return launch_mixture(a, false, act_host, n_act, (hipStream_t)stream, who);
}

int cnf_mixture_coupling_actconv(const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad,
                                 const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                 const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                 const float* length,
                                 int B, int N, int D, int K,
                                 double reg_max, double reg_factor, int is_training,
                                 void* workspace, int64_t workspace_bytes,
                                 int* flags, cnf_stream_t stream) {
    return mixture_coupling_actconv_impl("cnf_mixture_coupling_actconv", 0, z, nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows,
                                         mask_cols, act_host, n_act, pad, ldj_in, z_out, ldj_out, reg_out, an_bias, an_scales, conv_weight,
                                         conv_sldj, length, B, N, D, K, reg_max, reg_factor, is_training, workspace, workspace_bytes, flags,
                                         stream);
}

int cnf_mixture_coupling_compact_actconv(const float* z, const float* nn_compact,
                                         const float* scaling_factor, const float* mixture_scaling_factor,
                                         const float* mask, int mask_rows, int mask_cols,
                                         const int* act_host, int n_act,
                                         const float* pad,
                                         const float* ldj_in, float* z_out, float* ldj_out, float* reg_out,
                                         const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                         const float* length,
                                         int B, int N, int D, int K,
                                         double reg_max, double reg_factor, int is_training,
                                         void* workspace, int64_t workspace_bytes,
                                         int* flags, cnf_stream_t stream) {
    return mixture_coupling_actconv_impl("cnf_mixture_coupling_compact_actconv", 1, z, nn_compact, scaling_factor, mixture_scaling_factor, mask,
                                         mask_rows, mask_cols, act_host, n_act, pad, ldj_in, z_out, ldj_out, reg_out, an_bias, an_scales,
                                         conv_weight, conv_sldj, length, B, N, D, K, reg_max, reg_factor, is_training, workspace,
                                         workspace_bytes, flags, stream);
}

static int mixture_coupling_bwd_f32_impl(const char* who, int compact, const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad, int pad_in_transform, int pad_output,
                                 const float* g_zout, const float* g_ldj,
                                 float* g_z, float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor,
                                 float* workspace,
                                 int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                                 cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && g_z && g_nn && workspace, "%s: null tensor", who);
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && K > 0, "%s: bad shape", who);
    CNF_REQUIRE(!scaling_factor || g_scaling_factor, "%s: g_scaling_factor missing", who);
    CNF_REQUIRE(!mixture_scaling_factor || g_mixture_scaling_factor, "%s: g_mixture_scaling_factor missing", who);
    if (math_mode() == 1 && g_mix_kernel != 1 && D <= kMaxAct && (long)N * D < 65536) {
        MixArgs a = {};
        a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.msf = mixture_scaling_factor; a.mask = mask; a.pad = pad;
        a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask ? mask_rows : 1; a.mc = mask ? mask_cols : D;
        if (a.mr > N) a.mr = N;
        a.P = 2 + 3 * K; a.L = N * D;
        a.pad_in_transform = pad ? pad_in_transform : 0;
        a.pad_output = pad ? pad_output : 0;
        a.use_reg = (reg_max > 0 && is_training) ? 1 : 0;
        a.reg_max = reg_max; a.reg_factor = reg_factor;
        a.div_d = make_fastdiv((uint32_t)D);
        if (a.mr >= 1 && (a.mc == D || a.mc == 1)) {
            fill_act(a, act_host, n_act);
            a.div_da = make_fastdiv((uint32_t)a.DA);
            a.nn_D = D; a.nn_c0 = 0;
            a.compact = compact;
            const bool compact_ok = compact && mask && !a.per_item_mask && a.act_bits != 0;
            if (compact_ok) {
                a.nn_D = a.DA;
                a.nn_c0 = __builtin_ctzll(a.act_bits);
            }
            if ((!compact || compact_ok) &&
                launch_mixture_tok_bwd(a, g_zout, g_ldj, g_z, g_nn, g_scaling_factor, g_mixture_scaling_factor, workspace,
                                       (hipStream_t)stream, g_mix_lanes))
                return launch_status(who);
        }
    }
    if (compact) {
        // only the token-pass backward writes the compact layout: the caller expands nn_out / g_nn to the reference layout
        set_error("%s: shape / mode outside the compact-layout backward (B=%d N=%d D=%d K=%d)", who, B, N, D, K);
        return CNF_ERR_UNSUPPORTED;
    }
    return cnf_mixture_coupling_bwd(z, nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows, mask_cols, pad,
                                    pad_in_transform, pad_output, g_zout, g_ldj, g_z, g_nn, g_scaling_factor,
                                    g_mixture_scaling_factor, workspace, B, N, D, K, reg_max, reg_factor, is_training, stream);
}

int cnf_mixture_coupling_bwd_f32(const float* z, const float* nn_out,
                                 const float* scaling_factor, const float* mixture_scaling_factor,
                                 const float* mask, int mask_rows, int mask_cols,
                                 const int* act_host, int n_act,
                                 const float* pad, int pad_in_transform, int pad_output,
                                 const float* g_zout, const float* g_ldj,
                                 float* g_z, float* g_nn, float* g_scaling_factor, float* g_mixture_scaling_factor,
                                 float* workspace,
                                 int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                                 cnf_stream_t stream) {
    return mixture_coupling_bwd_f32_impl("cnf_mixture_coupling_bwd_f32", 0, z, nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows,
                                         mask_cols, act_host, n_act, pad, pad_in_transform, pad_output, g_zout, g_ldj, g_z, g_nn,
                                         g_scaling_factor, g_mixture_scaling_factor, workspace, B, N, D, K, reg_max, reg_factor, is_training,
                                         stream);
}

int cnf_mixture_coupling_compact_bwd_f32(const float* z, const float* nn_compact,
                                         const float* scaling_factor, const float* mixture_scaling_factor,
                                         const float* mask, int mask_rows, int mask_cols,
                                         const int* act_host, int n_act,
                                         const float* pad, int pad_in_transform, int pad_output,
                                         const float* g_zout, const float* g_ldj,
                                         float* g_z, float* g_nn_compact, float* g_scaling_factor, float* g_mixture_scaling_factor,
                                         float* workspace,
                                         int B, int N, int D, int K, double reg_max, double reg_factor, int is_training,
                                         cnf_stream_t stream) {
    return mixture_coupling_bwd_f32_impl("cnf_mixture_coupling_compact_bwd_f32", 1, z, nn_compact, scaling_factor, mixture_scaling_factor, mask,
                                         mask_rows, mask_cols, act_host, n_act, pad, pad_in_transform, pad_output, g_zout, g_ldj, g_z,
                                         g_nn_compact, g_scaling_factor, g_mixture_scaling_factor, workspace, B, N, D, K, reg_max,
                                         reg_factor, is_training, stream);
}

int cnf_mixture_transform(const double* z, const double* t, const double* log_s,
                          const double* log_pi, const double* mixt_t, const double* mixt_log_s,
                          const float* mask, int mask_rows, int mask_cols,
                          const int* act_host, int n_act, const float* pad,
                          double* z_out, double* ldj_out, double* reg_ldj,
                          int B, int N, int D, int K, int reverse,
                          double reg_max, double reg_factor, int is_training,
                          int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && t && log_s && log_pi && mixt_t && mixt_log_s && z_out && ldj_out,
                "cnf_mixture_transform: null tensor");
    MixArgs a = {};
    a.z64 = z; a.p_t = t; a.p_log_s = log_s; a.p_log_pi = log_pi; a.p_mu = mixt_t; a.p_ls = mixt_log_s;
    a.mask = mask; a.pad = pad; a.z_out64 = z_out; a.ldj_out64 = ldj_out; a.reg_elem64 = reg_ldj;
    a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.K = K; a.mr = mask_rows; a.mc = mask_cols;
    a.reverse = reverse != 0;
    a.pad_in_transform = pad ? 1 : 0;
    a.pad_output = 0;
    a.use_reg = (!reverse && reg_max > 0 && is_training) ? 1 : 0;
    a.reg_max = reg_max; a.reg_factor = reg_factor;
    return launch_mixture(a, true, act_host, n_act, (hipStream_t)stream, "cnf_mixture_transform");
}

int cnf_mixture_params(const float* nn_out, const float* scaling_factor,
                       const float* mixture_scaling_factor,
                       const float* mask, int mask_rows, int mask_cols,
                       double* t, double* log_s, double* log_pi, double* mixt_t, double* mixt_log_s,
                       int B, int N, int D, int K, cnf_stream_t stream) {
    CNF_REQUIRE(nn_out && t && log_s && log_pi && mixt_t && mixt_log_s, "cnf_mixture_params: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && K > 0, "cnf_mixture_params: bad shape");
    if (B == 0) return CNF_OK;
    if (mask && mask_rows > N) mask_rows = N;
    const long nelem = (long)B * N * D;
    const long total = nelem * (2 + 3 * K);
    const int grid = (int)std::min<long>((total + kBlock - 1) / kBlock, 256 * 16);
    CNF_LAUNCH(mixture_params_kernel, dim3(grid), dim3(kBlock), 0, (hipStream_t)stream,
                       nn_out, scaling_factor, mixture_scaling_factor, mask, mask_rows, mask_cols,
                       t, log_s, log_pi, mixt_t, mixt_log_s, nelem, N, D, K);
    return launch_status("cnf_mixture_params");
}

}  // extern "C"
