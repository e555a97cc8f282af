// Mixture-model categorical encoder (LinearCategoricalEncoding with num_flows == 0):
// forward = class-conditional affine push of logistic noise + the per-category log-prob over ALL
// classes (posterior) ; reverse = argmax decode.  linear_encoding.py:59-133,153-196.
//
// One lane per token.  The per-class flow (one ExtActNorm whose predictor input is the class
// embedding) collapses to a table [C, 2D]; its derived constants live in LDS:
//   bias[c][d], ts = tanh(scale_raw), e^{ts}, e^{-ts}, sum_d ts.
// The reference materialises [T*C, 1, D] tensors and runs an embedding + Linear for every class
// (:155-160); here the C x D loop runs in registers and the log-sum-exp is streamed.
#include "cnf_encoder.h"

#include <atomic>

#include <type_traits>

namespace cnf {



// logistic log-density pieces in base-2 units so that the hardware exp2 / log2 are used directly:
//   softplus(v) + softplus(-v) = |v| + 2 ln(1 + e^{-|v|}) = ln2 * (vs + 2 log2(1 + 2^{-vs})),  vs = |v| log2(e)
__device__ __forceinline__ float sp_pair2(float vs_abs) {
    return vs_abs + 2.f * __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-vs_abs));
}
// LogisticDistribution.log_prob (distributions.py:129-136,154-163), mu = 0
__device__ __forceinline__ float logistic_logp0(float x, float sigma, float log_sigma) {
    return -(kLn2 * sp_pair2(fabsf(x / sigma) * kLog2e) + log_sigma);
}

// LDS layout per class (stride 6D+3, odd -> no bank conflicts):
//   [bias D | ts D | e^ts D | e^-ts D | (A, C) pairs 2D | sum_ts | cst2 | E],
//   A = e^-ts log2e / sigma,  C = bias log2e / sigma  so that |z_back / sigma| log2e = |z A - C| is one FMA per
//   (class, channel);  cst2 = (prior - sum_ts - D log_sigma) log2e = the class score's constant in base-2 units;
//   E = 2^cst2 = the same constant as a FACTOR (the forward's posterior sums densities, see class_density).
__device__ __forceinline__ int class_stride(int D) { return 6 * D + 3; }
__device__ __forceinline__ void build_class_table(const EncArgs& a, float* tab) {
    const int stride = class_stride(a.D);
    const float k = kLog2e / a.sigma;
    for (int i = threadIdx.x; i < a.C * a.D; i += blockDim.x) {
        const int c = i / a.D, d = i - c * a.D;
        const float ts = tanhf(a.table[(size_t)c * 2 * a.D + a.D + d]);
        const float bias = a.table[(size_t)c * 2 * a.D + d];
        const float ems = expf(-ts);
        float* t = tab + c * stride;
        t[d] = bias;
        t[a.D + d] = ts;
        t[2 * a.D + d] = expf(ts);
        t[3 * a.D + d] = ems;
        t[4 * a.D + 2 * d] = ems * k;
        t[4 * a.D + 2 * d + 1] = bias * k;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += blockDim.x) {
        float* t = tab + c * stride;
        float s = 0.f;
        for (int d = 0; d < a.D; ++d) s += t[a.D + d];
        t[6 * a.D] = s;
        const float cst2 = ((a.prior[c] - s) - (float)a.D * a.log_sigma) * kLog2e;
        t[6 * a.D + 1] = cst2;
        t[6 * a.D + 2] = __builtin_amdgcn_exp2f(cst2);
    }
    __syncthreads();
}

// score of class j at point z (reverse flow log-prob + ldj + prior, linear_encoding.py:159-164) in BASE-2 units:
//   score log2e = cst2 - sum_d [vs_d + 2 log2(1 + 2^-vs_d)] = cst2 - sum_d vs_d - 2 log2 prod_d (1 + 2^-vs_d)
// one v_exp_f32 per (class, channel) and one v_log_f32 per class (the product of D <= 16 factors in [1, 2]
// cannot overflow)
template <int DT>
__device__ __forceinline__ float class_score2(const float* t, const float* z, int D) {
    float acc = 0.f, prod = 1.f;
    const int DD = DT > 0 ? DT : D;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
        const float vs = fabsf(fmaf(z[d], t[4 * D + 2 * d], -t[4 * D + 2 * d + 1]));
        acc += vs;
        prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);        // prod * (1 + 2^-vs) in one instruction
    }
    return t[6 * D + 1] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
}


// ---- the forward's posterior as a sum of DENSITIES (round 3) -----------------------------------------------------------
// class_prob_log = log_point - LSE_j(score_j) with the true class's entry replaced by the forward value log_point
// (linear_encoding.py:163-171), i.e.  -log(1 + sum_{j != c} e^{score_j - log_point}).  With q_jd = 2^-vs_jd the summand is
//   e^{score_j} = E_j prod_d q_jd / (prod_d (1 + q_jd))^2,      E_j = 2^cst2_j   (a logistic density is q / (sigma (1 + q)^2)),
// so the per-class work is the D exponentials the score needs anyway, two running products and ONE reciprocal: no
// logarithm per class and no exponential per class for the log-sum-exp (the streamed LSE spent 8.25 transcendental
// instructions per class at D = 6, this form 7, and 3 fewer plain ones; calibrated cost 118 -> 102 cycles per class and
// token).  Range: every summand is >= 0 and the true class contributes exactly 1, so an underflowing product only drops
// terms below 2^-126 (absolute) of the sum; relative to the total that is negligible while the token's own density 2^lp2 is
// well inside the fp32 range.  For a token whose own density is itself below ~2^-90 the other classes' densities — which may
// be comparable to it — can flush to zero in the products, and F = 2^-lp2 (or F * sum) eventually overflows: such a token
// takes the streamed log-sum-exp (lse2_of_classes), decided by density_sum_ok below (a wave-rare compare).
template <int DT>
__device__ __forceinline__ float class_density(const float* t, const float* z, int D) {
    float num = t[6 * D + 2], den = 1.f;
    const int DD = DT > 0 ? DT : D;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
        const float q = __builtin_amdgcn_exp2f(-fabsf(fmaf(z[d], t[4 * D + 2 * d], -t[4 * D + 2 * d + 1])));
        num *= q;
        den = fmaf(den, q, den);
    }
    const float r = __builtin_amdgcn_rcpf(den);
    return (num * r) * r;
}
// the cold path: base-2 log-sum-exp over the classes, the true class `c` at the forward value lp2 (the loop every token
// took in rounds 1-2); rolled, so that it adds no registers to its callers
template <int DT>
__device__ __forceinline__ float lse2_of_classes(const float* tab, int stride, const float* z, int D, int C, int c, float lp2) {
    float m = -3e38f, s = 0.f;
#pragma clang loop unroll(disable)
    for (int j = 0; j < C; ++j) {
        const float sc = class_score2<DT>(tab + j * stride, z, D);
        const float v = j == c ? lp2 : sc;
        const float mn = fmaxf(m, v);
        s = fmaf(s, __builtin_amdgcn_exp2f(m - mn), __builtin_amdgcn_exp2f(v - mn));
        m = mn;
    }
    return m + __builtin_amdgcn_logf(s);
}

// EPI: the ActNorm + 1x1 convolution of the flow step that follows the encoder (every flow of the reference starts with
// that pair) applied to the token's latents while they are in registers — activation_normalization.py:24-48,
// permutation_layers.py:106-136 in the arithmetic and order of actnorm_invconv_kernel (cnf_linear.hip), so the result is
// the chain's, bit for bit; the [B,N,D] round trip between the two kernels (8 B/elem) and a launch disappear.  The class
// posterior is taken at the encoder's own latents, as in the chain.
template <int DT, bool SAMPLE, bool EPI = false>
__global__ __launch_bounds__(kBlock) void encoder_forward_kernel(EncArgs a, RowTiling tl) {
    static_assert(!EPI || DT > 0, "the epilogue is built for the templated dimensions");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* part_all = reinterpret_cast<float*>(smem);
    float* tab = part_all + kWavesPerBlock * kMaxTileChunks;
    const int D = DT > 0 ? DT : a.D;
    const int stride = class_stride(D);
    // epilogue constants behind the class table: [bias D | e^scales D | W D*D | sum of the scales]
    float* etab = tab + (size_t)a.C * stride;
    if (EPI) {
        for (int i = threadIdx.x; i < D; i += kBlock) {
            etab[i] = a.e_bias[i];
            etab[D + i] = expf(a.e_scales[i]);
        }
        for (int i = threadIdx.x; i < D * D; i += kBlock) etab[2 * D + i] = a.e_w[i];
        if (threadIdx.x == 0) {
            float ssum = 0.f;
            for (int i = 0; i < D; ++i) ssum += a.e_scales[i];
            etab[2 * D + D * D] = ssum;
        }
    }
    build_class_table(a, tab);          // ends with a barrier: the epilogue constants are visible as well
    bool bad = false;

    auto chunk = [&](int row, int n) -> float {
        const size_t tok = (size_t)row * a.N + n;
        // an index outside [0, C) would read past the class table: clamp it and report it (the reference asserts in
        // one_hot, general/mutils.py:264)
        const long long craw = a.categ[tok];
        if (craw < 0 || craw >= a.C) raise_flag(a.flags, CNF_FLAG_CATEGORY);
        const int c = (int)(craw < 0 ? 0 : (craw >= a.C ? a.C - 1 : craw));
        const float* tc = tab + c * stride;
        float z[DT > 0 ? DT : kEncMaxD];
        // log-prob of the noise under the logistic prior: same product form as class_score
        float nacc = 0.f, nprod = 1.f;
        const float kn = kLog2e / a.sigma;
        float ev[DT > 0 ? DT : kEncMaxD];
        enc_token_noise<SAMPLE ? 1 : 0>(a, tok, D, ev);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float e = ev[d];
            const float vs = fabsf(e) * kn;
            nacc += vs;
            nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
            z[d] = (e + tc[d]) * tc[2 * D + d];
        }
        const float init_lp = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * a.log_sigma);
        const float ldj_f = tc[6 * D];
        const float log_point = (init_lp - ldj_f) + a.prior[c];
        const float lp2 = log_point * kLog2e;
#ifdef CNF_ENC_LSE
        // A/B build: the streamed log-sum-exp of rounds 1-2 for every token
        const float cpl = (lp2 - lse2_of_classes<DT>(tab, stride, z, D, a.C, c, lp2)) * kLn2;
#else
        // densities of the other classes relative to the token's own (class_density above), branch-free: every lane
        // scores every class and the true class's summand is dropped (it is the 1 of `tot`, :167-168)
        float dsum = 0.f;
        for (int j = 0; j < a.C; ++j) {
            const float dj = class_density<DT>(tab + j * stride, z, D);
            dsum += j == c ? 0.f : dj;
        }
        const float tot = fmaf(__builtin_amdgcn_exp2f(-lp2), dsum, 1.f);
        float cpl;
        if (density_sum_ok(tot, lp2)) cpl = -kLn2 * __builtin_amdgcn_logf(tot);
        else cpl = (lp2 - lse2_of_classes<DT>(tab, stride, z, D, a.C, c, lp2)) * kLn2;       // overflow / NaN: log domain
#endif
        const float pv = a.pad ? a.pad[tok] : 1.f;
        if (a.cpl) a.cpl[tok] = cpl;
        if (EPI) {
            float xv[DT > 0 ? DT : 1];
#pragma unroll
            for (int i = 0; i < DT; ++i) {
                float y = (z[i] * pv + etab[i]) * etab[DT + i];
                if (a.pad) y = y * pv;
                xv[i] = y;
            }
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < DT; ++i) acc = fmaf(xv[i], etab[2 * DT + i * DT + j], acc);
                if (a.pad) acc = acc * pv;
                bad |= isnan(acc);
                a.z_out[tok * D + j] = acc;
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float o = z[d] * pv;
                bad |= isnan(o);
                a.z_out[tok * D + d] = o;
            }
        }
        return (a.beta * cpl - (init_lp - ldj_f)) * pv;
    };
    auto finish = [&](int row, float sum) {
        float v = (a.ldj_in ? a.ldj_in[row] : 0.f) + sum;
        if (EPI) {
            // log-det of the two layers, same association as run in sequence: ActNorm uses length | sum(pad) | N, the
            // convolution length | N (actnorm_invconv_kernel)
            float len_a, len_c;
            if (a.e_length) {
                len_a = len_c = a.e_length[row];
            } else {
                len_c = (float)a.N;
                len_a = (float)a.N;
                if (a.pad) {
                    len_a = 0.f;
                    for (int n = 0; n < a.N; ++n) len_a += a.pad[(size_t)row * a.N + n];
                }
            }
            v = (v + etab[2 * D + D * D] * len_a) + a.e_sldj[0] * len_c;
        }
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    };
    walk_row_tile<float>(tl, part_all + (threadIdx.x >> 6) * kMaxTileChunks, chunk, finish);
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// EPI: the sampling direction's last three layers in one launch — the inverse 1x1 convolution and the inverse ActNorm of
// the first flow step (permutation_layers.py:106-136, activation_normalization.py:24-48 with reverse = True) applied to the
// token's latents as they are loaded, arithmetic and order of actnorm_invconv_kernel (cnf_linear.hip, reverse), then the
// arg-max decode; the running log-det gets the two layers' terms (the decode itself adds zero, linear_encoding.py:108-118).
template <int DT, bool EPI = false>
__global__ __launch_bounds__(kBlock) void encoder_decode_kernel(EncArgs a, long ntok) {
    static_assert(!EPI || DT > 0, "the prologue is built for the templated dimensions");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem);
    const int D = DT > 0 ? DT : a.D;
    const int stride = class_stride(D);
    float* etab = tab + (size_t)a.C * stride;          // [bias D | e^-scales D | W^-1 D*D | sum of the scales]
    if (EPI) {
        for (int i = threadIdx.x; i < D; i += kBlock) {
            etab[i] = a.e_bias[i];
            etab[D + i] = expf(-a.e_scales[i]);
        }
        for (int i = threadIdx.x; i < D * D; i += kBlock) etab[2 * D + i] = a.e_w[i];
        if (threadIdx.x == 0) {
            float ssum = 0.f;
            for (int i = 0; i < D; ++i) ssum += a.e_scales[i];
            etab[2 * D + D * D] = ssum;
        }
    }
    build_class_table(a, tab);          // ends with a barrier
    bool bad = false;
    for (long tok = (long)blockIdx.x * kBlock + threadIdx.x; tok < ntok; tok += (long)gridDim.x * kBlock) {
        float z[DT > 0 ? DT : kEncMaxD];
#pragma unroll
        for (int d = 0; d < D; ++d) z[d] = a.z_in[tok * D + d];
        if (EPI) {
            const float p = a.pad ? a.pad[tok] : 1.f;
            float xv[DT > 0 ? DT : 1];
#pragma unroll
            for (int i = 0; i < DT; ++i) xv[i] = z[i];
#pragma unroll
            for (int j = 0; j < DT; ++j) {
                float acc = 0.f;
#pragma unroll
                for (int i = 0; i < DT; ++i) acc = fmaf(xv[i], etab[2 * DT + i * DT + j], acc);
                if (a.pad) acc = acc * p;
                acc = acc * etab[DT + j] - etab[j];
                if (a.pad) acc = acc * p;
                bad |= isnan(acc);
                z[j] = acc;
            }
        }
        float best = -INFINITY;
        int arg = 0;
        for (int j = 0; j < a.C; ++j) {
            const float v = class_score2<DT>(tab + j * stride, z, D);
            if (j == 0 || v > best) {   // first maximum wins, like torch.argmax
                best = v;
                arg = j;
            }
        }
        a.categ_out[tok] = (int64_t)arg;
    }
    if (EPI) {
        if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
        const float ssum = etab[2 * D + D * D], sl = a.e_sldj[0];
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
            float len_a, len_c;
            if (a.e_length) {
                len_a = len_c = a.e_length[b];
            } else {
                len_c = (float)a.N;
                if (a.pad) {
                    len_a = 0.f;
                    for (int n = 0; n < a.N; ++n) len_a += a.pad[b * a.N + n];
                } else len_a = (float)a.N;
            }
            const float base = a.ldj_in ? a.ldj_in[b] : 0.f;
            // the two layers in reverse, same association as actnorm_invconv_kernel; + 0: the decode's own (zero) term
            const float v = ((base - sl * len_c) + (-ssum) * len_a) + 0.f;
            a.ldj_out[b] = v;
            if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        }
    }
}


// ---- round 3: two tokens per lane ------------------------------------------------------------------------------------
//
// Calibration on the device (tools/microbench/enc_micro.hip, profiles/r03_valu_calibration.txt): a plain wave64 VALU
// instruction occupies its SIMD for 2 cycles, v_exp_f32 / v_log_f32 for 8, and ONE wave issues at most one VALU
// instruction per ~5.3 cycles.  The class loop is 3 plain + 1 transcendental instruction per (class, channel): 14 cycles
// per wave, of which the exp2 alone is 8 — the loop's slope per class already sits at that figure, so what a kernel can
// still save is what is NOT the (class, channel) arithmetic:
//   * a lane scores TWO consecutive tokens against each class, so the class constants are read once per pair, as 16-byte
//     LDS vectors ([A0 C0 A1 C1 ... cst2] padded to a multiple of four floats): 4 ds_read_b128 per class and pair instead
//     of 13 ds_read2_b32 per two classes and ONE token, and two independent product chains per class;
//   * a pair's latents / noise are 8D contiguous bytes: 16-byte nontemporal loads (D even) instead of D dword loads at a
//     4D-byte lane stride; the decoded indices of a pair are one 16-byte store;
//   * the derived table is built with ONE tanhf per (class, channel) (the per-class sum reads the values back from LDS;
//     round 2 evaluated every tanhf twice, the second time D of them in one serial thread), and a thread's raw table
//     entries are loaded BEFORE its latents: vmcnt retires in order, so a load issued behind the latents could not be
//     waited for without them and the build would start only after the whole first burst had arrived.
// Arithmetic per token is unchanged (same operations in the same order as class_score2 / encoder_forward_kernel), so
// latents, class posteriors, log-det terms and decoded indices are bit-identical to the round-2 kernels, which stay as
// the fallback for shapes this layout does not take (block-per-row tilings, odd tile starts, unaligned views, D > 8).
typedef float vf4 __attribute__((ext_vector_type(4)));
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef long long ll2 __attribute__((ext_vector_type(2)));

template <int D>
struct PairIO {
    static constexpr bool kWide = (D % 2 == 0);            // 8D bytes per pair: a multiple of 16 iff D is even
    static constexpr int kN = kWide ? D / 2 : D;           // vectors per pair
    using V = typename std::conditional<kWide, vf4, vf2>::type;
    static constexpr int kW = kWide ? 4 : 2;
    // both tokens of the pair at p (2D floats, aligned to the vector width)
    static __device__ __forceinline__ void load_nt(const float* p, float (&x)[2][D]) {
        const V* src = reinterpret_cast<const V*>(p);
        float buf[2 * D];
#pragma unroll
        for (int q = 0; q < kN; ++q) {
            const V v = __builtin_nontemporal_load(src + q);
#pragma unroll
            for (int w = 0; w < kW; ++w) buf[kW * q + w] = v[w];
        }
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) x[i / D][i % D] = buf[i];
    }
    static __device__ __forceinline__ void store(float* p, const float (&x)[2][D]) {
        V* dst = reinterpret_cast<V*>(p);
        float buf[2 * D];
#pragma unroll
        for (int i = 0; i < 2 * D; ++i) buf[i] = x[i / D][i % D];
#pragma unroll
        for (int q = 0; q < kN; ++q) {
            V v;
#pragma unroll
            for (int w = 0; w < kW; ++w) v[w] = buf[kW * q + w];
            dst[q] = v;
        }
    }
};

template <int D>
struct PairTab {
    static constexpr int S = (2 * D + 1 + 3) / 4 * 4;      // score constants per class: [A0 C0 ... A(D-1) C(D-1) cst2 E pad]; E = 2^cst2 (2D + 1 is odd: the slot exists)
    static constexpr int F = 3 * D + 2;                    // forward constants per class: [bias D | e^ts D | ts D | sum_ts | prior]
};

// first half of the table build: this thread's raw entries (issued before the latents' loads)
template <int D>
__device__ __forceinline__ void pair_table_load(const EncArgs& a, float& r_b, float& r_s) {
    r_b = 0.f;
    r_s = 0.f;
    const int i = threadIdx.x;
    if (i < a.C * D) {
        const int c = i / D, d = i - c * D;
        r_b = a.table[(size_t)c * 2 * D + d];
        r_s = a.table[(size_t)c * 2 * D + D + d];
    }
}
// second half.  FWD: the forward constants as well; otherwise `ft` holds the tanh values only (stride D).
// Same expressions as build_class_table, so every constant has the same bits.
template <int D, bool FWD>
__device__ __forceinline__ void pair_table_finish(const EncArgs& a, float* vt, float* ft, float r_b, float r_s) {
    constexpr int S = PairTab<D>::S, F = FWD ? PairTab<D>::F : D, TS = FWD ? 2 * D : 0;
    const float k = kLog2e / a.sigma;
    for (int i = threadIdx.x; i < a.C * D; i += kBlock) {
        const int c = i / D, d = i - c * D;
        if (i != (int)threadIdx.x) {
            r_b = a.table[(size_t)c * 2 * D + d];
            r_s = a.table[(size_t)c * 2 * D + D + d];
        }
        const float ts = tanhf(r_s);
        const float ems = expf(-ts);
        vt[c * S + 2 * d] = ems * k;
        vt[c * S + 2 * d + 1] = r_b * k;
        ft[c * F + TS + d] = ts;
        if (FWD) {
            ft[c * F + d] = r_b;
            ft[c * F + D + d] = expf(ts);
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < a.C; c += kBlock) {
        float s = 0.f;
        for (int d = 0; d < D; ++d) s += ft[c * F + TS + d];
        const float pr = a.prior[c];
        const float cst2 = ((pr - s) - (float)D * a.log_sigma) * kLog2e;
        vt[c * S + 2 * D] = cst2;
        vt[c * S + 2 * D + 1] = __builtin_amdgcn_exp2f(cst2);
        if (FWD) {
            ft[c * F + 3 * D] = s;
            ft[c * F + 3 * D + 1] = pr;
        }
    }
    __syncthreads();
}

// scores of class j for both tokens of a pair (class_score2's arithmetic, constants from one row of 16-byte vectors)
template <int D>
__device__ __forceinline__ void pair_score(const float4* vt4, int j, const float (&z)[2][D], float (&sc)[2]) {
    constexpr int S = PairTab<D>::S;
    float k[S];
#pragma unroll
    for (int q = 0; q < S / 4; ++q) {
        const float4 v = vt4[j * (S / 4) + q];
        k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float acc = 0.f, prod = 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float vs = fabsf(fmaf(z[t][d], k[2 * d], -k[2 * d + 1]));
            acc += vs;
            prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
        }
        sc[t] = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
    }
}

// densities of class j at both tokens of a pair (class_density's arithmetic on the 16-byte rows)
template <int D>
__device__ __forceinline__ void pair_density(const float4* vt4, int j, const float (&z)[2][D], float (&dn)[2]) {
    constexpr int S = PairTab<D>::S;
    float k[S];
#pragma unroll
    for (int q = 0; q < S / 4; ++q) {
        const float4 v = vt4[j * (S / 4) + q];
        k[4 * q] = v.x; k[4 * q + 1] = v.y; k[4 * q + 2] = v.z; k[4 * q + 3] = v.w;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        float num = k[2 * D + 1], den = 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float q = __builtin_amdgcn_exp2f(-fabsf(fmaf(z[t][d], k[2 * d], -k[2 * d + 1])));
            num *= q;
            den = fmaf(den, q, den);
        }
        const float r = __builtin_amdgcn_rcpf(den);
        dn[t] = (num * r) * r;
    }
}
// cold path of the pair forward: streamed base-2 log-sum-exp of ONE token (lse2_of_classes on the 16-byte rows)
template <int D>
__device__ __forceinline__ float pair_lse2(const float* vt, const float (&z)[D], int C, int c, float lp2) {
    constexpr int S = PairTab<D>::S;
    float m = -3e38f, s = 0.f;
#pragma clang loop unroll(disable)
    for (int j = 0; j < C; ++j) {
        const float* k = vt + j * S;
        float acc = 0.f, prod = 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float vs = fabsf(fmaf(z[d], k[2 * d], -k[2 * d + 1]));
            acc += vs;
            prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
        }
        const float sc = k[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
        const float v = j == c ? lp2 : sc;
        const float mn = fmaxf(m, v);
        s = fmaf(s, __builtin_amdgcn_exp2f(m - mn), __builtin_amdgcn_exp2f(v - mn));
        m = mn;
    }
    return m + __builtin_amdgcn_logf(s);
}

// arg-max decode (linear_encoding.py:184-196): one tile of 128 consecutive tokens per wave, lane i owns tokens 2i, 2i+1
template <int D>
__global__ __launch_bounds__(kBlock) void encoder_decode_pair_kernel(EncArgs a, long ntok) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = PairTab<D>::S;
    float* vt = reinterpret_cast<float*>(smem);
    float* ts_sh = vt + a.C * S;
    const int lane = threadIdx.x & 63;
    const long base = ((long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)) * 128 + 2 * lane;
    float r_b, r_s;
    pair_table_load<D>(a, r_b, r_s);
    float z[2][D];
    const bool full = base + 1 < ntok;
    if (full) {
        PairIO<D>::load_nt(a.z_in + base * D, z);
    } else {
        const long t0 = min(base, ntok - 1);                 // a lone last token (or a lane past the end): scored twice
#pragma unroll
        for (int d = 0; d < D; ++d) z[0][d] = z[1][d] = a.z_in[t0 * D + d];
    }
    pair_table_finish<D, false>(a, vt, ts_sh, r_b, r_s);
    if (base >= ntok) return;
    const float4* vt4 = reinterpret_cast<const float4*>(smem);
    float best[2] = {-INFINITY, -INFINITY};
    int arg[2] = {0, 0};
    for (int j = 0; j < a.C; ++j) {
        float sc[2];
        pair_score<D>(vt4, j, z, sc);
#pragma unroll
        for (int t = 0; t < 2; ++t)
            if (sc[t] > best[t]) {                          // first maximum wins, like torch.argmax (class 0 holds a tie at -inf)
                best[t] = sc[t];
                arg[t] = j;
            }
    }
    if (full) {
        ll2 v = {(long long)arg[0], (long long)arg[1]};
        *reinterpret_cast<ll2*>(a.categ_out + base) = v;
    } else {
        a.categ_out[base] = (int64_t)arg[0];
    }
}

// forward (linear_encoding.py:59-133,153-174) on the row tiling of encoder_forward_kernel (a wave owns tl.rw whole rows;
// requires !tl.bpr, tl.rw >= 2 and an even number of tokens per tile, so that every pair starts at an even token):
// lane i owns the tile's tokens 2i, 2i+1, then 2i+128, 2i+129, ...; the next pair's inputs are in flight while the
// current pair is scored.  Token log-det terms go to the wave's LDS strip and are summed per row exactly as
// walk_row_tile_split does.
template <int D>
__global__ __launch_bounds__(kBlock) void encoder_forward_pair_kernel(EncArgs a, RowTiling tl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int S = PairTab<D>::S, F = PairTab<D>::F;
    float* part = reinterpret_cast<float*>(smem) + (threadIdx.x >> 6) * kMaxTileChunks;
    float* vt = reinterpret_cast<float*>(smem) + kWavesPerBlock * kMaxTileChunks;
    float* ft = vt + a.C * S;
    const float4* vt4 = reinterpret_cast<const float4*>(vt);
    const int lane = threadIdx.x & 63;
    const long tile = (long)walker_block() * kWavesPerBlock + (threadIdx.x >> 6);
    const bool live = tile < tl.ntiles;
    const int row0 = live ? (int)(tile * tl.rw) : 0;
    const int nrows = live ? min(tl.rw, tl.B - row0) : 0;
    const int nch = nrows * a.N;                            // tokens of this tile
    const long tok0 = (long)row0 * a.N;

    struct In {
        float e[2][D];
        long long c[2];
        float pv[2];
    };
    auto load = [&](int pc, In& in) {
        const long tok = tok0 + pc;
        if (pc + 1 < nch) {
            PairIO<D>::load_nt(a.eps + tok * D, in.e);
            if (a.eps_is_u) {
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int d = 0; d < D; ++d) in.e[t][d] = enc_noise_from_uniform(in.e[t][d], a.sigma, a.u_squeeze);
                if (a.eps_out) PairIO<D>::store(a.eps_out + tok * D, in.e);
            }
            const ll2 cc = *reinterpret_cast<const ll2*>(a.categ + tok);
            in.c[0] = cc[0];
            in.c[1] = cc[1];
            if (a.pad) {
                const vf2 p = *reinterpret_cast<const vf2*>(a.pad + tok);
                in.pv[0] = p[0];
                in.pv[1] = p[1];
            } else {
                in.pv[0] = in.pv[1] = 1.f;
            }
        } else {                                            // lone last token of the tile: its partner is a copy
            enc_token_noise<2>(a, (size_t)tok, D, in.e[0]);
#pragma unroll
            for (int d = 0; d < D; ++d) in.e[1][d] = in.e[0][d];
            in.c[0] = in.c[1] = a.categ[tok];
            in.pv[0] = in.pv[1] = a.pad ? a.pad[tok] : 1.f;
        }
    };
    float r_b, r_s;
    pair_table_load<D>(a, r_b, r_s);
    In nxt;
    if (2 * lane < nch) load(2 * lane, nxt);
    pair_table_finish<D, true>(a, vt, ft, r_b, r_s);
    bool bad = false;
    const float kn = kLog2e / a.sigma;
    for (int pc = 2 * lane; pc < nch; pc += 128) {
        const In cur = nxt;
        if (pc + 128 < nch) load(pc + 128, nxt);
        const bool two = pc + 1 < nch;
        float z[2][D], init_lp[2], ldj_f[2], lp2[2];
        int c[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            // an index outside [0, C) would read past the class table: clamp it and report it (general/mutils.py:264)
            const long long craw = cur.c[t];
            if (craw < 0 || craw >= a.C) raise_flag(a.flags, CNF_FLAG_CATEGORY);
            c[t] = (int)(craw < 0 ? 0 : (craw >= a.C ? a.C - 1 : craw));
            const float* tc = ft + c[t] * F;
            float nacc = 0.f, nprod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float e = cur.e[t][d];
                const float vs = fabsf(e) * kn;
                nacc += vs;
                nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
                z[t][d] = (e + tc[d]) * tc[D + d];
            }
            init_lp[t] = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * a.log_sigma);
            ldj_f[t] = tc[3 * D];
            const float log_point = (init_lp[t] - ldj_f[t]) + tc[3 * D + 1];
            lp2[t] = log_point * kLog2e;
        }
        // the other classes' densities relative to the token's own, as in encoder_forward_kernel
        float dsum[2] = {0.f, 0.f};
#ifndef CNF_ENC_LSE
        for (int j = 0; j < a.C; ++j) {
            float dn[2];
            pair_density<D>(vt4, j, z, dn);
#pragma unroll
            for (int t = 0; t < 2; ++t) dsum[t] += j == c[t] ? 0.f : dn[t];
        }
#endif
        float cpl[2], zo[2][D];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#ifdef CNF_ENC_LSE
            const float tot = INFINITY;
#else
            const float tot = fmaf(__builtin_amdgcn_exp2f(-lp2[t]), dsum[t], 1.f);
#endif
            if (density_sum_ok(tot, lp2[t])) cpl[t] = -kLn2 * __builtin_amdgcn_logf(tot);
            else cpl[t] = (lp2[t] - pair_lse2<D>(vt, z[t], a.C, c[t], lp2[t])) * kLn2;      // overflow / NaN: log domain
#pragma unroll
            for (int d = 0; d < D; ++d) {
                zo[t][d] = z[t][d] * cur.pv[t];
                bad |= isnan(zo[t][d]);
            }
            if (t == 0 || two) part[pc + t] = (a.beta * cpl[t] - (init_lp[t] - ldj_f[t])) * cur.pv[t];
        }
        const long tok = tok0 + pc;
        if (two) {
            PairIO<D>::store(a.z_out + tok * D, zo);
            if (a.cpl) {
                vf2 cv = {cpl[0], cpl[1]};
                *reinterpret_cast<vf2*>(a.cpl + tok) = cv;
            }
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) a.z_out[tok * D + d] = zo[0][d];
            if (a.cpl) a.cpl[tok] = cpl[0];
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    if (!live) return;
    // per-row sums: the reduction of walk_row_tile_split (rw > 1), chunk = token
    wave_lds_sync();
    const int g = kWave / tl.p2;
    const int sub = lane & (g - 1);
    for (int r0 = 0; r0 < nrows; r0 += tl.p2) {
        const int r = r0 + lane / g;
        float acc = 0.f;
        if (r < nrows)
            for (int i = sub; i < tl.cpr; i += g) acc += part[r * tl.cpr + i];
        acc = group_sum(acc, g);
        if (sub == 0 && r < nrows) {
            const float v = (a.ldj_in ? a.ldj_in[row0 + r] : 0.f) + acc;
            a.ldj_out[row0 + r] = v;
            if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        }
    }
}

static std::atomic<int> g_encoder_kernel{0};     // cnf_set_encoder_kernel: 0 = by measurement (below: the one-token kernels), 1 = one-token kernels, 2 = pair kernels wherever eligible


// ---- large vocabularies: the class table does not fit LDS, so it is walked in chunks ---------------------------------
//
// A workgroup owns 256 consecutive tokens per round (one per thread) and walks the classes in chunks of CC: the chunk's
// score constants (A, C pairs and cst2, 2D + 1 floats per class) are built cooperatively in LDS from the raw [C, 2D]
// table (tanhf / expf per (class, channel): 1 / 256 of the scoring work that follows), every thread scores the chunk
// against its token and keeps the streamed log-sum-exp (forward) or the running arg-max (decode).  The reference
// materialises [T * C, 1, D] tensors for this (linear_encoding.py:155-160) — 13 GB per tensor at 32 k tokens and 10 k
// classes; here nothing of that size exists.  Token log-det terms go to a [B * N] buffer; a second small kernel sums
// the rows in a fixed order.
template <int DT>
__device__ __forceinline__ float chunk_score2(const float* t, const float* z, int D) {
    float acc = 0.f, prod = 1.f;
    const int DD = DT > 0 ? DT : D;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
        const float vs = fabsf(fmaf(z[d], t[2 * d], -t[2 * d + 1]));
        acc += vs;
        prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
    }
    return t[2 * D] - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
}
// class_density on a chunk row
template <int DT>
__device__ __forceinline__ float chunk_density(const float* t, const float* z, int D) {
    float num = t[2 * D + 1], den = 1.f;
    const int DD = DT > 0 ? DT : D;
#pragma unroll
    for (int d = 0; d < DD; ++d) {
        const float q = __builtin_amdgcn_exp2f(-fabsf(fmaf(z[d], t[2 * d], -t[2 * d + 1])));
        num *= q;
        den = fmaf(den, q, den);
    }
    const float r = __builtin_amdgcn_rcpf(den);
    return (num * r) * r;
}
// Cold path of the tiled forward (a token whose density sum left the fp32 range, see class_density): the streamed base-2
// log-sum-exp over ALL classes straight from the raw [C, 2D] table — no LDS chunk, no barrier, so a single lane can
// take it; tanhf / expf per (class, channel) make it ~20x the cost of a chunk sweep, for tokens that essentially do
// not occur (own density below ~2^-100).
__device__ __noinline__ float lse2_from_raw_table(const float* table, const float* prior, const float* z, int D, int C, int c,
                                                  float lp2, float sigma, float log_sigma) {
    float m = -3e38f, s = 0.f;
    const float k = kLog2e / sigma;
    for (int j = 0; j < C; ++j) {
        const float* row = table + (size_t)j * 2 * D;
        float acc = 0.f, prod = 1.f, tsum = 0.f;
        for (int d = 0; d < D; ++d) {
            const float ts = tanhf(row[D + d]);
            const float vs = fabsf(fmaf(z[d], expf(-ts) * k, -(row[d] * k)));
            acc += vs;
            prod = fmaf(prod, __builtin_amdgcn_exp2f(-vs), prod);
            tsum += ts;
        }
        const float sc = ((prior[j] - tsum) - (float)D * log_sigma) * kLog2e - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
        const float v = j == c ? lp2 : sc;
        const float mn = fmaxf(m, v);
        s = fmaf(s, __builtin_amdgcn_exp2f(m - mn), __builtin_amdgcn_exp2f(v - mn));
        m = mn;
    }
    return m + __builtin_amdgcn_logf(s);
}

// PHASE 0: the whole class range and the epilogue in one kernel.  Very large vocabularies with moderate token counts
// (10^4 classes x 10^4..10^5 tokens: 40-400 token workgroups) would leave most of the chip idle, so the class range is
// also split over blockIdx.y: PHASE 1 sweeps one split and writes the token's partial (density sum, -) / (best, arg-max)
// to part[split][tok]; PHASE 2 (token lanes again, no class sweep) merges the partials in split order and runs the
// epilogue.  The forward sums the other classes' densities (class_density: absolute values, so the partials of the
// splits simply add up) and scales by 2^-lp2 at the end.  The number of splits depends on C only, so a sample's result does not depend on the batch it is in.
template <int DT, bool DECODE, int PHASE>
__global__ __launch_bounds__(kBlock) void encoder_tiled_kernel(EncArgs a, long ntok, int CC, float* tok_ldj, float* part, int KS) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem);
    const int D = DT > 0 ? DT : a.D;
    const int stride = chunk_stride(D);
    bool bad = false;
    const int per = (a.C + KS - 1) / KS;
    const int j_lo = PHASE == 1 ? (int)blockIdx.y * per : 0;
    const int j_hi = PHASE == 1 ? min(a.C, j_lo + per) : a.C;
    const long rounds = (ntok + kBlock - 1) / kBlock;
    for (long r = blockIdx.x; r < rounds; r += gridDim.x) {             // block-uniform: barriers inside are safe
        const long tok = r * kBlock + threadIdx.x;
        const bool live = tok < ntok;
        float z[DT > 0 ? DT : kEncMaxD];
        int c = 0;
        float lp2 = 0.f, init_lp = 0.f, ldj_f = 0.f;
        if (live && DECODE) {
#pragma unroll
            for (int d = 0; d < D; ++d) z[d] = a.z_in[tok * D + d];
        } else if (live) {
            const long long craw = a.categ[tok];
            if (craw < 0 || craw >= a.C) raise_flag(a.flags, CNF_FLAG_CATEGORY);
            c = (int)(craw < 0 ? 0 : (craw >= a.C ? a.C - 1 : craw));
            const float* row = a.table + (size_t)c * 2 * D;
            float nacc = 0.f, nprod = 1.f;
            const float kn = kLog2e / a.sigma;
            float ev[DT > 0 ? DT : kEncMaxD];
            enc_token_noise<2, (DT > 0 ? DT : kEncMaxD), PHASE != 1>(a, (size_t)tok, D, ev);
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float e = ev[d];
                const float vs = fabsf(e) * kn;
                nacc += vs;
                nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
                const float ts = tanhf(row[D + d]);
                z[d] = (e + row[d]) * expf(ts);
                ldj_f += ts;
            }
            init_lp = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * a.log_sigma);
            lp2 = ((init_lp - ldj_f) + a.prior[c]) * kLog2e;
        } else {
#pragma unroll
            for (int d = 0; d < D; ++d) z[d] = 0.f;
        }
        float dsum = 0.f, best = -INFINITY;
        int arg = 0;
        if (PHASE != 2) {
            bool first = true;
            for (int j0 = j_lo; j0 < j_hi; j0 += CC) {
                const int cc = min(CC, j_hi - j0);
                __syncthreads();                               // the previous chunk has been read by everyone
                build_class_chunk(a, tab, j0, cc, D);
                __syncthreads();
                for (int jj = 0; jj < cc; ++jj) {
                    if (DECODE) {
                        const float sc = chunk_score2<DT>(tab + jj * stride, z, D);
                        if (first || sc > best) {               // first maximum wins, like torch.argmax
                            best = sc;
                            arg = j0 + jj;
                            first = false;
                        }
                    } else {
                        const float dj = chunk_density<DT>(tab + jj * stride, z, D);
                        dsum += (j0 + jj) == c ? 0.f : dj;      // the true class is the 1 of the total (:167-168)
                    }
                }
            }
        } else if (live) {
            for (int k = 0; k < KS; ++k) {                     // split order: deterministic, ties go to the lower class
                const float p0 = part[((size_t)k * ntok + tok) * 2], p1 = part[((size_t)k * ntok + tok) * 2 + 1];
                if (DECODE) {
                    if (k == 0 || p0 > best) {
                        best = p0;
                        arg = __float_as_int(p1);
                    }
                } else {
                    dsum += p0;
                }
            }
        }
        if (!live) continue;
        if (PHASE == 1) {
            float* o = part + ((size_t)blockIdx.y * ntok + tok) * 2;
            o[0] = DECODE ? best : dsum;
            o[1] = DECODE ? __int_as_float(arg) : 0.f;
            continue;
        }
        if (DECODE) {
            a.categ_out[tok] = (int64_t)arg;
        } else {
            const float tot = fmaf(__builtin_amdgcn_exp2f(-lp2), dsum, 1.f);
            float cpl;
            if (density_sum_ok(tot, lp2)) cpl = -kLn2 * __builtin_amdgcn_logf(tot);
            else {
                // a copy goes to the callee (scratch memory): the latents themselves stay in registers for the sweep
                float zc[DT > 0 ? DT : kEncMaxD];
#pragma unroll
                for (int d = 0; d < (DT > 0 ? DT : kEncMaxD); ++d) zc[d] = z[d];
                cpl = (lp2 - lse2_from_raw_table(a.table, a.prior, zc, D, a.C, c, lp2, a.sigma, a.log_sigma)) * kLn2;
            }
            const float pv = a.pad ? a.pad[tok] : 1.f;
            if (a.cpl) a.cpl[tok] = cpl;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float o = z[d] * pv;
                bad |= isnan(o);
                a.z_out[tok * D + d] = o;
            }
            tok_ldj[tok] = (a.beta * cpl - (init_lp - ldj_f)) * pv;
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// ldj_out[b] = ldj_in[b] + sum_n tok_ldj[b, n]: one wave per row, lane-strided partial sums, fixed butterfly
__global__ __launch_bounds__(kBlock) void encoder_row_sum_kernel(const float* tok_ldj, const float* ldj_in, float* ldj_out,
                                                                 int B, int N, int* flags) {
    const int row = blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    if (row >= B) return;
    const int lane = threadIdx.x & 63;
    float acc = 0.f;
    for (int n = lane; n < N; n += kWave) acc += tok_ldj[(size_t)row * N + n];
    acc = wave_sum(acc);
    if (lane == 0) {
        const float v = (ldj_in ? ldj_in[row] : 0.f) + acc;
        ldj_out[row] = v;
        if (isnan(v)) raise_flag(flags, CNF_FLAG_NAN_LDJ);
    }
}


}  // namespace cnf

using namespace cnf;

static size_t table_bytes(int C, int D) { return (size_t)C * (6 * D + 3) * sizeof(float); }


// ---- round-3 pair kernels: eligibility ---------------------------------------------------------------------------------
static bool pair_has_d(int D) { return D == 1 || D == 2 || D == 3 || D == 4 || D == 6 || D == 8; }
static size_t pair_table_bytes(int C, int D, bool fwd) {
    const int S = (2 * D + 1 + 3) / 4 * 4;
    return (size_t)C * (S + (fwd ? 3 * D + 2 : D)) * sizeof(float);
}
static bool aligned_to(const void* p, size_t mask) { return ((uintptr_t)p & mask) == 0; }   // null passes
#define DISPATCH_PAIR_D(D, CALL)                          \
    switch (D) {                                          \
        case 1: { constexpr int DT = 1; CALL; } break;    \
        case 2: { constexpr int DT = 2; CALL; } break;    \
        case 3: { constexpr int DT = 3; CALL; } break;    \
        case 4: { constexpr int DT = 4; CALL; } break;    \
        case 6: { constexpr int DT = 6; CALL; } break;    \
        case 8: { constexpr int DT = 8; CALL; } break;    \
        default: break;                                   \
    }

extern "C" {

static std::atomic<int64_t> g_pair_launches{0};
void cnf_set_encoder_kernel(int which) { g_encoder_kernel = which; }
int64_t cnf_encoder_pair_launches(void) { return g_pair_launches; }

static int encoder_forward_impl(const int64_t* categ, const float* eps, const float* table,
                                const float* category_prior, const float* pad, float beta,
                                const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                int B, int N, int D, int C, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream, int eps_is_u, float u_squeeze, float* eps_out,
                                const float* e_bias = nullptr, const float* e_scales = nullptr, const float* e_w = nullptr,
                                const float* e_sldj = nullptr, const float* e_length = nullptr) {
    CNF_REQUIRE(categ && eps && table && category_prior && z_out && ldj_out, "cnf_encoder_forward: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0, "cnf_encoder_forward: bad shape");
    if (B == 0) return CNF_OK;
    CNF_REQUIRE(D <= kEncMaxD, "cnf_encoder_forward: D=%d exceeds %d", D, kEncMaxD);
    CNF_REQUIRE(N < 65536, "cnf_encoder_forward: N=%d exceeds 65535", N);
    const bool epi = e_w != nullptr;
    const size_t smem = (size_t)kWavesPerBlock * kMaxTileChunks * sizeof(float) + table_bytes(C, D) +
                        (epi ? (size_t)(2 * D + D * D + 1) * sizeof(float) : 0);
    if (smem > 64 * 1024) { set_error("cnf_encoder_forward: class table for C=%d D=%d exceeds LDS", C, D); return CNF_ERR_UNSUPPORTED; }
    if (epi && !(eps_is_u && e_bias && e_scales && e_sldj && (D <= 6 || D == 8))) {
        set_error("cnf_encoder_forward_actconv: the epilogue is built for the sampled forward and D in {1,2,3,4,5,6,8}");
        return CNF_ERR_UNSUPPORTED;
    }
    EncArgs a = {};
    a.categ = categ; a.eps = eps; a.table = table; a.prior = category_prior; a.pad = pad;
    a.ldj_in = ldj_in; a.z_out = z_out; a.ldj_out = ldj_out; a.cpl = class_prob_log; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.C = C; a.beta = beta; a.sigma = sigma; a.log_sigma = log_sigma;
    a.eps_is_u = eps_is_u; a.u_squeeze = u_squeeze; a.eps_out = eps_out;
    a.e_bias = e_bias; a.e_scales = e_scales; a.e_w = e_w; a.e_sldj = e_sldj; a.e_length = e_length;
    // Tokens per wave tile.  Interleaved runs on one MI355X after the density-sum loop (B=16384, N=64, D=6; us at tiles of
    // 64 / 128 / 256 tokens): 3 classes 14.4 / 12.9 / 14.1, 9 classes 19.2 / 18.3 / 19.2, 16 classes 25.1 / 25.1 / 26.4,
    // 32 classes 37.7 / 39.0 / 41.6, 51 classes 55.0 / 57.0 / 61.3 (rounds 1-3 used 256): one token per lane once the class
    // loop dominates (16 384 waves: two rounds of eight per SIMD overlap their load and store phases), two below that.
    // (The forced kernels of cnf_set_encoder_kernel(1 / 2) keep the 256-token tiles both were written for: the row sums'
    // order follows the tiling, and those two are compared bit for bit.)
    // A row length that fills a small tile badly (N = 38: one row = 38 of 64 lanes) moves on to the next larger tile.
    auto tiling_for = [&](int target) { return make_row_tiling(B, N, /*force_vec=*/1, target); };
    auto lane_use = [&](const RowTiling& t) {
        const long tok = (long)t.rw * N;
        return t.bpr ? 1.0 : (double)tok / (double)(((tok + kWave - 1) / kWave) * kWave);
    };
    RowTiling tl = tiling_for(256);
    if (g_encoder_kernel == 0) {
        const int first = C >= 24 ? 64 : 128;
        for (int target = first; target <= 256; target *= 2) {
            tl = tiling_for(target);
            if (lane_use(tl) >= 0.85) break;
        }
    }
    // two tokens per lane (round 3) when every pair of a tile starts at an even token and the views are aligned
    const size_t va = (D % 2 == 0) ? 15 : 7;
    const size_t smem_pair = (size_t)kWavesPerBlock * kMaxTileChunks * sizeof(float) + pair_table_bytes(C, D, true);
    // The pair kernel won from 24 classes on while every token streamed a log-sum-exp (72.0 -> 68.0 us at 51 classes);
    // with the density sum and the tiles above the one-token kernel is as fast or faster at every size measured (51 classes
    // 55.0 vs 55.0, 32: 37.7 vs 37.8, 16: 25.1 vs 26.3, 9: 18.3 vs 20.7; D = 8, 16 classes: 29.3 vs 29.3), so the pair kernel
    // is no longer selected automatically: it stays as the independent second implementation the bit-identity tests run
    // against (cnf_set_encoder_kernel(2)).
    const bool want_pair = g_encoder_kernel == 2 && !epi;
    const bool pair = want_pair && pair_has_d(D) && !tl.bpr && tl.rw >= 2 && ((long)tl.rw * N) % 2 == 0 &&
                      smem_pair <= 64 * 1024 && aligned_to(eps, va) && aligned_to(eps_out, va) && aligned_to(z_out, va) && aligned_to(categ, 15) &&
                      aligned_to(pad, 7) && aligned_to(class_prob_log, 7);
    if (pair) {
        ++g_pair_launches;
        DISPATCH_PAIR_D(D, CNF_LAUNCH((encoder_forward_pair_kernel<DT>), tiling_grid(tl), dim3(kBlock), smem_pair,
                                      (hipStream_t)stream, a, tl));
    } else {
        if (epi) {
            switch (D) {
                case 1: CNF_LAUNCH((encoder_forward_kernel<1, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                case 2: CNF_LAUNCH((encoder_forward_kernel<2, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                case 3: CNF_LAUNCH((encoder_forward_kernel<3, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                case 4: CNF_LAUNCH((encoder_forward_kernel<4, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                case 5: CNF_LAUNCH((encoder_forward_kernel<5, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                case 6: CNF_LAUNCH((encoder_forward_kernel<6, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
                default: CNF_LAUNCH((encoder_forward_kernel<8, true, true>), tiling_grid(tl), dim3(kBlock), smem, (hipStream_t)stream, a, tl); break;
            }
        } else if (eps_is_u) {
            DISPATCH_D(D, CNF_LAUNCH((encoder_forward_kernel<DT, true>), tiling_grid(tl), dim3(kBlock), smem,
                                     (hipStream_t)stream, a, tl));
        } else {
            DISPATCH_D(D, CNF_LAUNCH((encoder_forward_kernel<DT, false>), tiling_grid(tl), dim3(kBlock), smem,
                                     (hipStream_t)stream, a, tl));
        }
    }
    return launch_status("cnf_encoder_forward");
}

int cnf_encoder_forward(const int64_t* categ, const float* eps, const float* table,
                        const float* category_prior, const float* pad, float beta,
                        const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                        int B, int N, int D, int C, float sigma, float log_sigma,
                        int* flags, cnf_stream_t stream) {
    return encoder_forward_impl(categ, eps, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, class_prob_log,
                                B, N, D, C, sigma, log_sigma, flags, stream, 0, 0.f, nullptr);
}

int cnf_encoder_forward_sampled(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                const float* category_prior, const float* pad, float beta,
                                const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log, float* eps_out,
                                int B, int N, int D, int C, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream) {
    if (math_mode() != 1) {
        set_error("cnf_encoder_forward_sampled: the fused sampler is the fp32 one of math mode 1; run cnf_logistic_from_uniform + cnf_encoder_forward");
        return CNF_ERR_UNSUPPORTED;
    }
    return encoder_forward_impl(categ, u, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, class_prob_log,
                                B, N, D, C, sigma, log_sigma, flags, stream, 1, squeeze_eps, eps_out);
}

int cnf_encoder_forward_actconv(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                const float* category_prior, const float* pad, float beta,
                                const float* act_bias, const float* act_scales, const float* conv_weight, const float* conv_sldj,
                                const float* length,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                int B, int N, int D, int C, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(act_bias && act_scales && conv_weight && conv_sldj, "cnf_encoder_forward_actconv: null tensor");
    if (math_mode() != 1) {
        set_error("cnf_encoder_forward_actconv: the fused sampler is the fp32 one of math mode 1; run the layers separately");
        return CNF_ERR_UNSUPPORTED;
    }
    return encoder_forward_impl(categ, u, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, nullptr,
                                B, N, D, C, sigma, log_sigma, flags, stream, 1, squeeze_eps, nullptr,
                                act_bias, act_scales, conv_weight, conv_sldj, length);
}

int cnf_encoder_forward_actconv_cpl(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                    const float* category_prior, const float* pad, float beta,
                                    const float* act_bias, const float* act_scales, const float* conv_weight, const float* conv_sldj,
                                    const float* length,
                                    const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                    int B, int N, int D, int C, float sigma, float log_sigma,
                                    int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(act_bias && act_scales && conv_weight && conv_sldj, "cnf_encoder_forward_actconv_cpl: null tensor");
    if (math_mode() != 1) {
        set_error("cnf_encoder_forward_actconv_cpl: the fused sampler is the fp32 one of math mode 1; run the layers separately");
        return CNF_ERR_UNSUPPORTED;
    }
    return encoder_forward_impl(categ, u, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, class_prob_log,
                                B, N, D, C, sigma, log_sigma, flags, stream, 1, squeeze_eps, nullptr,
                                act_bias, act_scales, conv_weight, conv_sldj, length);
}

int cnf_encoder_decode(const float* z, const float* table, const float* category_prior,
                       int64_t* categ_out, int B, int N, int D, int C, float sigma, float log_sigma,
                       cnf_stream_t stream) {
    CNF_REQUIRE(z && table && category_prior && categ_out, "cnf_encoder_decode: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0, "cnf_encoder_decode: bad shape");
    if (B == 0) return CNF_OK;
    CNF_REQUIRE(D <= kEncMaxD, "cnf_encoder_decode: D=%d exceeds %d", D, kEncMaxD);
    const size_t smem = table_bytes(C, D);
    if (smem > 64 * 1024) { set_error("cnf_encoder_decode: class table for C=%d D=%d exceeds LDS", C, D); return CNF_ERR_UNSUPPORTED; }
    EncArgs a = {};
    a.z_in = z; a.table = table; a.prior = category_prior; a.categ_out = categ_out;
    a.B = B; a.N = N; a.D = D; a.C = C; a.sigma = sigma; a.log_sigma = log_sigma;
    const long ntok = (long)B * N;
    const size_t smem_pair = pair_table_bytes(C, D, false);
    const long pair_grid = ((ntok + 127) / 128 + kWavesPerBlock - 1) / kWavesPerBlock;   // one tile of 128 tokens per wave
    // the pair decode is within +-5 % of the round-2 kernel at every vocabulary size measured (20.3 vs 21.2 us at 16 classes,
    // 53.7 vs 55.1 at 51, 9.9 vs 9.7 at 3; profiles/r03_encoder_ab.txt): not selected automatically, kept for the A/B
    const bool pair = g_encoder_kernel == 2 && pair_has_d(D) && smem_pair <= 64 * 1024 && pair_grid < (1L << 31) &&
                      aligned_to(z, (D % 2 == 0) ? 15 : 7) && aligned_to(categ_out, 15);
    if (pair) {
        ++g_pair_launches;
        DISPATCH_PAIR_D(D, CNF_LAUNCH((encoder_decode_pair_kernel<DT>), dim3((unsigned)pair_grid), dim3(kBlock), smem_pair,
                                      (hipStream_t)stream, a, ntok));
    } else {
        // one trip per lane once the class loop dominates (grid caps of 2048 / 4096 / none at 10^6 tokens, D = 6: 51.6 / 50.6 /
        // 50.3 us at 51 classes, 33.9 / 33.1 / 33.1 at 32, but 13.1 / 13.7 / 13.6 at 9): the forward's rule
        const long cap = C >= 24 ? (1L << 22) : 256 * 8;
        const int grid = (int)std::min<long>((ntok + kBlock - 1) / kBlock, cap);
        DISPATCH_D(D, CNF_LAUNCH((encoder_decode_kernel<DT>), dim3(grid), dim3(kBlock), smem,
                                 (hipStream_t)stream, a, ntok));
    }
    return launch_status("cnf_encoder_decode");
}



int cnf_encoder_decode_actconv(const float* z, const float* act_bias, const float* act_scales, const float* conv_weight_inv,
                               const float* conv_sldj, const float* pad, const float* length,
                               const float* table, const float* category_prior,
                               const float* ldj_in, int64_t* categ_out, float* ldj_out,
                               int B, int N, int D, int C, float sigma, float log_sigma, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && act_bias && act_scales && conv_weight_inv && conv_sldj && table && category_prior && categ_out && ldj_out,
                "cnf_encoder_decode_actconv: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0, "cnf_encoder_decode_actconv: bad shape");
    if (B == 0) return CNF_OK;
    const size_t smem = table_bytes(C, D) + (size_t)(2 * D + D * D + 1) * sizeof(float);
    if (!(D <= 6 || D == 8) || smem > 64 * 1024) {
        set_error("cnf_encoder_decode_actconv: built for D in {1,2,3,4,5,6,8} and a class table that fits LDS (C=%d D=%d)", C, D);
        return CNF_ERR_UNSUPPORTED;
    }
    EncArgs a = {};
    a.z_in = z; a.table = table; a.prior = category_prior; a.categ_out = categ_out; a.pad = pad;
    a.ldj_in = ldj_in; a.ldj_out = ldj_out; a.flags = flags;
    a.e_bias = act_bias; a.e_scales = act_scales; a.e_w = conv_weight_inv; a.e_sldj = conv_sldj; a.e_length = length;
    a.B = B; a.N = N; a.D = D; a.C = C; a.sigma = sigma; a.log_sigma = log_sigma;
    const long ntok = (long)B * N;
    const dim3 grid((unsigned)std::min<long>(std::max<long>((ntok + kBlock - 1) / kBlock, 1), C >= 24 ? (1L << 22) : 256 * 8)), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    switch (D) {
        case 1: CNF_LAUNCH((encoder_decode_kernel<1, true>), grid, block, smem, st, a, ntok); break;
        case 2: CNF_LAUNCH((encoder_decode_kernel<2, true>), grid, block, smem, st, a, ntok); break;
        case 3: CNF_LAUNCH((encoder_decode_kernel<3, true>), grid, block, smem, st, a, ntok); break;
        case 4: CNF_LAUNCH((encoder_decode_kernel<4, true>), grid, block, smem, st, a, ntok); break;
        case 5: CNF_LAUNCH((encoder_decode_kernel<5, true>), grid, block, smem, st, a, ntok); break;
        case 6: CNF_LAUNCH((encoder_decode_kernel<6, true>), grid, block, smem, st, a, ntok); break;
        default: CNF_LAUNCH((encoder_decode_kernel<8, true>), grid, block, smem, st, a, ntok); break;
    }
    return launch_status("cnf_encoder_decode_actconv");
}

int64_t cnf_encoder_workspace_floats(int B, int N, int D, int C) {
    (void)D;
    const int ks = tiled_class_splits(C);
    return (int64_t)B * N * (1 + (ks > 1 ? 2 * ks : 0));
}

static int encoder_forward_tiled_impl(const int64_t* categ, const float* eps, const float* table,
                                      const float* category_prior, const float* pad, float beta,
                                      const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                      float* workspace,
                                      int B, int N, int D, int C, float sigma, float log_sigma,
                                      int* flags, cnf_stream_t stream, int eps_is_u, float u_squeeze, float* eps_out) {
    CNF_REQUIRE(categ && eps && table && category_prior && z_out && ldj_out && workspace, "cnf_encoder_forward_tiled: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0 && D <= kEncMaxD, "cnf_encoder_forward_tiled: bad shape");
    if (B == 0) return CNF_OK;
    EncArgs a = {};
    a.categ = categ; a.eps = eps; a.table = table; a.prior = category_prior; a.pad = pad;
    a.z_out = z_out; a.cpl = class_prob_log; a.flags = flags;
    a.B = B; a.N = N; a.D = D; a.C = C; a.beta = beta; a.sigma = sigma; a.log_sigma = log_sigma;
    a.eps_is_u = eps_is_u; a.u_squeeze = u_squeeze; a.eps_out = eps_out;
    const long ntok = (long)B * N;
    const int CC = std::min(C, tiled_chunk_classes(D));
    const size_t smem = (size_t)CC * (2 * D + 2) * sizeof(float);
    const int grid = (int)std::min<long>((ntok + kBlock - 1) / kBlock, 256 * 8);
    hipStream_t st = (hipStream_t)stream;
    const int KS = tiled_class_splits(C);
    float* part = workspace + ntok;
    if (KS == 1) {
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, false, 0>), dim3(grid), dim3(kBlock), smem, st, a, ntok, CC, workspace,
                                 (float*)nullptr, 1));
    } else {
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, false, 1>), dim3(grid, KS), dim3(kBlock), smem, st, a, ntok, CC, workspace,
                                 part, KS));
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, false, 2>), dim3(grid), dim3(kBlock), 0, st, a, ntok, CC, workspace,
                                 part, KS));
    }
    CNF_LAUNCH(encoder_row_sum_kernel, dim3((B + kWavesPerBlock - 1) / kWavesPerBlock), dim3(kBlock), 0, st,
               (const float*)workspace, ldj_in, ldj_out, B, N, flags);
    return launch_status("cnf_encoder_forward_tiled");
}

int cnf_encoder_forward_tiled(const int64_t* categ, const float* eps, const float* table,
                              const float* category_prior, const float* pad, float beta,
                              const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                              float* workspace,
                              int B, int N, int D, int C, float sigma, float log_sigma,
                              int* flags, cnf_stream_t stream) {
    return encoder_forward_tiled_impl(categ, eps, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, class_prob_log,
                                      workspace, B, N, D, C, sigma, log_sigma, flags, stream, 0, 0.f, nullptr);
}

int cnf_encoder_forward_tiled_sampled(const int64_t* categ, const float* u, float squeeze_eps, const float* table,
                                      const float* category_prior, const float* pad, float beta,
                                      const float* ldj_in, float* z_out, float* ldj_out, float* class_prob_log,
                                      float* eps_out, float* workspace,
                                      int B, int N, int D, int C, float sigma, float log_sigma,
                                      int* flags, cnf_stream_t stream) {
    if (math_mode() != 1) {
        set_error("cnf_encoder_forward_tiled_sampled: the fused sampler is the fp32 one of math mode 1; run cnf_logistic_from_uniform + cnf_encoder_forward_tiled");
        return CNF_ERR_UNSUPPORTED;
    }
    return encoder_forward_tiled_impl(categ, u, table, category_prior, pad, beta, ldj_in, z_out, ldj_out, class_prob_log,
                                      workspace, B, N, D, C, sigma, log_sigma, flags, stream, 1, squeeze_eps, eps_out);
}

int cnf_encoder_decode_tiled(const float* z, const float* table, const float* category_prior,
                             int64_t* categ_out, float* workspace, int B, int N, int D, int C, float sigma, float log_sigma,
                             cnf_stream_t stream) {
    CNF_REQUIRE(z && table && category_prior && categ_out, "cnf_encoder_decode_tiled: null tensor");
    CNF_REQUIRE(workspace || tiled_class_splits(C) == 1, "cnf_encoder_decode_tiled: this vocabulary needs the workspace");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0 && D <= kEncMaxD, "cnf_encoder_decode_tiled: bad shape");
    if (B == 0) return CNF_OK;
    EncArgs a = {};
    a.z_in = z; a.table = table; a.prior = category_prior; a.categ_out = categ_out;
    a.B = B; a.N = N; a.D = D; a.C = C; a.sigma = sigma; a.log_sigma = log_sigma;
    const long ntok = (long)B * N;
    const int CC = std::min(C, tiled_chunk_classes(D));
    const size_t smem = (size_t)CC * (2 * D + 2) * sizeof(float);
    const int grid = (int)std::min<long>((ntok + kBlock - 1) / kBlock, 256 * 8);
    hipStream_t st = (hipStream_t)stream;
    const int KS = tiled_class_splits(C);
    float* part = workspace ? workspace + ntok : nullptr;
    if (KS == 1) {
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, true, 0>), dim3(grid), dim3(kBlock), smem, st, a, ntok, CC,
                                 (float*)nullptr, (float*)nullptr, 1));
    } else {
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, true, 1>), dim3(grid, KS), dim3(kBlock), smem, st, a, ntok, CC,
                                 (float*)nullptr, part, KS));
        DISPATCH_D(D, CNF_LAUNCH((encoder_tiled_kernel<DT, true, 2>), dim3(grid), dim3(kBlock), 0, st, a, ntok, CC,
                                 (float*)nullptr, part, KS));
    }
    return launch_status("cnf_encoder_decode_tiled");
}

}  // extern "C"
