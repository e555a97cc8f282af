// Backward of the logistic-mixture CDF coupling FORWARD transform (mixture_cdf_layer.py:95-123,145-180), fp32 token-pass
// kernel: the training-side twin of mixture_tok_kernel (cnf_mixture_tok.hip).  The reference differentiates its eager op
// chain with autograd; here ONE kernel produces g_z, g_nn (through the tanh bounds and the channel mask) and the
// partial sums of g_scaling_factor [D] / g_mixture_scaling_factor [D, K].
//
// Data movement: the parameter rows of a pass are DMA-staged into LDS exactly like in the forward kernel; the
// per-parameter gradients then overwrite the staged row IN PLACE (slot k is read, then replaced by its own gradient),
// and the stage goes back to g_nn with coalesced stores (16 / 8 / 4 bytes per lane, the widest the spans' alignment
// allows).  The parameter blocks of channels that are not transformed get their zeros from the same kernel (no separate
// memset pass over g_nn: at K = 8 that tensor is 26x the latents).
//
// Arithmetic: fp32 with both tails as sums of positive terms — 1/u = se/cdf and 1/(1-u) = se/ccdf are exact to fp32
// rounding, (1 - 2 sigma) is formed as sigma(-z) - sigma(z) — and an fp64 branch, the arithmetic of the fp64 kernel
// (cnf_mixture_bwd.hip), for elements whose u, 1 - u or pdf sum underflows the fp32 forms.
//
// Parameter gradients: every lane always works on the same channel (and the same mixtures when several lanes share an
// item), so it keeps its partial sums in registers (compile-time K) or in lane-private LDS slots (run-time K) across
// all its passes; they are combined once per wave, then per workgroup, in a fixed order — no atomics, bit-reproducible —
// and one row per workgroup goes to the partials buffer that mix_reduce_partials_kernel sums in fp64.
#include "cnf_mixture_tok.h"

#include <algorithm>
#include <atomic>

namespace cnf {

constexpr int kTokBwdGrid = 1024;       // rows of the partials buffer (cnf_bwd_workspace_floats)
#ifndef CNF_MIXBWD_MERGED
#define CNF_MIXBWD_MERGED 1
#endif
#ifndef CNF_MIXBWD_NT_MB
#define CNF_MIXBWD_NT_MB 128
#endif
constexpr int kTokBwdNtMB = CNF_MIXBWD_NT_MB;      // MB of gradient rows above which the write-back is nontemporal
static std::atomic<int> g_tok_bwd_big_mb{kTokBwdNtMB};
static std::atomic<int> g_tok_bwd_w4{-1};
static inline int tok_bwd_w4() { return g_tok_bwd_w4.load(std::memory_order_relaxed); }

struct TokBwdArgs {
    const float* g_zout;      // [B,N,D] or null
    const float* g_ldj;       // [B] or null
    float* g_z;               // [B,N,D]
    float* g_nn;              // [B,N,D*P]
    float* partials;          // [gridDim.x, D + D*K]
    int wb_align;             // 16, 8 or 4: bytes per lane of the write-back; 24: 16-byte stores on the tokens' 16-byte grid although
                              // the span starts or ends on an odd multiple of 8 bytes (u_lo .. hi_half below)
    int u_lo, nu, ntu;        // mode 24: first 16-byte unit of a token the span touches, units it touches, units per token
    int lo_half, hi_half;     // mode 24: the span's first / last unit is half zeros (an untransformed block's bytes)
    int wb_nt;                // nontemporal write-back (large launches whose write-back instructions cover whole contiguous kilobytes)
    int merged;               // reference layout, modes 16 / 24: ONE loop writes the pass's tokens in address order, zero blocks included
    int src_stride;           // merged: bytes between the staged spans of consecutive tokens
    FastDiv div_ntu;          // merged: by ntu
    FastDiv div_nu, div_nz;   // mode 24: by nu; by ntu - nu
    int nacc;                 // run-time K: lane-private accumulator slots = 1 + ceil(K / G)
    int lacc_off;             // byte offset of the lane-private accumulators / the reduction scratch in dynamic LDS
    int wrow_off;             // byte offset of the per-wave parameter-gradient rows [4][PP]
    float* fix_partials;      // [gridDim.x, D + D*K] rows of the fix-up launch (read for flagged workgroups only)
    int* wave_flags;          // [gridDim.x * 4] 1 = the wave met a tail element
    int* block_flags;         // [gridDim.x]
    long nunits;              // wave work units: tiles (whole rows) or (row, wave-of-row) pairs
    FastDiv div_ncp, div_span;   // by the zero-fill units per token ((D - DA) * P * 4 / wb_align); by DA * P * 4 bytes (one span)
};

// tanh bound f tanh(raw / max(f,1)) and its derivatives w.r.t. raw and w.r.t. the log-factor sf (f = e^sf; the clamp
// passes the gradient for f >= 1), from the forward's table entry (x3 = 2 log2e / fc, m2f = -2 f, f)
__device__ __forceinline__ void bound_grads_f(float raw, const BoundTab& b, float& d_raw, float& d_sf) {
    const float inv_fc = b.x3 * 0.34657359027997264f;              // 1 / max(f, 1)
    const float th = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(raw * b.x3) + 1.f), -2.f, 1.f);
    const float sech2 = 1.f - th * th;
    const float uu = raw * inv_fc;
    d_raw = b.f * sech2 * inv_fc;
    d_sf = b.f >= 1.f ? b.f * (th - uu * sech2) : b.f * th;
}

// Stores of the gradient rows and zero blocks (g_nn is written once and not read again by this kernel).  `nt` = nontemporal: set by
// the host for launches whose every write-back instruction covers whole contiguous kilobytes (round 6, profiles/
// r06_mixture_bwd_floor.txt section 4: S* compact 239 -> 188 us; on write-backs whose cache lines are completed by a SECOND
// instruction — the reference layout's zero blocks written by a loop of their own — the same hint costs 10-30 %, which is why the
// reference layout writes its tokens in address order, zeros included, in one loop)
typedef float mb_f4 __attribute__((ext_vector_type(4)));
typedef float mb_f2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void wb_store(float4* p, float4 v, bool nt = false) {       // (nt: constant false outside the BIG builds)
    if (nt) __builtin_nontemporal_store(mb_f4{v.x, v.y, v.z, v.w}, reinterpret_cast<mb_f4*>(p));
    else *p = v;
}
__device__ __forceinline__ void wb_store(float2* p, float2 v, bool nt = false) {
    if (nt) __builtin_nontemporal_store(mb_f2{v.x, v.y}, reinterpret_cast<mb_f2*>(p));
    else *p = v;
}
__device__ __forceinline__ void wb_store(float* p, float v, bool nt = false) {
    if (nt) __builtin_nontemporal_store(v, p);
    else *p = v;
}

// g_z of an element the streaming kernel left to the fix-up launch: a quiet NaN with this payload
constexpr uint32_t kTailSentinel = 0x7fc0a11eu;

// One tail element (token `tok`, channel d) redone by a whole wave in fp64 — the arithmetic of the fp64 kernel
// (cnf_mixture_bwd.hip): lane k owns mixture k (k, k + 64, ... for K > 64), the four mixture sums are formed in mixture
// order, every lane writes its mixture's three gradients to g_nn, lane 0 the element's g_z and the t / log_s slots.
// Scaling-factor gradients are added to the wave's parameter row prow [D + D K] (one writer per word).
__device__ __forceinline__ void tail_fixup(const MixArgs& a, const TokBwdArgs& w, const BoundTab* mt, const BoundTab* sf_tab,
                                           size_t tok, int d, int K, int P, int lane, float* prow) {
    const size_t e = tok * a.D + d;
    const size_t pe = tok * a.nn_D + (d - a.nn_c0);          // the element's parameter block (reference / compact layout)
    const float* prm = a.nn + pe * (size_t)P;
    float* gprm = w.g_nn + pe * (size_t)P;
    const float pv = a.pad ? a.pad[tok] : 1.f;
    const float outscale = a.pad_output ? pv : 1.f;
    const double xd = (double)a.z[e];
    const double gzd = w.g_zout ? (double)(w.g_zout[e] * outscale) : 0.0;
    const double gld = w.g_ldj ? (double)w.g_ldj[tok / a.N] : 0.0;
    const float t = prm[0], raw_ls = prm[1];
    const float log_s = a.sf ? apply_bound(raw_ls, sf_tab[d]) : raw_ls;
    float mx = -INFINITY;
    for (int k = lane; k < K; k += kWave) mx = fmaxf(mx, prm[2 + k]);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m, kWave));
    // the four mixture sums in the fp64 kernel's own order (k = 0, 1, 2, ...): beyond |logit| ~ 36 the value of 1 - u is
    // rounding noise of these sums, and only the same order reproduces the fp64 kernel there
    double sed = 0.0, cdfd = 0.0, pdfd = 0.0, dpdfd = 0.0;
    for (int k0 = 0; k0 < K; k0 += kWave) {
        const int k = k0 + lane;
        double wd = 0.0, t_cdf = 0.0, t_pdf = 0.0, t_dpdf = 0.0;
        if (k < K) {
            const float lsf = a.msf ? apply_bound(prm[2 + 2 * K + k], mt[k]) : prm[2 + 2 * K + k];
            wd = exp((double)prm[2 + k] - (double)mx);
            const double isd = exp(-(double)lsf);
            const double zd = (xd - (double)prm[2 + K + k]) * isd;
            const double ed = exp(-fabs(zd));
            const double rd = 1.0 / (1.0 + ed);
            const double sg = zd >= 0.0 ? rd : ed * rd;
            const double pk = ed * rd * rd * isd;
            t_cdf = wd * sg;
            t_pdf = wd * pk;
            t_dpdf = wd * pk * (1.0 - 2.0 * sg) * isd;
        }
        const int nk = min(kWave, K - k0);
        for (int l = 0; l < nk; ++l) {
            sed += __shfl(wd, l, kWave);
            cdfd += __shfl(t_cdf, l, kWave);
            pdfd += __shfl(t_pdf, l, kWave);
            dpdfd += __shfl(t_dpdf, l, kWave);
        }
    }
    const double ud = cdfd / sed, pdf_n = pdfd / sed;
    const double a_s = exp((double)log_s);
    // torch.clamp keeps a NaN (fmax returns its other operand): a diverged element gives NaN gradients, as autograd does
    const double ucl = ud != ud ? ud : fmax(ud, 1e-22), u1cl = ud != ud ? ud : fmax(1.0 - ud, 1e-22);
    const double lud = log(ucl), l1ud = log(u1cl);
    const double dlu = ud > 1e-22 ? 1.0 / ud : (ud != ud ? ud : 0.0);
    const double dl1u = (1.0 - ud) > 1e-22 ? -1.0 / (1.0 - ud) : (ud != ud ? ud : 0.0);
    const double zt = ((lud - l1ud) + (double)t) * a_s;
    double g_ud = gzd * a_s * (dlu - dl1u) + gld * (-dlu - dl1u);
    if (a.use_reg) {
        double dreg = 0.0;
        if (lud / kLn10 <= -a.reg_max) dreg += dlu / kLn10;
        if (l1ud / kLn10 <= -a.reg_max) dreg += dl1u / kLn10;
        g_ud += gld * a.reg_factor * dreg;
    }
    const double inv_pdf = pdf_n > 1e-290 ? 1.0 / pdf_n : 0.0;
    const float g_logs = (float)(gzd * zt + gld);
    float d_raw0 = 1.f, d_sf0 = 0.f;
    if (a.sf) bound_grads_f(raw_ls, sf_tab[d], d_raw0, d_sf0);
    if (lane == 0) {
        w.g_z[e] = (float)(g_ud * pdf_n + gld * (dpdfd / sed) * inv_pdf);
        gprm[0] = (float)(gzd * a_s);
        gprm[1] = g_logs * d_raw0;
        prow[d] += g_logs * d_sf0;
    }
    for (int k = lane; k < K; k += kWave) {
        const float raw = prm[2 + 2 * K + k];
        float lsf = raw, d_raw = 1.f, d_sf = 0.f;
        if (a.msf) {
            lsf = apply_bound(raw, mt[k]);
            bound_grads_f(raw, mt[k], d_raw, d_sf);
        }
        const double pi = exp((double)prm[2 + k] - (double)mx) / sed;
        const double isd = exp(-(double)lsf);
        const double zd = (xd - (double)prm[2 + K + k]) * isd;
        const double ed = exp(-fabs(zd));
        const double rd = 1.0 / (1.0 + ed);
        const double sg = zd >= 0.0 ? rd : ed * rd;
        const double s1s = ed * rd * rd;
        const double pk = s1s * isd;
        const double resp = pi * pk * inv_pdf;
        const double g_lp = g_ud * pi * (sg - ud) + gld * (resp - pi);
        const double g_mu = g_ud * (-pi * pk) + gld * (-resp * (1.0 - 2.0 * sg) * isd);
        const double g_ls = g_ud * (-pi * zd * s1s) + gld * (resp * (-1.0 - zd * (1.0 - 2.0 * sg)));
        gprm[2 + k] = (float)g_lp;
        gprm[2 + K + k] = (float)g_mu;
        gprm[2 + 2 * K + k] = (float)(g_ls * (double)d_raw);
        prow[a.D + d * K + k] += (float)g_ls * d_sf;
    }
}

// The fix-up launch: same grid and the same wave -> unit mapping as the streaming kernel.  A workgroup none of whose waves
// met a tail returns at once (one load); a flagged wave walks its units again, finds its tail elements by their sentinel
// g_z and redoes them one by one with tail_fixup.  Its parameter-gradient sums go to the workgroup's row of a second
// partials region that the reduction adds for flagged workgroups only.
__global__ __launch_bounds__(kBlock) void mixture_tok_bwd_fixup_kernel(MixArgs a, TokGeom gm, TokBwdArgs w) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if (w.block_flags[blockIdx.x] == 0) return;
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int K = a.K, P = a.P, PP = a.D + a.D * K;
    BoundTab* sf_tab = reinterpret_cast<BoundTab*>(smem);
    BoundTab* msf_tab = sf_tab + a.D;
    float* rows = reinterpret_cast<float*>(msf_tab + a.D * K);
    float* prow = rows + (size_t)wave * PP;
    for (int i = threadIdx.x; i < a.D; i += blockDim.x) sf_tab[i] = make_bound(a.sf ? a.sf[i] : 0.f);
    for (int i = threadIdx.x; i < a.D * K; i += blockDim.x) msf_tab[i] = make_bound(a.msf ? a.msf[i] : 0.f);
    for (int i = lane; i < PP; i += kWave) prow[i] = 0.f;
    __syncthreads();
    if (w.wave_flags[blockIdx.x * kWavesPerBlock + wave]) {
        for (long unit = (long)blockIdx.x * kWavesPerBlock + wave; unit < w.nunits; unit += (long)gridDim.x * kWavesPerBlock) {
            int row0, n_first, ntok;
            if (!gm.split) {
                row0 = (int)(unit * gm.rw);
                n_first = 0;
                ntok = min(gm.rw, a.B - row0) * a.N;
            } else {
                const int nw = gm.S * kWavesPerBlock;
                row0 = (int)(unit / nw);
                const int wv = (int)(unit - (long)row0 * nw);
                const int p_lo = (wv * gm.ppr) / nw, p_hi = ((wv + 1) * gm.ppr) / nw;
                n_first = p_lo * gm.TPP;
                ntok = max(0, min(a.N, p_hi * gm.TPP) - n_first);
            }
            const size_t tok_g0 = (size_t)row0 * a.N + n_first;
            const int items = ntok * gm.DA;
            for (int it0 = 0; it0 < items; it0 += kWave) {
                const int it = it0 + lane;
                const bool valid = it < items;
                const int tokl = valid ? it / gm.DA : 0;
                const int d = gm.d0 + (valid ? it - tokl * gm.DA : 0);
                const bool is_tail = valid && __float_as_uint(w.g_z[(tok_g0 + tokl) * a.D + d]) == kTailSentinel;
                unsigned long long todo = __ballot(is_tail);
                while (todo) {
                    const int src = __builtin_ctzll(todo);
                    todo &= todo - 1;
                    const int s_tokl = __builtin_amdgcn_readlane(tokl, src), s_d = __builtin_amdgcn_readlane(d, src);
                    tail_fixup(a, w, msf_tab + s_d * K, sf_tab, tok_g0 + s_tokl, s_d, K, P, lane, prow);
                    wave_lds_sync();
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < PP; i += blockDim.x) {
        float t = 0.f;
        for (int wv = 0; wv < kWavesPerBlock; ++wv) t += rows[wv * PP + i];
        w.fix_partials[(size_t)blockIdx.x * PP + i] = t;
    }
}

// partials [nrows, P] and, for workgroups whose flag is set, fix_partials [nrows, P] -> column sums in fp64, fixed order
__global__ __launch_bounds__(kBlock) void mix_reduce_partials_fix_kernel(const float* partials, const float* fix_partials,
                                                                         const int* block_flags, int nrows, int P,
                                                                         float* out_a, float* out_b, int split) {
    const int p = blockIdx.x;
    double accd = 0.0;
    for (int r = threadIdx.x; r < nrows; r += kBlock) {
        accd += (double)partials[(size_t)r * P + p];
        if (block_flags[r]) accd += (double)fix_partials[(size_t)r * P + p];
    }
    __shared__ double sh[kWavesPerBlock];
    accd = wave_sum(accd);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = accd;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w2 = 0; w2 < kWavesPerBlock; ++w2) t += sh[w2];
        if (p < split) {
            if (out_a) out_a[p] = (float)t;
        } else if (out_b) {
            out_b[p - split] = (float)t;
        }
    }
}

template <int KT, int G, bool BIG>
__device__ __forceinline__ void mixture_tok_bwd_body(const MixArgs& a, const TokGeom& gm, const TokBwdArgs& w) {
    static_assert(G == 1 || KT == 0, "several lanes per item only with a run-time K");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int K = KT > 0 ? KT : a.K;
    const int P = a.P;
    const int PP = a.D + a.D * K;
    // one stage per wave.  (A second stage with the next pass's DMA issued behind this pass's arithmetic was built in round 4 —
    // bit-identical, S* 322 -> 326 us, Zinc nodes 26.6 -> 31.6: profiles/r04_sweep_mixture_bwd.txt — and removed in round 6.)
    char* const stage_b = smem + (size_t)wave * gm.stage_bytes;
    BoundTab* sf_tab = reinterpret_cast<BoundTab*>(smem + (size_t)kWavesPerBlock * gm.stage_bytes);
    BoundTab* msf_tab = sf_tab + a.D;
    // lane-private partial sums [64][nacc] per wave: a region of their own for a run-time K (they live there during the
    // passes); with a compile-time K they live in registers and only visit LDS for the final reduction, in the stage
    float* lacc = KT > 0 ? reinterpret_cast<float*>(stage_b)
                         : reinterpret_cast<float*>(smem + w.lacc_off) + (size_t)wave * kWave * w.nacc;
    float* wrow = reinterpret_cast<float*>(smem + w.wrow_off) + (size_t)wave * PP;             // [PP] per wave
    for (int i = threadIdx.x; i < a.D; i += blockDim.x)
        sf_tab[i] = make_bound(a.sf ? a.sf[i] : 0.f);
    for (int i = threadIdx.x; i < a.D * K; i += blockDim.x)
        msf_tab[i] = make_bound(a.msf ? a.msf[i] : 0.f);
    if (KT == 0)
        for (int i = lane; i < kWave * w.nacc; i += kWave) lacc[i] = 0.f;
    for (int i = lane; i < PP; i += kWave) wrow[i] = 0.f;
    __syncthreads();

    const int tli = (int)fdiv((uint32_t)lane, gm.div_lpt);
    const int rem = lane - tli * gm.lpt;
    const int j = rem / G, sub = rem - j * G;
    const int d = gm.d0 + j;
    const BoundTab* mt = msf_tab + d * K;
    const char* nn_lo = reinterpret_cast<const char*>(a.nn);
    const char* nn_last = nn_lo + ((size_t)a.B * a.N * a.nn_D * P - 4) * sizeof(float);
    constexpr int KK = KT > 0 ? KT : 1;
    float acc_sf = 0.f, acc_m[KK];
#pragma unroll
    for (int k = 0; k < KK; ++k) acc_m[k] = 0.f;
    float* my_acc = lacc + lane * w.nacc;       // run-time K: slot 0 = scaling_factor, slot 1 + i = mixture k = sub + i G
    bool any_tail = false;

    for (long unit = (long)blockIdx.x * kWavesPerBlock + wave; unit < w.nunits; unit += (long)gridDim.x * kWavesPerBlock) {
        int row0, n_first, ntok;
        if (!gm.split) {
            row0 = (int)(unit * gm.rw);
            n_first = 0;
            ntok = min(gm.rw, a.B - row0) * a.N;
        } else {
            const int nw = gm.S * kWavesPerBlock;
            row0 = (int)(unit / nw);
            const int wv = (int)(unit - (long)row0 * nw);
            const int p_lo = (wv * gm.ppr) / nw, p_hi = ((wv + 1) * gm.ppr) / nw;
            n_first = p_lo * gm.TPP;
            ntok = max(0, min(a.N, p_hi * gm.TPP) - n_first);
        }
        const size_t tok_g0 = (size_t)row0 * a.N + n_first;
        const float* z_tile = a.z + tok_g0 * a.D;
        const float* gzo_tile = w.g_zout ? w.g_zout + tok_g0 * a.D : nullptr;
        float* gz_tile = w.g_z + tok_g0 * a.D;
        const float* pad_tile = a.pad ? a.pad + tok_g0 : nullptr;
        const char* span0 = nn_lo + (tok_g0 * a.nn_D + (gm.d0 - a.nn_c0)) * (size_t)P * sizeof(float);
        float* gnn_tile = w.g_nn + tok_g0 * a.nn_D * (size_t)P;          // first token's first block

        for (int tp = 0; tp < ntok; tp += gm.TPP) {
            const int npt = min(gm.TPP, ntok - tp);
            const bool valid = tli < npt;
            const int tokl = tp + (valid ? tli : 0);
            int rl = 0, n = n_first + tokl;
            if (!gm.split) {
                rl = (int)fdiv((uint32_t)tokl, gm.div_n);
                n = tokl - rl * a.N;
            }
            const float pv = pad_tile ? pad_tile[tokl] : 1.f;
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 4
            const float x = 0.f;             // A/B build: none of the small loads either (the DMA stream alone)
#else
            const float x = valid ? z_tile[(size_t)tokl * a.D + d] : 0.f;
#endif
            const float outscale = a.pad_output ? pv : 1.f;
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 4
            const float gzo = outscale, gl = 0.f;
#else
            const float gzo = (valid && gzo_tile) ? gzo_tile[(size_t)tokl * a.D + d] * outscale : 0.f;
            const float gl = (valid && w.g_ldj) ? w.g_ldj[row0 + rl] : 0.f;
#endif
            bool active = valid;
            if (a.per_item_mask) active = active && mask_at(a.mask, a.mr, a.mc, n, d) == 0.f;
            if (a.pad_in_transform && pv == 0.f) active = false;
            // the upstream gradients of the channels that are copied through (this lane's element `lane` of the pass's
            // npt * ncopy), loaded HERE with the pass's other inputs: loaded where they are stored — behind the write-back — their
            // wait drained the write-back's stores as well (one more serial round trip per pass, under the write stream's
            // back-pressure: profiles/r06_mixture_bwd_floor.txt)
            constexpr int CTB = 1;
            float ctg[CTB];
#pragma unroll
            for (int i = 0; i < CTB; ++i) {
                ctg[i] = 0.f;
                const int e = lane + i * kWave;
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 4
                if (false)
#endif
                if (gm.ncopy > 0 && gzo_tile && e < npt * gm.ncopy) {
                    const int tk = (int)fdiv((uint32_t)e, gm.div_nc);
                    const int jj = e - tk * gm.ncopy;
                    const int c = jj < gm.d0 ? jj : jj + gm.DA;
                    const int tl2 = tp + tk;
                    const float pv2 = (pad_tile && a.pad_output) ? pad_tile[tl2] : 1.f;
                    ctg[i] = gzo_tile[(size_t)tl2 * a.D + c] * pv2;
                }
            }

            const char* pass_addr = span0 + (size_t)tp * gm.tokstride;
            int my_pos = stage_pass(gm, stage_b, pass_addr, nn_last, npt, lane, tli, j, P);
            if (!valid) my_pos = 0;
            float* my = reinterpret_cast<float*>(stage_b + my_pos);
            wave_lds_sync();

            float g_x = gzo;                       // an element that is not transformed passes its gradient through
#ifdef CNF_MIXBWD_ABLATE
            // A/B build: the data movement alone (rows staged, the latents' gradient passed through, rows and zero blocks written
            // back) with the arithmetic compiled out: the floor of this kernel's access pattern (profiles/r05_mixture_bwd_floor.txt)
            if (false) {
#else
            if (active) {
#endif
                const float t = my[0];
                const float raw_ls = my[1];
                const float log_s = a.sf ? apply_bound(raw_ls, sf_tab[d]) : raw_ls;
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
                // pass 1: the mixture sums
                float se = 0.f, cdf = 0.f, ccdf = 0.f, pdf = 0.f, dpdf = 0.f;
                auto sums = [&](float lpk, float muk, float lsk, int k) {
                    const float ls = a.msf ? apply_bound(lsk, mt[k]) : lsk;
                    const float inv_s = __builtin_amdgcn_exp2f(-ls * kLog2eF);
                    const float wk = __builtin_amdgcn_exp2f((lpk - mx) * kLog2eF);
                    const float zk = (x - muk) * inv_s;
                    const float e = __builtin_amdgcn_exp2f(-fabsf(zk) * kLog2eF);
                    const float rr = __builtin_amdgcn_rcpf(1.f + e);
                    const float er = e * rr;
                    const bool pos = zk >= 0.f;
                    const float sig = pos ? rr : er, sigc = pos ? er : rr;
                    const float wpk = wk * inv_s * (er * rr);
                    se += wk;
                    cdf = fmaf(wk, sig, cdf);
                    ccdf = fmaf(wk, sigc, ccdf);
                    pdf += wpk;
                    dpdf = fmaf(wpk * inv_s, sigc - sig, dpdf);
                };
                if (KT > 0) {
#pragma unroll
                    for (int k = 0; k < KK; ++k) sums(lp[k], mu[k], lsr[k], k);
                } else {
                    for (int k = sub; k < K; k += G) sums(my[2 + k], my[2 + K + k], my[2 + 2 * K + k], k);
                    se = gsum<G>(se); cdf = gsum<G>(cdf); ccdf = gsum<G>(ccdf); pdf = gsum<G>(pdf); dpdf = gsum<G>(dpdf);
                }
                const float inv_se = __builtin_amdgcn_rcpf(se);
                const float u = cdf * inv_se, uc = ccdf * inv_se;
                float g_t, g_logs;
                if (u > 1e-9f && uc > 1e-9f && pdf > 1e-30f) {
                    const float a_s = __builtin_amdgcn_exp2f(log_s * kLog2eF);
                    const float dlu = se * __builtin_amdgcn_rcpf(cdf);             // 1 / u
                    const float dl1u = -se * __builtin_amdgcn_rcpf(ccdf);          // -1 / (1 - u)
                    const float l2se = __builtin_amdgcn_logf(se);
                    const float lu = (__builtin_amdgcn_logf(cdf) - l2se) * kLn2F;
                    const float l1u = (__builtin_amdgcn_logf(ccdf) - l2se) * kLn2F;
                    const float zt = ((lu - l1u) + t) * a_s;
                    float g_u = gzo * a_s * (dlu - dl1u) + gl * (-dlu - dl1u);
                    if (a.use_reg) {
                        const float rmax = (float)a.reg_max;
                        float dreg = 0.f;
                        if (lu * 0.43429448190325176f <= -rmax) dreg += dlu * 0.43429448190325176f;
                        if (l1u * 0.43429448190325176f <= -rmax) dreg += dl1u * 0.43429448190325176f;
                        g_u = fmaf(gl * (float)a.reg_factor, dreg, g_u);
                    }
                    const float inv_pdf_se = __builtin_amdgcn_rcpf(pdf);          // 1 / (se pdf_n)
                    g_x = g_u * pdf * inv_se + gl * dpdf * inv_pdf_se;
                    g_logs = fmaf(gzo, zt, gl);
                    g_t = gzo * a_s;
                    // pass 2: per-mixture gradients, written over the staged parameters
                    auto grads = [&](float lpk, float muk, float lsk, int k, float& acc_slot) {
                        float ls = lsk, d_raw = 1.f, d_sf = 0.f;
                        if (a.msf) {
                            ls = apply_bound(lsk, mt[k]);
                            bound_grads_f(lsk, mt[k], d_raw, d_sf);
                        }
                        const float inv_s = __builtin_amdgcn_exp2f(-ls * kLog2eF);
                        const float wk = __builtin_amdgcn_exp2f((lpk - mx) * kLog2eF);
                        const float zk = (x - muk) * inv_s;
                        const float e = __builtin_amdgcn_exp2f(-fabsf(zk) * kLog2eF);
                        const float rr = __builtin_amdgcn_rcpf(1.f + e);
                        const float er = e * rr;
                        const bool pos = zk >= 0.f;
                        const float sig = pos ? rr : er, sigc = pos ? er : rr;
                        const float s1s = er * rr;
                        const float pk = s1s * inv_s;
                        const float pi = wk * inv_se;
                        const float resp = wk * pk * inv_pdf_se;                   // pi_k p_k / pdf
                        const float dsg = sigc - sig;                              // 1 - 2 sigma
                        // sigma_k - u without cancellation: near u = 1 both are 1 - O(1e-7), so use the complements there
                        const float dsu = u <= 0.5f ? sig - u : uc - sigc;
                        const float g_lp = g_u * pi * dsu + gl * (resp - pi);
                        const float g_mu = -g_u * pi * pk - gl * resp * dsg * inv_s;
                        const float g_ls = -g_u * pi * zk * s1s + gl * resp * (-1.f - zk * dsg);
                        my[2 + k] = g_lp;
                        my[2 + K + k] = g_mu;
                        my[2 + 2 * K + k] = g_ls * d_raw;
                        acc_slot = fmaf(g_ls, d_sf, acc_slot);
                    };
                    if (KT > 0) {
#pragma unroll
                        for (int k = 0; k < KK; ++k) grads(lp[k], mu[k], lsr[k], k, acc_m[k]);
                    } else {
                        int slot = 1;
                        for (int k = sub; k < K; k += G, ++slot) grads(my[2 + k], my[2 + K + k], my[2 + 2 * K + k], k, my_acc[slot]);
                    }
                } else {
                    // rare: a tail or an underflow.  The element is redone in fp64 by the whole wave at the END of the pass
                    // (tail_fixup), where none of the unrolled mixture state is live: inside this branch the fp64 code cost
                    // the streaming path 21-36 VGPRs.  Zeros go to the stage meanwhile.
                    any_tail = true;
                    g_x = __uint_as_float(kTailSentinel);
                    g_t = 0.f;
                    g_logs = 0.f;
                    for (int k = sub; k < K; k += G) {
                        my[2 + k] = 0.f;
                        my[2 + K + k] = 0.f;
                        my[2 + 2 * K + k] = 0.f;
                    }
                }
                // t and log_s (the shared first two slots: one lane of the item writes)
                float d_raw = 1.f, d_sf = 0.f;
                if (a.sf) bound_grads_f(raw_ls, sf_tab[d], d_raw, d_sf);
                if (sub == 0) {
                    my[0] = g_t;
                    my[1] = g_logs * d_raw;
                    if (KT > 0) acc_sf = fmaf(g_logs, d_sf, acc_sf);
                    else my_acc[0] = fmaf(g_logs, d_sf, my_acc[0]);
                }
            } else if (valid) {
                // padded token / masked item: its parameters get no gradient
                for (int i = sub; i < P; i += G) my[i] = 0.f;
            }
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 3
            if (g_x == 12345.678f)           // A/B build: no stores of g_z either
#endif
            if (valid && sub == 0) gz_tile[(size_t)tokl * a.D + d] = g_x;
            wave_lds_sync();

            // ---- the staged gradient rows go back to g_nn: the transformed spans of the pass's tokens
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 2
            if (g_x == 12345.678f)           // A/B build: no write-back of g_nn either
#endif
            {
                const int span_b = gm.DA * P * 4;
                const int total_b = npt * span_b;
                float* gspan0 = gnn_tile + ((size_t)tp * a.nn_D + (gm.d0 - a.nn_c0)) * P;       // first token's span in g_nn
                const uintptr_t src0 = reinterpret_cast<uintptr_t>(span0 + (size_t)tp * gm.tokstride);
                // BIG (large launches, one lane per item): nontemporal stores, the reference layout in address order — a build of its
                // own, so that the ordinary write-back's loops stay as they were (a run-time flag in them cost configs[1] 11 %)
                const bool nt = BIG && w.wb_nt != 0;
                if (w.wb_align == 32) {
                    // the pass is ONE contiguous span (compact layout, or no mask) whose token size is not a multiple of 16 bytes
                    // (S* compact: 312-byte tokens; the language model's 1860): 16-byte stores on the span's own 16-byte grid — nn and
                    // g_nn share the phase, the stage was filled at it — with up to three 4-byte stores at either end
                    const int ph = (int)(src0 & 15);
                    const int head = (16 - ph) & 15;
                    const int hb = min(head, total_b);
                    char* gdst = reinterpret_cast<char*>(gspan0);
                    if (lane * 4 < hb) wb_store(reinterpret_cast<float*>(gdst + lane * 4), *reinterpret_cast<const float*>(stage_b + ph + lane * 4));
                    const int body = (total_b - hb) & ~15;
                    for (int b = hb + lane * 16; b < hb + body; b += kWave * 16)
                        wb_store(reinterpret_cast<float4*>(gdst + b), *reinterpret_cast<const float4*>(stage_b + ph + b), nt);
                    const int tb = hb + body + lane * 4;
                    if (tb < total_b) wb_store(reinterpret_cast<float*>(gdst + tb), *reinterpret_cast<const float*>(stage_b + ph + tb));
                } else if (BIG && w.merged) {
                    // reference layout, everything on the tokens' 16-byte grid: the pass's tokens are written in address order — a
                    // unit inside the transformed span comes from the stage (a unit it shares with an untransformed block takes that
                    // block's zeros along), every other unit is zeros — so that each store instruction covers 1 KiB of whole lines
                    const int total_u = npt * w.ntu;
                    char* gtok = reinterpret_cast<char*>(gnn_tile + (size_t)tp * a.nn_D * P);
                    for (int e = lane; e < total_u; e += kWave) {
                        const int s = (int)fdiv((uint32_t)e, w.div_ntu);
                        const int ui = e - s * w.ntu;
                        const int k = ui - w.u_lo;
                        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
                        if (k >= 0 && k < w.nu) {
                            v = *reinterpret_cast<const float4*>(stage_b + s * w.src_stride + 16 * k);
                            if (k == 0 && w.lo_half) v.x = v.y = 0.f;
                            if (k == w.nu - 1 && w.hi_half) v.z = v.w = 0.f;
                        }
                        wb_store(reinterpret_cast<float4*>(gtok + (size_t)s * gm.tokstride + 16 * ui), v, nt);
                    }
                } else if (w.wb_align == 24) {
                    // S* (D = 6: the span is bytes 312..623 of a 624-byte token): 8-byte stores were twice the instructions at half
                    // the width.  A token's base is 16-byte aligned, so the span is written as the 16-byte units it touches — the unit
                    // it shares with an untransformed block takes that block's zeros along — and the zero loop writes the other units.
                    const int total_u = npt * w.nu;
                    char* gtok = reinterpret_cast<char*>(gnn_tile + (size_t)tp * a.nn_D * P);
                    for (int e = lane; e < total_u; e += kWave) {
                        const int s = (int)fdiv((uint32_t)e, w.div_nu);
                        const int ui = e - s * w.nu;
                        float4 v = *reinterpret_cast<const float4*>(stage_b + s * gm.slot + 16 * ui);
                        if (ui == 0 && w.lo_half) v.x = v.y = 0.f;
                        if (ui == w.nu - 1 && w.hi_half) v.z = v.w = 0.f;
                        wb_store(reinterpret_cast<float4*>(gtok + (size_t)s * gm.tokstride + 16 * (w.u_lo + ui)), v);
                    }
                } else if (w.wb_align == 16) {
                    for (int b = lane * 16; b < total_b; b += kWave * 16) {
                        const int s = (int)fdiv((uint32_t)b, w.div_span);
                        const int r = b - s * span_b;
                        const int lpos = gm.contig ? (int)(src0 & 15) + s * gm.tokstride + r
                                                   : s * gm.slot + (int)((src0 + (uintptr_t)s * gm.tokstride) & 15) + r;
                        const float4 v = *reinterpret_cast<const float4*>(stage_b + lpos);
                        wb_store(reinterpret_cast<float4*>(reinterpret_cast<char*>(gspan0) + (size_t)s * gm.tokstride + r), v, nt);
                    }
                } else if (w.wb_align == 8) {
                    for (int b = lane * 8; b < total_b; b += kWave * 8) {
                        const int s = (int)fdiv((uint32_t)b, w.div_span);
                        const int r = b - s * span_b;
                        const int lpos = gm.contig ? (int)(src0 & 15) + s * gm.tokstride + r
                                                   : s * gm.slot + (int)((src0 + (uintptr_t)s * gm.tokstride) & 15) + r;
                        const float2 v = *reinterpret_cast<const float2*>(stage_b + lpos);
                        wb_store(reinterpret_cast<float2*>(reinterpret_cast<char*>(gspan0) + (size_t)s * gm.tokstride + r), v);
                    }
                } else {
                    for (int b = lane * 4; b < total_b; b += kWave * 4) {
                        const int s = (int)fdiv((uint32_t)b, w.div_span);
                        const int r = b - s * span_b;
                        const int lpos = gm.contig ? (int)(src0 & 15) + s * gm.tokstride + r
                                                   : s * gm.slot + (int)((src0 + (uintptr_t)s * gm.tokstride) & 15) + r;
                        const float v = *reinterpret_cast<const float*>(stage_b + lpos);
                        wb_store(reinterpret_cast<float*>(reinterpret_cast<char*>(gspan0) + (size_t)s * gm.tokstride + r), v);
                    }
                }
            }
            // ---- channels that are not transformed: their latents pass the gradient through, their parameter blocks get zeros
            // (the compact layout has no such blocks)
#if defined(CNF_MIXBWD_ABLATE) && CNF_MIXBWD_ABLATE + 0 >= 3
            if (g_x == 12345.678f)
#endif
            if (gm.ncopy > 0) {
                const int ne = npt * gm.ncopy;
                int it = 0;
                for (int e = lane; e < ne; e += kWave, ++it) {
                    const int tk = (int)fdiv((uint32_t)e, gm.div_nc);
                    const int jj = e - tk * gm.ncopy;
                    const int c = jj < gm.d0 ? jj : jj + gm.DA;
                    const int tl2 = tp + tk;
                    float v;
                    if (it < CTB) v = ctg[it < CTB ? it : 0];
                    else {
                        const float pv2 = (pad_tile && a.pad_output) ? pad_tile[tl2] : 1.f;
                        v = gzo_tile ? gzo_tile[(size_t)tl2 * a.D + c] * pv2 : 0.f;
                    }
                    gz_tile[(size_t)tl2 * a.D + c] = v;
                }
            }
            if (gm.ncopy > 0 && a.nn_D == a.D && !(BIG && w.merged)) {
                // zeros for the parameter blocks of the untransformed channels, as wide as the write-back
                const int ncp_b = gm.ncopy * P * 4;            // bytes of untransformed parameter blocks per token
                const int head_b = gm.d0 * P * 4, span_b2 = gm.DA * P * 4;
                char* gtok0 = reinterpret_cast<char*>(gnn_tile + (size_t)tp * a.nn_D * P);
                auto zero_fill = [&](auto zero, int width) {
                    const int upt = ncp_b / width;             // units per token
                    const int nz = npt * upt;
                    for (int e = lane; e < nz; e += kWave) {
                        const int tk = (int)fdiv((uint32_t)e, w.div_ncp);
                        const int rb = (e - tk * upt) * width;
                        const int col = rb < head_b ? rb : rb + span_b2;
                        wb_store(reinterpret_cast<decltype(zero)*>(gtok0 + (size_t)tk * gm.tokstride + col), zero);
                    }
                };
                if (w.wb_align == 24) {
                    const int nzu = w.ntu - w.nu, nz = npt * nzu;
                    for (int e = lane; e < nz; e += kWave) {
                        const int tk = (int)fdiv((uint32_t)e, w.div_nz);
                        const int zi = e - tk * nzu;
                        const int u = zi < w.u_lo ? zi : zi + w.nu;
                        wb_store(reinterpret_cast<float4*>(gtok0 + (size_t)tk * gm.tokstride + 16 * u), make_float4(0.f, 0.f, 0.f, 0.f));
                    }
                } else if (w.wb_align == 16) zero_fill(make_float4(0.f, 0.f, 0.f, 0.f), 16);
                else if (w.wb_align == 8) zero_fill(make_float2(0.f, 0.f), 8);
                else zero_fill(0.f, 4);
            }
            // the stage is overwritten by the next pass
            wave_lds_sync();
        }
    }

    // ---- parameter gradients: lanes -> wave row -> workgroup row (fixed orders)
    if (KT > 0) {
        my_acc[0] = acc_sf;
#pragma unroll
        for (int k = 0; k < KK; ++k) my_acc[1 + k] = acc_m[k];
    }
    wave_lds_sync();
    if (lane < gm.lpt) {
        // lane (tli = 0, j, sub) sums its column over the tokens of a pass
        const int jj = lane / G, ss = lane - jj * G;
        const int dd = gm.d0 + jj;
        for (int slot = 0; slot < w.nacc; ++slot) {
            float t = 0.f;
            for (int tk = 0; tk < gm.TPP; ++tk) t += lacc[(tk * gm.lpt + lane) * w.nacc + slot];
            if (slot == 0) {
                if (ss == 0) wrow[dd] = t;
            } else {
                const int k = ss + (slot - 1) * G;
                if (k < K) wrow[a.D + dd * K + k] = t;
            }
        }
    }
    // which waves / workgroups the fix-up launch has to visit (always written: the workspace is not initialised)
    const int wave_tail = __ballot(any_tail) != 0ull;
    if (lane == 0) w.wave_flags[blockIdx.x * kWavesPerBlock + wave] = wave_tail;
    const int block_tail = __syncthreads_or(wave_tail);
    if (threadIdx.x == 0) w.block_flags[blockIdx.x] = block_tail;
    const float* rows = reinterpret_cast<const float*>(smem + w.wrow_off);
    for (int i = threadIdx.x; i < PP; i += blockDim.x) {
        float t = 0.f;
        for (int wv = 0; wv < kWavesPerBlock; ++wv) t += rows[wv * PP + i];
        w.partials[(size_t)blockIdx.x * PP + i] = t;
    }
}

template <int KT, int G, bool BIG = false>
__global__ __launch_bounds__(kBlock) void mixture_tok_bwd_kernel(MixArgs a, TokGeom gm, TokBwdArgs w) {
    mixture_tok_bwd_body<KT, G, BIG>(a, gm, w);
}
// the same kernel held to 128 VGPRs (4 waves per SIMD): a few registers of scratch buy twice the resident waves, which is
// what large launches are short of (selected by size in launch_mixture_tok_bwd; measured in profiles/r04_sweep_mixture_bwd.txt)
template <int KT, int G>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(4, 4))) void mixture_tok_bwd_kernel_w4(MixArgs a, TokGeom gm, TokBwdArgs w) {
    mixture_tok_bwd_body<KT, G, false>(a, gm, w);
}

}  // namespace cnf

using namespace cnf;

extern "C" void cnf_set_mixture_bwd_big_mb(int megabytes) {
    cnf::g_tok_bwd_big_mb.store(megabytes < 0 ? cnf::kTokBwdNtMB : megabytes, std::memory_order_relaxed);
}

extern "C" void cnf_set_mixture_bwd_waves(int mode) {
    if (mode >= -1 && mode <= 7) cnf::g_tok_bwd_w4.store(mode, std::memory_order_relaxed);
}

namespace cnf {

// false = shape outside what the token-pass backward is built for (the caller runs the fp64 kernel)
static bool launch_mixture_tok_bwd_with(MixArgs& a, const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                                        float* g_sf, float* g_msf, float* workspace, hipStream_t st, int kt, int force_g) {
    TokGeom gm;
    int G = 1;
    size_t lds_fwd = 0;
    MixArgs probe = a;
    probe.ws_acc = reinterpret_cast<long long*>(8);      // rows may be split freely: the backward has no per-row sums
    probe.ws_cnt = reinterpret_cast<int*>(8);
    if (!make_tok_geom(probe, kt, force_g, gm, G, lds_fwd)) return false;
#ifndef CNF_MIXBWD_ROWS
#define CNF_MIXBWD_ROWS 1
#endif
    if (CNF_MIXBWD_ROWS && !gm.split) {
        // Rows per wave tile.  The forward's choice (~128 items per wave: one 64-token row at S*) leaves a last pass of ONE token
        // behind three of 21 — a full DMA round trip for 312 bytes, a quarter of all passes.  The backward has no per-row sums,
        // so its tiles can be any number of whole rows: the count with the fewest wasted pass slots that still leaves one unit
        // for every wave of the persistent grid (S*: 4 rows = 256 tokens = 13 passes, 94 % full, 4096 units).
        const long want_units = (long)kTokBwdGrid * kWavesPerBlock;
        int best_r = gm.rw;
        double best = (double)(gm.rw * a.N) / (double)((((long)gm.rw * a.N + gm.TPP - 1) / gm.TPP) * gm.TPP);
        const int rcap = (int)std::min<long>(std::min<long>(a.B, 65535 / std::max(a.N, 1)), 64);
        for (int r = 1; r <= rcap; ++r) {
            if (((long)a.B + r - 1) / r < want_units && r > gm.rw) break;
            const long tk = (long)r * a.N;
            const double eff = (double)tk / (double)(((tk + gm.TPP - 1) / gm.TPP) * gm.TPP);
            if (eff > best + 1e-9) {
                best = eff;
                best_r = r;
            }
        }
        gm.rw = best_r;
        gm.ntiles = ((long)a.B + best_r - 1) / best_r;
    }
    const int K = a.K, P = a.P, PP = a.D + a.D * K;
    TokBwdArgs w = {};
    w.g_zout = g_zout; w.g_ldj = g_ldj; w.g_z = g_z; w.g_nn = g_nn; w.partials = workspace;
    w.nacc = kt > 0 ? 1 + kt : 1 + (K + G - 1) / G;
    const size_t tabs = (((size_t)(a.D + a.D * K) * sizeof(BoundTab)) + 15) & ~(size_t)15;
    w.lacc_off = (int)((size_t)kWavesPerBlock * gm.stage_bytes + tabs);
    w.wrow_off = w.lacc_off + (kt > 0 ? 0 : (int)((size_t)kWavesPerBlock * kWave * w.nacc * sizeof(float)));
    if (kt > 0 && (size_t)gm.stage_bytes < (size_t)kWave * w.nacc * sizeof(float)) return false;
    const size_t lds = (size_t)w.wrow_off + (size_t)kWavesPerBlock * PP * sizeof(float);
    if (lds > 65536) return false;
    const int span_b = gm.DA * P * 4;
    const int first_b = (gm.d0 - a.nn_c0) * P * 4;
    const bool base16 = (reinterpret_cast<uintptr_t>(g_nn) & 15) == 0;
    if (base16 && span_b % 16 == 0 && gm.tokstride % 16 == 0 && first_b % 16 == 0 && (gm.contig || gm.slot % 16 == 0)) w.wb_align = 16;
    else if (base16 && gm.contig && span_b == gm.tokstride) w.wb_align = 32;
    else if (base16 && !gm.contig && gm.ncopy > 0 && gm.tokstride % 16 == 0 && gm.slot % 16 == 0 && span_b % 8 == 0 && first_b % 8 == 0) {
        w.wb_align = 24;
        w.u_lo = first_b / 16;
        w.nu = (first_b + span_b + 15) / 16 - w.u_lo;
        w.ntu = gm.tokstride / 16;
        w.lo_half = first_b % 16 != 0;
        w.hi_half = (first_b + span_b) % 16 != 0;
        w.div_nu = make_fastdiv((uint32_t)w.nu);
        w.div_nz = make_fastdiv((uint32_t)std::max(w.ntu - w.nu, 1));
        if (w.nu * 16 > gm.slot || gm.TPP * w.ntu >= 65536) w.wb_align = 8;      // the stage slot must hold the units that are read
    }
    else if ((reinterpret_cast<uintptr_t>(g_nn) & 7) == 0 && span_b % 8 == 0 && gm.tokstride % 8 == 0 && first_b % 8 == 0) w.wb_align = 8;
    else w.wb_align = 4;
    // (modes 1 and 5-7: the builds held to 4 waves per SIMD — K = 8 unrolled: 160 -> 128 VGPRs, 34 of them spilled; no difference for
    // the rolled kernels, which fit anyway)
    const int w4_knob = tok_bwd_w4();
    // (round 6: with the copied-through gradients prefetched the rolled kernels with 2 / 4 lanes per item need 136 VGPRs as compiled
    // freely — three waves per SIMD; their builds held to 128, 5 registers spilled, keep the fourth wave and are the default)
    const bool w4 = w4_knob == 1 || w4_knob >= 5 || (w4_knob < 0 && kt == 0 && G > 1);
    // Large launches (one lane per item, more than kTokBwdNtMB of gradient rows — past what the memory-side cache absorbs; sweep in
    // profiles/r06_mixture_bwd_floor.txt section 4): nontemporal DMA loads and a nontemporal write-back whose every instruction
    // covers whole contiguous kilobytes — passes that are one contiguous span (compact layout, no mask) as they are; the reference
    // layout on the tokens' 16-byte grid by ONE address-ordered loop over gradient spans and zero blocks.  cnf_set_mixture_nt_mb(0)
    // switches all of it off.
    bool big = false;
    {
        const size_t gnn_bytes = (size_t)a.B * a.N * a.nn_D * P * sizeof(float);
        // (its kernels: the rolled one and the unrolled K = 16 one — the two that large launches take by default)
        big = G == 1 && (kt == 0 || kt == 16) && !w4 && mixture_nt_mb() != 0 && gnn_bytes > ((size_t)g_tok_bwd_big_mb.load(std::memory_order_relaxed) << 20);
        if (big && CNF_MIXBWD_MERGED && gm.ncopy > 0 && a.nn_D == a.D && (w.wb_align == 16 || w.wb_align == 24) &&
            gm.TPP * (gm.tokstride / 16) < 65536) {
            w.merged = 1;
            if (w.wb_align == 16) {
                w.u_lo = first_b / 16; w.nu = span_b / 16; w.ntu = gm.tokstride / 16;
                w.lo_half = w.hi_half = 0;
            }
            w.src_stride = gm.contig ? gm.tokstride : gm.slot;
            w.div_ntu = make_fastdiv((uint32_t)w.ntu);
        }
        const bool whole_passes = w.merged || w.wb_align == 32 || (w.wb_align == 16 && gm.contig && span_b == gm.tokstride);
        w.wb_nt = (big && whole_passes) ? 1 : 0;
        gm.nt = big ? 1 : 0;          // (spans of the reference layout too: 280 against 288 us at S*)
    }
    w.nunits = gm.split ? (long)a.B * gm.S * kWavesPerBlock : gm.ntiles;
    w.div_ncp = make_fastdiv((uint32_t)std::max(gm.ncopy * P * 4 / (w.wb_align == 24 ? 8 : w.wb_align), 1));        // zero-fill units per token
    w.div_span = make_fastdiv((uint32_t)span_b);
    if (span_b >= 65536 || gm.TPP * span_b >= 65536 || gm.ncopy * P * gm.TPP >= 65536) return false;
    const int grid = (int)std::min<long>((w.nunits + kWavesPerBlock - 1) / kWavesPerBlock, kTokBwdGrid);
    // workspace (cnf_bwd_workspace_floats(PP) floats): [kTokBwdGrid rows] partials | [kTokBwdGrid rows] fix-up partials |
    // wave flags [4 kTokBwdGrid] | workgroup flags [kTokBwdGrid]
    w.fix_partials = workspace + (size_t)kTokBwdGrid * PP;
    w.wave_flags = reinterpret_cast<int*>(workspace + (size_t)2 * kTokBwdGrid * PP);
    w.block_flags = w.wave_flags + kWavesPerBlock * kTokBwdGrid;
    const dim3 g(grid), b(kBlock);
#define TOK_BWD(KT_, G_)                                                                        \
    do {                                                                                        \
        if (w4) CNF_LAUNCH((mixture_tok_bwd_kernel_w4<KT_, G_>), g, b, lds, st, a, gm, w);      \
        else CNF_LAUNCH((mixture_tok_bwd_kernel<KT_, G_>), g, b, lds, st, a, gm, w);            \
    } while (0)
    if (big && kt == 16) CNF_LAUNCH((mixture_tok_bwd_kernel<16, 1, true>), g, b, lds, st, a, gm, w);
    else if (big) CNF_LAUNCH((mixture_tok_bwd_kernel<0, 1, true>), g, b, lds, st, a, gm, w);
    else if (kt == 4) TOK_BWD(4, 1);
    else if (kt == 8) TOK_BWD(8, 1);
    else if (kt == 16) TOK_BWD(16, 1);
    else if (G == 1) TOK_BWD(0, 1);
    else if (G == 2) TOK_BWD(0, 2);
    else TOK_BWD(0, 4);
#undef TOK_BWD
    const size_t lds_fix = tabs + (size_t)kWavesPerBlock * PP * sizeof(float);
    CNF_LAUNCH(mixture_tok_bwd_fixup_kernel, g, b, lds_fix, st, a, gm, w);
    CNF_LAUNCH(mix_reduce_partials_fix_kernel, dim3(PP), dim3(kBlock), 0, st, workspace, w.fix_partials, w.block_flags, grid, PP,
               a.sf ? g_sf : nullptr, a.msf ? g_msf : nullptr, a.D);
    return true;
}
// false = shape outside what the token-pass backward is built for (the caller runs the fp64 kernel)
bool launch_mixture_tok_bwd(MixArgs& a, const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                            float* g_sf, float* g_msf, float* workspace, hipStream_t st, int force_g) {
    // Which kernel (cnf_set_mixture_bwd_waves; profiles/r04_sweep_mixture_bwd.txt): since the fp64 tail arithmetic left the streaming
    // kernel (round 4) the ROLLED run-time-K kernel — ~100 VGPRs, 5 waves per SIMD, lane-private LDS sums — beats the unrolled
    // register-slot kernels (K = 8: 160 VGPRs) at every measured shape: S* 357 -> 331-340 us, configs[1] 59.4 -> 55, Zinc edges
    // 79 -> 54, Zinc nodes 35 -> 26, graph colouring (K = 16) 32 -> 23, a 1024-set training batch 23 -> 17-18.  Lanes per item G
    // by the amount of work: one lane per item from ~400 k transformed elements on, four below (more waves for
    // small launches); K > 32 needs four for its stage to fit.  Modes 0 / 1 keep the unrolled kernels (natural registers / held
    // to 4 waves per SIMD) for A/B runs, 2-4 force G = 1 / 2 / 4.
    int kt = 0;
    const int knob0 = tok_bwd_w4();
    if (knob0 == 0 || knob0 == 1) {
        kt = (a.K == 4 || a.K == 8 || a.K == 16) ? a.K : 0;
    } else if (knob0 >= 2) {
        force_g = 1 << ((knob0 - 2) % 3);
    } else if (force_g <= 0) {
        const int da = a.per_item_mask ? a.D : __builtin_popcountll(a.act_bits);
        const long items = (long)a.B * a.N * std::max(da, 1);
        // (the one shape family where the unrolled kernel still wins: K = 16 at ~10^6 tokens — 578 us against 638 / 838 with four / two
        // lanes per item, whose stages leave LDS room for fewer workgroups: tools/mixture_bwd_variants.py)
        if (a.K == 16 && items >= 2000000L &&
            launch_mixture_tok_bwd_with(a, g_zout, g_ldj, g_z, g_nn, g_sf, g_msf, workspace, st, 16, 0))
            return true;
        // (round 5, profiles/r05_mixture_bwd_floor.txt: at S* with K = 8 two lanes per item beat one — 320-334 against 347-371 us on two
        // boxes: passes of 10 tokens instead of 21 move the same bytes faster, and the kernel is 90 % data movement; at K = 4 one lane
        // per item stays ahead, 207 against 229 us, profiles/r04_mixture_bwd_variants.txt)
        // (round 6, profiles/r06_mixture_bwd_floor.txt, with the copied-through gradients loaded ahead of the write-back: one lane per
        // item from ~400 k transformed elements on — configs[1] 60.5 against 63 us — except the reference layout at S* size with 8
        // or more mixtures, 316-320 us with two lanes against 338-345; the compact layout takes one lane there: 231-235 against 250)
        // (later in round 6, section 4 of the same file: with the reference layout's tokens written in address order by nontemporal
        // stores one lane per item wins there too — S* 280 us against 334-343 with two lanes, 316-327 before)
        int g = items >= 400000L ? 1 : 4;
        if (a.K > 32) g = 4;
        while (g > 1 && g > a.K) g >>= 1;
        // the rule's G, or the next one whose stage fits LDS
        for (; g <= 4; g <<= 1)
            if (launch_mixture_tok_bwd_with(a, g_zout, g_ldj, g_z, g_nn, g_sf, g_msf, workspace, st, 0, g)) return true;
        return false;
    }
    return launch_mixture_tok_bwd_with(a, g_zout, g_ldj, g_z, g_nn, g_sf, g_msf, workspace, st, kt, force_g);
}
}  // namespace cnf
