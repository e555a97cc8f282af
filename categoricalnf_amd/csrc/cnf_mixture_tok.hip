// Logistic-mixture CDF coupling, fp32 "token-pass" kernel (math mode 1): forward, Newton inverse, log-det and the
// optional NLL epilogue, following mixture_cdf_layer.py:95-180, 197-276 of the reference.
//
// Data movement.  The coupling sub-network's output holds, per token (b, n), D blocks of P = 2 + 3K floats; only the
// blocks of the DA transformed channels (a contiguous channel range: a channel mask keeps the first or the last
// channels) are needed, and they are ONE contiguous span of DA * P floats per token.  A wave works in passes of
// TPP = 64 / (DA * G) consecutive tokens: the spans of the pass are copied global -> LDS by the DMA path
// (global_load_lds_dwordx4: 16 bytes per lane, 1 KiB per instruction, no staging registers, no ds_write), each token
// into its own slot at the span's own 16-byte phase, so every global address is 16-byte aligned and every 128-byte
// line of a span is fetched once.  Without a mask (D == DA) the whole pass is one contiguous span.  Then lane
// (token, channel, g) reads its P-float row from LDS; G = 1, 2 or 4 lanes share an item (run-time K only) and
// split its mixtures, so that a pass of the K = 51 language-model rows (620 B each) still fits ~10 KiB per wave.
//
// Work decomposition.  Many short rows (samples): a wave owns `rw` whole rows and adds the per-item log-det terms
// to per-row LDS accumulators in 31.32 fixed point (integer adds are associative: the sum is independent of the
// order, bit-reproducible).  Few long rows (training-sized batches of graphs / sentences): S workgroups x 4 waves
// share one row, so that B = 128 rows still put thousands of waves on the 256 CUs; a workgroup combines its four
// partial sums in wave order and, when S > 1, adds them to the row's fixed-point words in a caller-provided
// workspace with device-scope integer atomics; the workgroup that draws the last ticket of a row finishes it and
// leaves the workspace zeroed for the next launch.
//
// Arithmetic (unchanged from round 1): fp32 on hardware exp2 / log2 / rcp with BOTH tails as
// sums of positive terms; elements with u or 1 - u below 1e-9, or an underflowing PDF sum, take the fp64 branch that
// reproduces the reference's clamps.
#include "cnf_mixture_tok.h"
#include "cnf_f64_math.h"

#include <algorithm>
#include <atomic>
#include <climits>
#include <map>
#include <mutex>
#include <utility>

namespace cnf {

// ED > 0: the ActNorm + 1x1 convolution of the next flow step (D = ED channels) are applied to the coupling's output
// while it is on chip (forward only): the [B,N,D] round trip between the two kernels (8 B/elem) disappears.  Same
// arithmetic, in the same order, as actnorm_invconv_kernel (cnf_linear.hip): results are bit-identical to the chain.
//
// Mixtures per lane.  KT > 0: a lane keeps KT mixtures in registers and every loop over them is unrolled — with PR ==
// false these are exactly the K = KT mixtures of the item (G == 1); with PR == true ("predicated slots") the K <= KT * G
// run-time mixtures of an item are dealt to its G lanes, lane `sub` holding k = sub + G * i in slot i, and a slot past
// K repeats mixture K - 1 with weight zero (so minima / maxima over the slots are those over the mixtures and nothing
// outside the row is read).  KT == 0: the general fallback, a rolled loop over the LDS row.  Round 2 ran every K other
// than 4 / 8 / 16 on the rolled loop: three dependent LDS reads and an `s_waitcnt lgkmcnt(0)` per mixture and Newton
// evaluation, one serial chain per wave (PTB shape, K = 51: inverse 45 us against a 25 us forward).
// Register budget: the K = 8 inverse (configs[1]) needs 97 VGPRs as compiled freely, one more than five waves per SIMD
// allow; asking for five costs nothing in the loop (no spills) and buys the fifth wave.
#ifndef CNF_TOK_PRE_F32
#define CNF_TOK_PRE_F32 0      // 1: the fp32 kernels keep their DMA source offsets too.  Measured (+4 VGPRs everywhere, four
                               // instantiations lose a wave per SIMD): S* forward 102 -> 101 us, inverse 109.4 -> 113.3; configs[1]
                               // 18.8 -> 18.9 / 23.3 -> 22.5: they are not short of VALU issue slots the way the fp64 kernels are
#endif
#ifndef CNF_X64_FWD_WAVES
#define CNF_X64_FWD_WAVES 4
#endif
#ifndef CNF_X64_FWD_UNROLL
#define CNF_X64_FWD_UNROLL 2
#endif
#ifndef CNF_X64_ONE_EVAL
#define CNF_X64_ONE_EVAL 1
#endif
#ifndef CNF_X64_INV_WAVES
#define CNF_X64_INV_WAVES 1
#endif
// stores of z' (A/B build -DCNF_MIXFWD_NT_STORES: nontemporal)
__device__ __forceinline__ void zstore(float* p, float v) {
#ifdef CNF_MIXFWD_NT_STORES
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}
constexpr int kTokPre = 8;      // DMA instructions per pass whose source offsets the fp64 kernels keep (8 KiB stages)
constexpr int tok_min_waves(int kt, bool reverse, int g, bool pr, bool x64 = false, bool nll = false) {
    if (x64) return reverse ? CNF_X64_INV_WAVES : CNF_X64_FWD_WAVES;
    if (kt == 16 && nll && g == 1 && !pr) return 4;        // 129 VGPRs as compiled freely (round 6: the escape words of the row sums)
    return (kt == 8 && reverse && g == 1 && !pr) ? 5 : 1;
}

//
// X64 (math mode 0, the reference's precision: mixture_cdf_layer.py:62,173-178 compute in fp64): the same passes with
// the arithmetic of the fp64 kernel of cnf_mixture.hip — every element takes what is the rare branch of the fp32
// forward; the inverse polishes the fp32 Newton root with safeguarded Newton steps in fp64 (quadratic convergence: two
// evaluations from a 1e-7 start) inside the widened component-quantile bracket.  KT > 0 only.
template <int KT, bool REVERSE, int G, bool NLL, int ED = 0, bool PR = false, bool X64 = false>
__global__ __launch_bounds__(kBlock) __attribute__((amdgpu_waves_per_eu(tok_min_waves(KT, REVERSE, G, PR, X64, NLL))))
void mixture_tok_kernel(MixArgs a, TokGeom gm) {
    static_assert(!X64 || (KT > 0 && !NLL && ED == 0), "fp64 arithmetic: register slots, plain coupling");
    static_assert(G == 1 || KT == 0 || PR, "several lanes per item: run-time K (rolled loop) or predicated slots");
    static_assert(!(NLL && REVERSE), "the NLL epilogue belongs to the forward pass");
    static_assert(ED == 0 || (!REVERSE && !NLL), "the ActNorm + convolution epilogue belongs to a forward pass inside a flow");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63;
    const int K = (KT > 0 && !PR) ? KT : a.K;
    const int P = a.P;
    char* stage_b = smem + (size_t)wave * gm.stage_bytes;
    BoundTab* sf_tab = reinterpret_cast<BoundTab*>(smem + gm.tab_off);
    BoundTab* msf_tab = sf_tab + a.D;
    // accumulators: split == 0: [wave][rw][2] fixed-point row sums, then [wave][rw][2] fp64 escape words for the terms the
    // fixed-point words cannot take (|term| >= 2^14, +-inf, NaN: cnf_common.h fix_pair_add); split == 1: [wave][2] fp64 wave partials
    long long* rowacc = reinterpret_cast<long long*>(smem + gm.acc_off) + (size_t)wave * gm.rw * 2;
    double* rowbig = reinterpret_cast<double*>(smem + gm.acc_off) + (size_t)(kWavesPerBlock + wave) * gm.rw * 2;
    double* wpart = reinterpret_cast<double*>(smem + gm.acc_off);
    // epilogue: constants [bias D | e^scales D | W D*D | sum scales] and this wave's strip of one pass of tokens
    float* etab = reinterpret_cast<float*>(smem + gm.epi_off);
    float* ep = etab + (2 * ED + ED * ED + 4) + (size_t)wave * gm.TPP * ED;
    // ---- this wave's tile: tokens [0, ntok) counted from (row0, n_first)
    int row0, nrows, n_first, ntok;
    if (!gm.split) {
        const long tile = (long)blockIdx.x * kWavesPerBlock + wave;
        row0 = (int)min(tile * gm.rw, (long)a.B);
        nrows = min(gm.rw, a.B - row0);         // 0: a wave past the last tile (it still meets the barrier below)
        n_first = 0;
        ntok = nrows * a.N;
    } else {
        // the row's passes are dealt to its 4 S waves in contiguous, balanced runs (they differ by at most one pass)
        row0 = (int)(blockIdx.x / (unsigned)gm.S);
        const int seg = (int)blockIdx.x - row0 * gm.S;
        const int w = seg * kWavesPerBlock + wave, nw = gm.S * kWavesPerBlock;
        const int p_lo = (w * gm.ppr) / nw, p_hi = ((w + 1) * gm.ppr) / nw;
        n_first = p_lo * gm.TPP;
        ntok = max(0, min(a.N, p_hi * gm.TPP) - n_first);
        nrows = 1;
    }
    const size_t tok_g0 = (size_t)row0 * a.N + n_first;
    // row scalars of the row this lane finishes, fetched now: loaded in `finish` they would cost every wave one more
    // serial memory round trip at its very end (whole rows per wave; a split row is finished by one workgroup only)
    const bool own_row = !gm.split && lane < nrows;
    float my_ldj = 0.f, my_len = (float)a.N;
    if (own_row) {
        if (a.ldj_in) my_ldj = a.ldj_in[row0 + lane];
        if (NLL && a.length) my_len = a.length[row0 + lane];
    }
    const float* z_tile = a.z + tok_g0 * a.D;
    float* zo_tile = a.z_out + tok_g0 * a.D;
    const float* pad_tile = a.pad ? a.pad + tok_g0 : nullptr;
    const char* nn_lo = reinterpret_cast<const char*>(a.nn);
    const char* nn_last = nn_lo + ((size_t)a.B * a.N * a.nn_D * P - 4) * sizeof(float);      // last aligned 16-byte chunk
    const char* span0 = nn_lo + (tok_g0 * a.nn_D + (gm.sd0 - a.nn_c0)) * (size_t)P * sizeof(float);       // first token's staged span

    // lane -> (token in pass, channel, share of the mixtures)
    const int tli = (int)fdiv((uint32_t)lane, gm.div_lpt);
    const int rem = lane - tli * gm.lpt;
    const int j = rem / G, sub = rem - j * G;
    const int d = gm.d0 + j;
    // the first pass's DMA goes out BEFORE the workgroup builds its tables: the table inputs are global loads too, and a
    // wave that first waited for them and the barrier started its DMA one memory round trip late (split rows: every
    // workgroup, on the critical path of launches that are a single round of workgroups)
    int my_pos = 0;
    // X64: the per-lane source offsets of a pass's DMA instructions do not change from pass to pass when every span
    // starts at the same 16-byte phase (token stride a multiple of 16 bytes); stage_pass recomputes them with a
    // division per instruction — ~35 VALU instructions each, 250 per pass at S*, which an issue-bound kernel pays for
    // They are kept in LDS ([instruction][lane] behind the accumulators; registers are what the fp64 kernels are short of).
    uint32_t* goff = reinterpret_cast<uint32_t*>(smem + gm.epi_off) + (size_t)wave * (kTokPre * kWave) + lane;
    bool pre = false;
    int phase = 0;
    constexpr bool kPreOk = X64 || (CNF_TOK_PRE_F32 && ED == 0);
    if constexpr (kPreOk) {
        pre = gm.pre != 0;
        phase = (int)(reinterpret_cast<uintptr_t>(span0) & 15);
        if (pre) {
            for (int i = 0; i < kTokPre; ++i) {
                const uint32_t cb = (uint32_t)(i * kWave + lane) << 4;
                const uint32_t sl = fdiv(cb, gm.div_slot);
                goff[i * kWave] = sl * (uint32_t)gm.tokstride + (cb - sl * (uint32_t)gm.slot);
            }
        }
    }
    auto stage = [&](const char* pass_addr, int npt) {
        if constexpr (!kPreOk) {
            return stage_pass(gm, stage_b, pass_addr, nn_last, npt, lane, tli, j + (gm.d0 - gm.sd0), P);
        } else if (pre) {
            const int ni = (npt * gm.slot + 1023) >> 10;
            const char* base = pass_addr - phase;                                   // wave-uniform, 16-byte aligned
            const uint32_t last = (uint32_t)min((ptrdiff_t)(nn_last - base), (ptrdiff_t)0x7fffffff);
#pragma unroll
            for (int i = 0; i < kTokPre; ++i) {
                if (i < ni) {
                    const uint32_t off = min(goff[i * kWave], last);
                    dma_1k(base + off, stage_b + (i << 10), gm.nt);
                }
            }
            return tli * gm.slot + phase + (j + (gm.d0 - gm.sd0)) * P * 4;
        } else {
            return stage_pass(gm, stage_b, pass_addr, nn_last, npt, lane, tli, j + (gm.d0 - gm.sd0), P);
        }
    };
    if (ntok > 0) my_pos = stage(span0, min(gm.TPP, ntok));
    if (ED > 0) {
        for (int i = threadIdx.x; i < ED; i += blockDim.x) {
            etab[i] = a.e_bias[i];
            etab[ED + i] = expf(a.e_scales[i]);
        }
        for (int i = threadIdx.x; i < ED * ED; i += blockDim.x) etab[2 * ED + i] = a.e_w[i];
        if (threadIdx.x == 0) {
            float ssum = 0.f;
            for (int i = 0; i < ED; ++i) ssum += a.e_scales[i];
            etab[2 * ED + ED * ED] = ssum;
        }
    }
    for (int i = threadIdx.x; i < a.D; i += blockDim.x)
        if (a.sf) sf_tab[i] = make_bound(a.sf[i]);
    for (int i = threadIdx.x; i < a.D * K; i += blockDim.x)
        if (a.msf) msf_tab[i] = make_bound(a.msf[i]);
    if (!gm.split)
    {
        for (int i = lane; i < gm.rw * 2; i += kWave) {
            rowacc[i] = 0;
            rowbig[i] = 0.0;
        }
    }
    __syncthreads();
    if (!gm.split && nrows <= 0) return;            // no barrier follows in this mode
    const BoundTab* mt = msf_tab + d * K;
    const PriorConst prior = a.prior;
    // slot i of this lane -> mixture index, and whether the slot holds a mixture of its own
    auto kidx = [&](int i) { return PR ? min(sub + G * i, K - 1) : i; };
    auto kown = [&](int i) { return !PR || sub + G * i < K; };
    auto bound_of = [&](float v, const BoundTab& b) { return X64 ? apply_bound_exact(v, b) : apply_bound(v, b); };
    bool bad = false, range = false, badl = false;
    double acc_ldj = 0.0, acc_nlp = 0.0;          // split mode: this lane's running sums

    // one pass of tokens [tp, tp + npt) of the tile, its parameter rows staged at `stg`; `zload` / `padload` fetch a latent
    // (index relative to the tile) / a token's padding value from wherever the caller keeps them
    auto pass_body = [&](int tp, int npt, char* stg, int my_pos_in, auto zload, auto padload) {
        const bool valid = tli < npt;
        const int tokl = tp + (valid ? tli : 0);
        int rl = 0, n = n_first + tokl;
        if (!gm.split) {
            rl = (int)fdiv((uint32_t)tokl, gm.div_n);
            n = tokl - rl * a.N;
        }
        const float pv = pad_tile ? padload(tokl) : 1.f;
#if defined(CNF_MIXFWD_ABLATE) && CNF_MIXFWD_ABLATE >= 3
        const float x = 0.f;            // A/B build: no loads of z either
#else
        const float x = valid ? zload((size_t)tokl * a.D + d) : 0.f;
#endif
        bool active = valid;
        if (a.per_item_mask) active = active && mask_at(a.mask, a.mr, a.mc, n, d) == 0.f;
        if (a.pad_in_transform && pv == 0.f) active = false;
#ifdef CNF_MIXFWD_ABLATE
        // A/B builds (tools/mixture_fwd_floor.py, profiles/r06_mixture_fwd_floor.txt): 1 = the data movement alone — rows staged,
        // latents in and out, row sums — with the arithmetic compiled out: the floor of this kernel's access pattern and launch
        // geometry; 2 = without the stores of z' as well; 3 = without the loads of z either (the DMA stream and the launch alone)
        active = false;
#endif
        float* my = reinterpret_cast<float*>(stg + (valid ? my_pos_in : 0));
        float of = a.pad_output ? x * pv : x;       // a lane that transforms nothing copies its element through
        float contrib = 0.f;
        double contrib64 = 0.0;
        bool use64 = false;
        // log-space mixture log-pdf on the staged row, for a direct sum that underflows
        auto logspace_pdf = [&](double xd, float mx, double sed) {
            // log-space form (:217-223)
            const double lse_pi = (double)mx + log(sed);
            double m = -INFINITY;
#pragma clang loop unroll(disable)
            for (int k = 0; k < K; ++k) {
                const float lsf = a.msf ? bound_of(my[2 + 2 * K + k], mt[k]) : my[2 + 2 * K + k];
                const double zd = (xd - (double)my[2 + K + k]) * exp(-(double)lsf);
                const double tk = (double)my[2 + k] - lse_pi + zd - (double)lsf - 2.0 * softplus64(zd);
                m = (tk != tk || m != m) ? (double)NAN : fmax(m, tk);       // torch.max keeps a NaN
            }
            double ssum = 0.0;
#pragma clang loop unroll(disable)
            for (int k = 0; k < K; ++k) {
                const float lsf = a.msf ? bound_of(my[2 + 2 * K + k], mt[k]) : my[2 + 2 * K + k];
                const double zd = (xd - (double)my[2 + K + k]) * exp(-(double)lsf);
                ssum += exp((double)my[2 + k] - lse_pi + zd - (double)lsf - 2.0 * softplus64(zd) - m);
            }
            return isinf(m) ? m : m + log(ssum);      // torch.logsumexp: every term -inf (a latent of -inf) gives -inf
        };
        if (active && REVERSE) {
            // ---- inverse (:125-134, :235-264) in fp32: safeguarded Newton on the two-sided CDF.  u = sigmoid(v)
            // is clamped to [1e-5, 1 - 1e-5] by the reference, so the root lies where fp32 sums of positive
            // terms are accurate to ~1e-6 relative in BOTH tails: solve cdf(x) = u se for u <= 1/2 and
            // ccdf(x) = (1 - u) se otherwise.  A relative error eps of the sum moves x by ~eps * s_k.
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
            // Only the side that is solved for is accumulated in the iterations: with the standardised argument's sign
            // flipped for u > 1/2, `side` below is cdf (u <= 1/2) or ccdf (u > 1/2) from the same instructions, and
            // log2(e) rides in the reciprocal scales, so that an evaluation is 10 plain + 2 transcendental
            // instructions per mixture (round 2: 13 + 2).
            const float sgn = upper ? -1.f : 1.f;
            constexpr int KK = KT > 0 ? KT : 1;
            float wr[KK], isr[KK], mur[KK];
            float mx = -INFINITY;
            if (KT > 0) {
#pragma unroll
                for (int i = 0; i < KK; ++i) mx = fmaxf(mx, my[2 + kidx(i)]);
                mx = qmax<G>(mx);
            } else {
                for (int k = sub; k < K; k += G) mx = fmaxf(mx, my[2 + k]);
                mx = qmax<G>(mx);
            }
            float se = 0.f, spread = 0.f, lb = INFINITY, ub = -INFINITY, qws = 0.f;
            float smin = INFINITY;                  // X64: smallest component scale (the polish's tolerance)
            auto setup = [&](int k, int slot, bool own) {
                const float lsk = my[2 + 2 * K + k];
                const float ls = a.msf ? apply_bound(lsk, mt[k]) : lsk;
                float w = __builtin_amdgcn_exp2f((my[2 + k] - mx) * kLog2eF);
                if (PR && !own) w = 0.f;
                const float sk = __builtin_amdgcn_exp2f(ls * kLog2eF);
                const float ik = __builtin_amdgcn_rcpf(sk) * (sgn * kLog2eF);      // signed, in log2 units
                const float mk = my[2 + K + k];
                // every component's own u-quantile: the mixture quantile lies between their min and max
                const float qk = fmaf(sk, logit_u, mk);
                se += w;
                spread += (PR && !own) ? 0.f : sk;
                if (X64) smin = fminf(smin, sk);
                lb = fminf(lb, qk);
                ub = fmaxf(ub, qk);
                qws = fmaf(w, qk, qws);
                if (KT > 0) {
                    wr[slot < KK ? slot : 0] = w; isr[slot < KK ? slot : 0] = ik; mur[slot < KK ? slot : 0] = mk;
                } else {
                    my[2 + k] = w;              // run-time K: the constants replace the raw row in LDS
                    my[2 + 2 * K + k] = ik;
                }
            };
            if (KT > 0) {
#pragma unroll
                for (int i = 0; i < KK; ++i) setup(kidx(i), i, kown(i));
            } else {
                for (int k = sub; k < K; k += G) setup(k, 0, true);
            }
            if (G > 1) {
                se = qsum<G>(se); spread = qsum<G>(spread); qws = qsum<G>(qws);
                lb = qmin<G>(lb); ub = qmax<G>(ub);
                if (X64) smin = qmin<G>(smin);
            }
            const float lb0 = lb, ub0 = ub;         // every component's own u-quantile lies in here, and so does the root
            const float target = (upper ? uc : u) * se;
            const float tol_scale = 1e-7f * spread;
            float xb = fminf(fmaxf(qws * __builtin_amdgcn_rcpf(se), lb), ub);
            float dx_prev = ub - lb, dens = 0.f, diff = INFINITY;
            auto eval = [&](float xq, float& f_out, float& dens_out) {
                float side = 0.f, dn = 0.f;
                auto one = [&](float wk, float ik, float mk) {
                    const float zk = (xq - mk) * ik;
                    const float e = __builtin_amdgcn_exp2f(-fabsf(zk));
                    const float rr = __builtin_amdgcn_rcpf(1.f + e);
                    const float er = e * rr;
                    side = fmaf(wk, zk >= 0.f ? rr : er, side);
                    dn = fmaf(wk * fabsf(ik), er * rr, dn);
                };
                if (KT > 0) {
#pragma unroll
                    for (int k = 0; k < KK; ++k) one(wr[k], isr[k], mur[k]);
                } else {
                    for (int k = sub; k < K; k += G) one(my[2 + k], my[2 + 2 * K + k], my[2 + K + k]);
                }
                if (G > 1) {
                    side = qsum<G>(side); dn = qsum<G>(dn);
                }
                f_out = sgn * (side - target);                      // increasing in x either way
                dens_out = dn * kLn2F;
            };
#ifdef CNF_MIX32_COUNT_ITERS
            int n_eval32 = 0;               // diagnostic build (tools/mix32_iters.py): evaluations of the fp32 Newton loop
#endif
            for (int iter = 0; iter < 64; ++iter) {
                float f;
                eval(xb, f, dens);
#ifdef CNF_MIX32_COUNT_ITERS
                ++n_eval32;
#endif
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
            if constexpr (X64) {
                // ---- the reference's precision: u, the weights and the scales in fp64 (:125-134), then safeguarded Newton
                // in fp64 from the fp32 root.  Its error ~1e-7 (|x| + sum s_k) squares with every accepted step, so the
                // second evaluation usually ends the loop; the bracket is the fp32 quantile bracket, widened past
                // anything fp32 rounding of q_k can move it, and the bisection fallback of rtsafe stays in place.
                float ls_raw = my[1];
                if (a.sf) ls_raw = apply_bound_exact(ls_raw, sf_tab[d]);
                const double lsd = (double)ls_raw;
                const double vd = (double)x * exp(-lsd) - (double)t;
                // sigmoid(v) and softplus(v) + softplus(-v) (:127-129) from ONE exponential E = e^{-|v|}: u = 1 / (1 + E) or
                // E / (1 + E); softplus(|v|) = |v| + log1p(E) — F.softplus returns its argument above the threshold 20 —
                // and softplus(-|v|) = log1p(E)
                const double av = fabs(vd);
                const double Ev = exp(-av);
                const double rv1 = rcp64(1.0 + Ev);
                double ud = vd >= 0.0 ? rv1 : Ev * rv1;
                const double l1pE = log1p64_unit(Ev);
                double mldj = av > 20.0 ? av + l1pE : av + (l1pE + l1pE);
                ud = fmin(fmax(ud, 1e-5), 1.0 - 1e-5);
                // NaN in: NaN out (torch.clamp keeps a NaN, fmin / fmax return their other operand: set after the clamp)
                if (!(av == av)) { ud = vd; mldj = vd; }
                if (!(ud > 0.0 && ud < 1.0)) range = true;
                // With ONE evaluation per element in the usual case nothing is reused between evaluations, so the weights e^{lp - max}
                // and the inverse scales e^{-ls} are not kept: every evaluation is one rolled pass over the staged row like the
                // forward's (no register arrays: 2K doubles per lane cost the unrolled form its third and fourth wave per SIMD — 226
                // VGPRs at K = 8, spills at K = 16); the rare further evaluations pay two exponentials per mixture more.
                double sed = 0.0;
                const double widen = 1e-4 * (12.0 * (double)spread + fabs((double)lb0) + fabs((double)ub0));
                double lbd = (double)lb0 - widen, ubd = (double)ub0 + widen;
                double xq = fmin(fmax((double)xb, lbd), ubd);
                double dxp = ubd - lbd, dn = 0.0;
                const double tol_s = 1e-11 * (double)smin;
                int n_eval = 0;
                for (int iter = 0; iter < 100; ++iter) {
                    double c = 0.0, ddn = 0.0, se = 0.0;
                    dn = 0.0;
                    ++n_eval;
#pragma unroll CNF_X64_FWD_UNROLL
                    for (int i = 0; i < KK; ++i) {
                        const int k = kidx(i);
                        const float lsk = my[2 + 2 * K + k];
                        const float lsf = a.msf ? apply_bound_exact(lsk, mt[k]) : lsk;
                        double w_ = exp((double)my[2 + k] - (double)mx);
                        if (PR && !kown(i)) w_ = 0.0;
                        const double is_ = exp(-(double)lsf);
                        const double zk = (xq - (double)my[2 + K + k]) * is_;
                        const double e = exp(-fabs(zk));
                        const double rr = rcp64(1.0 + e);
                        const double sg = zk >= 0.0 ? rr : e * rr;
                        const double pk = w_ * is_ * (e * rr * rr);
                        se += w_;
                        c += w_ * sg;
                        dn += pk;
                        ddn += pk * (is_ * fma(-2.0, sg, 1.0));       // d/dx of the component's density: w s^-2 sigma''(z)
                    }
                    if (G > 1) {
                        se = qsum64<G>(se); c = qsum64<G>(c); dn = qsum64<G>(dn); ddn = qsum64<G>(ddn);
                    }
                    sed = se;
                    const double tgt = ud * sed;
                    const double f = c - tgt;
                    double nx;
                    if (f > 0.0) {
                        nx = 0.5 * (xq + lbd);
                        ubd = xq;
                    } else {
                        nx = 0.5 * (xq + ubd);
                        lbd = xq;
                    }
                    bool newton = false;
                    if (dn > 0.0 && fabs(2.0 * f) <= fabs(dxp * dn)) {
                        const double xn = xq - f / dn;
                        if (xn >= lbd && xn <= ubd) {
                            nx = xn;
                            newton = true;
                        }
                    }
                    const double dd = fabs(nx - xq);
                    // A Newton step of length dd leaves an error <= dd^2 max|pdf'/pdf| / 2 <= dd^2 / (2 s_min) (every logistic
                    // component has |p'/p| <= 1/s): below 4.47e-6 s_min that is under the 1e-11 s_min the loop asks for, and the
                    // density at the new point follows to first order from its derivative (relative error <= 2e-11).  From the
                    // fp32 root (error ~1e-7 (|x| + sum s)) this is the usual case: one fp64 evaluation per element
                    // (profiles/r05_mixture_fp64_newton_evaluations.txt: 1.00 on average, 1.05 with latents out to 8 sigma).
                    if (CNF_X64_ONE_EVAL && newton && dd <= 4.47e-6 * (double)smin) {
                        dn = fma(ddn, nx - xq, dn);
                        xq = nx;
                        break;
                    }
                    dxp = dd;
                    xq = nx;
                    // the density of the last evaluation stands in for the one at the root: relative error <= dd / s_min
                    if (!(dd > fmax(tol_s, 4e-16 * fabs(xq)))) break;
                }
                const double lpdfd = dn > 1e-290 ? log64_pos(dn / sed) : logspace_pdf(xq, mx, sed);
                of = (float)xq;
#ifdef CNF_MIX64_COUNT_ITERS
                of = (float)n_eval;             // diagnostic build (tools/mix64_iters.py): evaluations of the fp64 loop
#endif
                if (a.pad_output) of = of * pv;
                contrib64 = lsd + mldj + lpdfd;
                use64 = true;
            } else {
            const float lpdf = (__builtin_amdgcn_logf(dens) - __builtin_amdgcn_logf(se)) * kLn2F;
            of = xb;
#ifdef CNF_MIX32_COUNT_ITERS
            of = (float)n_eval32;
#endif
            if (a.pad_output) of = of * pv;
            contrib = log_s + mixt_ldj + lpdf;
            }
        }
        if (active && !REVERSE) {
            const float t = my[0];
            float log_s = my[1];
            if (a.sf) log_s = apply_bound(log_s, sf_tab[d]);
            constexpr int KK = KT > 0 ? KT : 1;
            float lp[KK], mu[KK], lsr[KK];
            float mx = -INFINITY;
            if (X64) {
                // the fp64 loop below reads the row where it lies: no register copies of it
                for (int i = 0; i < KK; ++i) mx = fmaxf(mx, my[2 + kidx(i)]);
                mx = qmax<G>(mx);
            } else if (KT > 0) {
#pragma unroll
                for (int i = 0; i < KK; ++i) {
                    const int k = kidx(i);
                    lp[i] = my[2 + k];
                    mu[i] = my[2 + K + k];
                    lsr[i] = my[2 + 2 * K + k];
                }
#pragma unroll
                for (int i = 0; i < KK; ++i) mx = fmaxf(mx, lp[i]);
                mx = qmax<G>(mx);
            } else {
                for (int k = sub; k < K; k += G) mx = fmaxf(mx, my[2 + k]);
                mx = qmax<G>(mx);
            }
            float se = 0.f, cdf = 0.f, ccdf = 0.f, pdf = 0.f;
            double reg = 0.0;
            // the reference's u -> (z, log-det, regulariser) in fp64 from the three fp64 sums (:100-123, :217-233)
            auto wide_tail = [&](double sed, double cdfd, double pdfd) {
                const double xd = (double)x;
                const double ud = cdfd / sed;
                double lpdfd;
                if (pdfd > 1e-290) {
                    lpdfd = X64 ? log64_pos(pdfd / sed) : log(pdfd / sed);
                } else {
                    lpdfd = logspace_pdf(xd, mx, sed);
                }
                // safe_log (:266-268) on arguments that are positive normals after its clamp
                // (the fp32 kernels' rare branch keeps the library's log: 9 VGPRs fewer in kernels that sit at an occupancy step)
                const double lud = X64 ? log64_pos(fmax(ud, 1e-22)) : safe_log(ud);
                const double l1ud = X64 ? log64_pos(fmax(1.0 - ud, 1e-22)) : safe_log(1.0 - ud);
                if (a.use_reg) {
                    const double r1 = lud / kLn10, r2 = l1ud / kLn10;
                    reg = (fmin(r1, -a.reg_max) + a.reg_max) + (fmin(r2, -a.reg_max) + a.reg_max);
                }
                double yd = lud - l1ud;
                // below the clamp the reference's own form (:273); a NaN u stays a NaN (torch.clamp keeps it, fmax does not)
                if (!(ud >= 1e-22)) yd = ud != ud ? ud : -safe_log(1.0 / ud - 1.0);
                of = (float)((yd + (double)t) * exp((double)log_s));
                contrib64 = (double)log_s + (-lud - l1ud) + lpdfd + reg * a.reg_factor;
                use64 = true;
            };
            if constexpr (X64) {
                double sed = 0.0, cdfd = 0.0, pdfd = 0.0;
                // the reference's arithmetic on this lane's slots; log_s / log-scale bounds by the library tanh
                if (a.sf) log_s = apply_bound_exact(my[1], sf_tab[d]);
                const double xd = (double)x;
                // a rolled loop over the LDS row (register arrays under a run-time index become select chains)
#pragma unroll CNF_X64_FWD_UNROLL
                for (int i = 0; i < KK; ++i) {
                    const int k = kidx(i);
                    const float lsk = my[2 + 2 * K + k];
                    const float lsf = a.msf ? apply_bound_exact(lsk, mt[k]) : lsk;
                    double wd = exp((double)my[2 + k] - (double)mx);
                    if (PR && !kown(i)) wd = 0.0;
                    const double isd = exp(-(double)lsf);
                    const double zd = (xd - (double)my[2 + K + k]) * isd;
                    const double ed = exp(-fabs(zd));
                    const double rd = rcp64(1.0 + ed);          // 1 + e in [1, 2]: no scaling, correctly rounded on test
                    sed += wd;
                    cdfd += wd * (zd >= 0.0 ? rd : ed * rd);
                    pdfd += wd * isd * (ed * rd * rd);
                }
                if (G > 1) {
                    sed = qsum64<G>(sed); cdfd = qsum64<G>(cdfd); pdfd = qsum64<G>(pdfd);
                }
                wide_tail(sed, cdfd, pdfd);
            } else {
            auto one = [&](float lpk, float muk, float lsk, int k, bool own) {
                const float ls = a.msf ? apply_bound(lsk, mt[k]) : lsk;
                const float inv_s = __builtin_amdgcn_exp2f(-ls * kLog2eF);
                float w = __builtin_amdgcn_exp2f((lpk - mx) * kLog2eF);
                if (PR && !own) w = 0.f;
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
                for (int i = 0; i < KK; ++i) one(lp[i], mu[i], lsr[i], kidx(i), kown(i));
            } else {
                for (int k = sub; k < K; k += G) one(my[2 + k], my[2 + K + k], my[2 + 2 * K + k], k, true);
            }
            if (G > 1) {
                se = qsum<G>(se); cdf = qsum<G>(cdf); ccdf = qsum<G>(ccdf); pdf = qsum<G>(pdf);
            }
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
                contrib = log_s + (-lu - l1u) + lpdf + regf * (float)a.reg_factor;
                reg = (double)regf;
            } else {
                // rare: a tail or an underflow.  The reference's fp64 arithmetic (u, then 1 - u by subtraction,
                // safe_log clamps; :100-123, :217-233) on the staged row, one mixture at a time — rolled loops
                // keep this branch's registers below the fast path's.
#ifdef CNF_MIXFWD_NOTAIL
                of = 0.f;       // A/B build: what the fast path alone needs in registers (66 VGPRs at K = 8 against 112)
#else
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
                wide_tail(sed, cdfd, pdfd);
#endif
            }
            }
            if (a.pad_output) of = of * pv;
            if (sub == 0 && a.use_reg && a.reg_out && reg != 0.0) atomicAdd(&a.reg_out[row0 + rl], (float)reg);
        }
        // ---- outputs of the item lanes
        const bool owner = valid && sub == 0;       // one lane per element stores and accounts
#if defined(CNF_MIXFWD_ABLATE) && CNF_MIXFWD_ABLATE >= 2
        if (of == 12345.678f)
#endif
        if (owner) {
            if (ED > 0) ep[tli * ED + d] = of;
            else zstore(&zo_tile[(size_t)tokl * a.D + d], of);
            bad |= isnan(of);
        }
        double cd = use64 ? contrib64 : (double)contrib;
        if (!owner || !active) cd = 0.0;
        badl |= isnan(cd);
        double nlp = 0.0;
        if (NLL && owner) nlp = (double)(-prior_logp(of, prior) * (pad_tile ? pv : 1.f));
        if (gm.split) {
            acc_ldj += cd;
            if (NLL) acc_nlp += nlp;
        } else if (owner) {
            if (active) fix_pair_add(reinterpret_cast<unsigned long long*>(rowacc + rl * 2), rowbig + rl * 2, cd, kRowTermMax);
            if (NLL) fix_pair_add(reinterpret_cast<unsigned long long*>(rowacc + rl * 2 + 1), rowbig + rl * 2 + 1, nlp, kRowTermMax);
        }
        // ---- the pass's channels that are not transformed: copied through (times the padding mask)
#if defined(CNF_MIXFWD_ABLATE) && CNF_MIXFWD_ABLATE >= 2
        if (of == 12345.678f)
#endif
        if (gm.ncopy > 0) {
            const int ne = npt * gm.ncopy;
            for (int e = lane; e < ne; e += kWave) {
                const int tk = (int)fdiv((uint32_t)e, gm.div_nc);
                const int jj = e - tk * gm.ncopy;
                const int c = jj < gm.d0 ? jj : jj + gm.DA;
                const int tl2 = tp + tk;
                const float zv = zload((size_t)tl2 * a.D + c);
                const float pv2 = pad_tile ? padload(tl2) : 1.f;
                const float o = a.pad_output ? zv * pv2 : zv;
                if (ED > 0) ep[tk * ED + c] = o;
                else zstore(&zo_tile[(size_t)tl2 * a.D + c], o);
                if (NLL) {
                    const double lp2 = (double)(-prior_logp(o, prior) * (pad_tile ? pv2 : 1.f));
                    if (gm.split) {
                        acc_nlp += lp2;
                    } else {
                        const int rl2 = (int)fdiv((uint32_t)tl2, gm.div_n);
                        fix_pair_add(reinterpret_cast<unsigned long long*>(rowacc + rl2 * 2 + 1), rowbig + rl2 * 2 + 1, lp2, kRowTermMax);
                    }
                }
            }
        }
        if (ED > 0) {
            // ActNorm -> 1x1 convolution on the pass's tokens, one lane per token (activation_normalization.py:24-48,
            // permutation_layers.py:106-136), then a coalesced store of the strip
            wave_lds_sync();
            if (lane < npt) {
                float xv[ED > 0 ? ED : 1];
                const float p = pad_tile ? padload(tp + lane) : 1.f;
#pragma unroll
                for (int i = 0; i < ED; ++i) {
                    float y = (ep[lane * ED + i] + etab[i]) * etab[ED + i];
                    if (pad_tile) y = y * p;
                    xv[i] = y;
                }
#pragma unroll
                for (int jo = 0; jo < ED; ++jo) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < ED; ++i) acc = fmaf(xv[i], etab[2 * ED + i * ED + jo], acc);
                    if (pad_tile) acc = acc * p;
                    bad |= isnan(acc);
                    ep[lane * ED + jo] = acc;
                }
            }
            wave_lds_sync();
            for (int e = lane; e < npt * ED; e += kWave) zo_tile[(size_t)tp * ED + e] = ep[e];
        }
    };

    // one stage per wave: DMA (the compiler waits for it before the first LDS read), compute, next pass; the other waves of
    // the CU cover the latency.  (A second stage per wave with the next pass's DMA in flight during the compute — inline
    // assembly DMA, explicit vmcnt waits — was built and measured: slower at equal LDS, 16.7 vs 15.3 us at K = 4, because the
    // doubled stages halve the resident waves; profiles/HISTORY.md section 4.)
    for (int tp = 0; tp < ntok; tp += gm.TPP) {                               // wave-uniform
        const int npt = min(gm.TPP, ntok - tp);
        wave_lds_sync();
        pass_body(tp, npt, stage_b, my_pos, [&](size_t i) { return z_tile[i]; }, [&](int t) { return pad_tile[t]; });
        wave_lds_sync();      // the stage is overwritten by the next pass
        if (tp + gm.TPP < ntok)
            my_pos = stage(span0 + (size_t)(tp + gm.TPP) * gm.tokstride, min(gm.TPP, ntok - tp - gm.TPP));
    }

    // ---- per-sample results
    auto finish = [&](int row, double ldj_sum, double nlp_sum, float base, float len) {
        float v = base + (float)(REVERSE ? -ldj_sum : ldj_sum);
        if (ED > 0) {
            // log-det of the two layers, same association as run in sequence: ActNorm uses length | sum(pad) | N,
            // the convolution length | N
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
            v = (v + etab[2 * ED + ED * ED] * len_a) + a.e_sldj[0] * len_c;
        }
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        if (NLL) {
            // task.py:96-118: nll = (-ldj + neglog) / length
            const float neglog = (float)nlp_sum;
            if (a.neglog_out) a.neglog_out[row] = neglog;
            const float nll = (-v) / len + neglog / len;
            a.nll_out[row] = nll;
            if (a.nll_acc) nll_acc_add(a.nll_acc, row & 63, nll);
        }
    };
    if (!gm.split) {
        wave_lds_sync();
        if (own_row)
            finish(row0 + lane, fix_pair_value(rowacc[lane * 2], rowbig[lane * 2]), fix_pair_value(rowacc[lane * 2 + 1], rowbig[lane * 2 + 1]),
                   my_ldj, my_len);
    } else {
        acc_ldj = wave_sum(acc_ldj);
        if (NLL) acc_nlp = wave_sum(acc_nlp);
        if (lane == 0) {
            wpart[wave * 2] = acc_ldj;
            wpart[wave * 2 + 1] = acc_nlp;
        }
        __syncthreads();
        if (threadIdx.x == 0) {
            double t0 = 0.0, t1 = 0.0;
            for (int w = 0; w < kWavesPerBlock; ++w) {
                t0 += wpart[w * 2];
                t1 += wpart[w * 2 + 1];
            }
            if (gm.S == 1) {
                finish(row0, t0, t1, a.ldj_in ? a.ldj_in[row0] : 0.f, (NLL && a.length) ? a.length[row0] : (float)a.N);
            } else {
                // fixed-point row words in the workspace: device-scope integer atomics on both sides.  The adds
                // return (and are waited for) before the ticket is drawn, so the last arriver sees every term.  A partial
                // the fixed-point word cannot take (|partial| >= 2^16, +-inf, NaN) goes to the row's fp64 escape word
                // (returning atomic, waited for like the others).
                unsigned long long* wa = reinterpret_cast<unsigned long long*>(a.ws_acc) + (size_t)row0 * 2;
                double* wb = reinterpret_cast<double*>(a.ws_big) + (size_t)row0 * 2;
                unsigned long long o0 = 0ull, o1 = 0ull;
                if (fabs(t0) < kRowPartMax) o0 = atomicAdd(wa, (unsigned long long)to_fix(t0));
                else o0 = (unsigned long long)__double_as_longlong(unsafeAtomicAdd(wb, t0));
                if (NLL) {
                    if (fabs(t1) < kRowPartMax) o1 = atomicAdd(wa + 1, (unsigned long long)to_fix(t1));
                    else o1 = (unsigned long long)__double_as_longlong(unsafeAtomicAdd(wb + 1, t1));
                }
                asm volatile("s_waitcnt vmcnt(0)" : "+v"(o0), "+v"(o1) : : "memory");
                const int ticket = atomicAdd(a.ws_cnt + row0, 1);
                if (ticket == gm.S - 1) {
                    const long long s0 = (long long)atomicExch(wa, 0ull);
                    const long long s1 = NLL ? (long long)atomicExch(wa + 1, 0ull) : 0ll;
                    const double b0 = __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long*>(wb), 0ull));
                    const double b1 = NLL ? __longlong_as_double((long long)atomicExch(reinterpret_cast<unsigned long long*>(wb + 1), 0ull)) : 0.0;
                    atomicExch(a.ws_cnt + row0, 0);
                    finish(row0, fix_pair_value(s0, b0), fix_pair_value(s1, b1),
                           a.ldj_in ? a.ldj_in[row0] : 0.f, (NLL && a.length) ? a.length[row0] : (float)a.N);
                }
            }
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    if (badl) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    if (range) raise_flag(a.flags, CNF_FLAG_RANGE);
}

// ---- host side: geometry ------------------------------------------------------------------------------------
static std::atomic<int> g_split_waves{4096};      // waves a split launch aims at (cnf_set_mixture_split)
void set_mixture_split_waves(int w) { g_split_waves = w; }
static std::atomic<int> g_mix_nt{64};            // MB of staged parameters above which the DMA loads are nontemporal (0 = never)
void set_mixture_nt_mb(int mb) { g_mix_nt = mb < 0 ? 64 : mb; }
int mixture_nt_mb() { return g_mix_nt.load(); }
static std::atomic<int> g_whole_tokens{1};        // cnf_set_mixture_whole_tokens: A/B switch of the whole-token staging
void set_mixture_whole_tokens(int on) { g_whole_tokens = on ? 1 : 0; }
static bool whole_tokens_enabled() { return g_whole_tokens != 0; }

// Returns false when the shape is outside what this kernel is built for (the caller falls back to the fp64 kernel).
// slot_g > 0: the caller fixes the lanes per item (predicated-slot kernels); otherwise kt > 0 means one lane per item
// and kt == 0 picks G for the rolled-loop kernel from the stage size (or takes force_g).
//
// whole_tokens: the caller can stage ALL D parameter blocks of a token (forward / inverse; the backward writes its
// stage back and cannot).  Taken when skipping the untransformed blocks skips no 128-byte lines anyway — a span of
// DA * P * 4 bytes at a D * P * 4 stride touches ~(span + 128) / stride of the lines — because then a pass is ONE
// contiguous, fully coalesced range instead of 16-byte-aligned pieces of two lines each (Zinc edges / graph colouring
// tiny, D = 2, K = 8: 104-byte spans at a 208-byte stride; profiles/r03_sweep_mixture.txt).
//
// max_wgs > 0: workgroups of this kernel the device holds at once (launch_mixture_tok asks the runtime).  A split
// launch never exceeds it: its workgroups are equal, so a launch of 1.3 rounds takes two (PTB shape, 43 KB of LDS per
// workgroup -> 768 resident: 1024 workgroups 23.7 us, 512 workgroups 20.6 us; Zinc edges 1024 vs 512: 22.3 vs 16.5 us).
bool make_tok_geom(const MixArgs& a, int kt, int force_g, TokGeom& gm, int& G, size_t& lds, int slot_g, bool whole_tokens,
                   long max_wgs) {
    const int P = a.P;
    // transformed channels must be one contiguous range
    int d0 = 0, DA = a.D;
    if (!a.per_item_mask) {
        if (a.act_bits == 0) return false;
        d0 = __builtin_ctzll(a.act_bits);
        DA = __builtin_popcountll(a.act_bits);
        const unsigned long long run = (DA >= 64 ? ~0ull : ((1ull << DA) - 1ull)) << d0;
        if (run != a.act_bits) return false;
    }
    if ((reinterpret_cast<uintptr_t>(a.nn) & 15) != 0) return false;
    // compact layout: nn holds the DA transformed channels' blocks only — every pass is one contiguous span
    const bool compact = a.nn_D != a.D;
    if (compact && (a.per_item_mask || a.nn_D != DA || a.nn_c0 != d0)) return false;
    if (((size_t)a.B * a.N * a.nn_D * P) % 4 != 0) return false;
    const int span = DA * P * 4;
    const int tokstride = a.nn_D * P * 4;
    const bool whole = !compact && whole_tokens && DA < a.D && (span + 128) * 20 >= tokstride * 19 && whole_tokens_enabled();
    const bool contig = DA == a.nn_D || whole;
    const bool phase0 = tokstride % 16 == 0 && (d0 * P * 4) % 16 == 0;       // every span starts on a 16-byte boundary
    const int slot = contig ? tokstride : ((span + (phase0 ? 0 : 12) + 15) & ~15);
    if (slot >= 32768) return false;
    auto stage_of = [&](int g, int& tpp) {
        tpp = kWave / (DA * g);
        if (tpp < 1) return -1;
        const int bytes = contig ? tpp * tokstride + 12 : tpp * slot;
        return ((bytes + 1023) >> 10) << 10;
    };
    G = slot_g > 0 ? slot_g : 1;
    int tpp = 0, stage = stage_of(G, tpp);
    if (kt == 0 && slot_g <= 0) {
        const int cands[3] = {1, 2, 4};
        for (int c = 0; c < 3; ++c) {
            int tp2 = 0;
            const int st = stage_of(cands[c], tp2);
            if (st < 0) break;
            G = cands[c]; tpp = tp2; stage = st;
            if (force_g ? cands[c] == force_g : st <= 12 * 1024) break;
        }
    }
    if (stage < 0 || stage > kMaxDmaInstr * 1024) return false;
    gm.sd0 = whole ? 0 : d0;
    gm.d0 = d0; gm.DA = DA; gm.lpt = DA * G; gm.TPP = tpp; gm.ncopy = a.D - DA; gm.contig = contig ? 1 : 0;
    gm.tokstride = tokstride; gm.slot = slot; gm.stage_bytes = stage;
    gm.div_slot = make_fastdiv((uint32_t)slot);
    gm.div_n = make_fastdiv((uint32_t)a.N);
    gm.div_lpt = make_fastdiv((uint32_t)gm.lpt);
    gm.div_nc = make_fastdiv((uint32_t)std::max(gm.ncopy, 1));
    if (a.N >= 65536) return false;

    // decomposition
    const int ppr = (a.N + tpp - 1) / tpp;                      // passes per row
    // tokens per wave tile: ~128 items by default; fewer for mid-sized batches, so that a few thousand waves exist
    // (a wave walks its passes one after the other: at 10^4 tokens, one pass per wave beats four)
    int target_tokens = std::max(tpp, mixture_tile_items() / std::max(DA, 1));
    const long total_tokens = (long)a.B * a.N;
    if (total_tokens / target_tokens < 2048) target_tokens = (int)std::max<long>(tpp, total_tokens / 2048);
    int rw = 1;
    {
        double best = -1.0;
        const int rmax = std::max(1, std::min(std::min(kMaxRowSlots, a.B), target_tokens / a.N));
        for (int r = 1; r <= rmax; ++r) {
            const int tk = r * a.N;
            const double eff = (double)tk / (double)(((tk + tpp - 1) / tpp) * tpp);
            if (eff >= best - 1e-9) {
                best = std::max(best, eff);
                rw = r;
            }
        }
    }
    {
        // (round 6) more rows per tile where that fills the passes better and still leaves two rounds of waves: S* (64-token rows,
        // 21 tokens per pass) 1 row = 21 + 21 + 21 + 1 tokens -> 2 rows = 7 passes, 87 % full, 8192 tiles — fp32 forward / inverse
        // 108 / 124 -> 104 / 119 us, at the reference's precision 175 / 254 -> 156 / 231 (3 rows: 5462 tiles, 1.33 rounds, slower;
        // 4 rows: one round, the same as 2; profiles/r06_mixture_tile_rows.txt)
        const auto eff_of = [&](int r) { const int tk = r * a.N; return (double)tk / (double)(((tk + tpp - 1) / tpp) * tpp); };
        double best = eff_of(rw) + 0.05;
        const int rcap = std::min(std::min(kMaxRowSlots, a.B), std::max(1, 65535 / std::max(a.N, 1)));
        for (int r = rw + 1; r <= rcap && ((long)a.B + r - 1) / r >= 8192; ++r) {
            if (eff_of(r) > best) {
                best = eff_of(r);
                rw = r;
            }
        }
    }
    const long tiles0 = ((long)a.B + rw - 1) / rw;
    gm.split = 0; gm.rw = rw; gm.S = 1; gm.ntiles = tiles0;
    gm.ppr = ppr;
    gm.pre = 0;
    gm.nt = 0;
    if (tiles0 < 2048 && ppr >= 2 && (long)ppr * kWavesPerBlock * 64 < 0x7fffffffL) {
        // few long rows: S workgroups x 4 waves per row, each wave a run of whole passes
        int s_hi = (int)std::max<long>(1, (g_split_waves + 4L * a.B - 1) / (4L * a.B));
        s_hi = std::min(s_hi, std::max(1, ppr / kWavesPerBlock));
        if (max_wgs > 0) s_hi = (int)std::max<long>(1, std::min<long>(s_hi, max_wgs / a.B));
        if (s_hi > 1 && !(a.ws_acc && a.ws_cnt)) s_hi = 1;      // no workspace: the four waves of ONE workgroup share a row
        // the fewest workgroups per row that reach the smallest number of passes per wave (a wave walks its passes one
        // after the other; every further workgroup costs its table build and its arrival at the row's ticket)
        int S = 1, best_pp = INT_MAX;
        for (int c = 1; c <= s_hi; ++c) {
            const int pp = (ppr + c * kWavesPerBlock - 1) / (c * kWavesPerBlock);
            if (pp < best_pp) {
                best_pp = pp;
                S = c;
            }
        }
        if ((long)a.B * S * kWavesPerBlock > tiles0) {
            gm.split = 1; gm.S = S; gm.rw = 1;
        }
    }
    if ((long)gm.rw * a.N >= 65536) return false;
    const size_t tabs = (((size_t)(a.D + a.D * a.K) * sizeof(BoundTab)) + 15) & ~(size_t)15;
    gm.tab_off = kWavesPerBlock * stage;
    gm.acc_off = (int)((size_t)gm.tab_off + tabs);
    const size_t accb = gm.split ? (size_t)kWavesPerBlock * 2 * sizeof(double)
                                 : (size_t)kWavesPerBlock * gm.rw * 4 * sizeof(long long);       // fixed-point words + their fp64 escape words
    gm.epi_off = (int)(((size_t)gm.acc_off + accb + 15) & ~(size_t)15);
    lds = (size_t)gm.epi_off;
    if (a.e_w) {
        if (a.reverse || a.nll_out || (a.pad && !a.pad_output)) return false;
        lds += ((size_t)(2 * a.D + a.D * a.D + 4) + (size_t)kWavesPerBlock * tpp * a.D) * sizeof(float);
    }
    return lds <= 65536;
}

// Mixtures per lane (KT) and lanes per item (G) for a run-time K that has no exact instantiation, from the
// instantiated pairs with KT * G >= K — a function of K ONLY, so that a sample's result does not depend on the batch it
// sits in (the order in which the G partial sums of an item are combined follows G).  A launch that fills the chip costs
// ~ KT * G per item (slots past K are executed with weight zero), one that does not costs ~ KT (the length of a lane's
// chain): the smallest capacity wins, and among the pairs within 15 % of it the one with the most lanes per item
// (K = 10: 7 x 2 rather than 13 x 1 — at B = 1024, N = 64, D = 4 that is 11.8 / 13.1 us forward / inverse against
// 15.7 / 23.1).  kt = 0: none (rolled loop).
static void slots_for(int K, int& kt, int& g) {
    static const int pairs[7][2] = {{7, 1}, {16, 1}, {7, 2}, {16, 2}, {7, 4}, {13, 4}, {16, 4}};
    kt = 0; g = 0;
    int smallest = 1 << 30;
    for (const auto& pr : pairs)
        if (pr[0] * pr[1] >= K) smallest = std::min(smallest, pr[0] * pr[1]);
    for (const auto& pr : pairs) {
        const int cap = pr[0] * pr[1];
        if (cap < K || cap * 100 > smallest * 115) continue;
        if (pr[1] > g || (pr[1] == g && cap < kt * g)) {
            kt = pr[0]; g = pr[1];
        }
    }
}

using TokKernel = void (*)(MixArgs, TokGeom);

template <int KT, int G, bool PR>
static TokKernel tok_variant64(const MixArgs& a) {
    if (a.reverse) return mixture_tok_kernel<KT, true, G, false, 0, PR, true>;
    return mixture_tok_kernel<KT, false, G, false, 0, PR, true>;
}

template <int KT, int G, bool PR>
static TokKernel tok_variant(const MixArgs& a, bool nll) {
    if (a.reverse) return mixture_tok_kernel<KT, true, G, false, 0, PR>;
    if (nll) return mixture_tok_kernel<KT, false, G, true, 0, PR>;
    if (a.e_w) {
        if constexpr (G == 1 && !PR) {
            switch (a.D) {
                case 2: return mixture_tok_kernel<KT, false, 1, false, 2>;
                case 3: return mixture_tok_kernel<KT, false, 1, false, 3>;
                case 4: return mixture_tok_kernel<KT, false, 1, false, 4>;
                case 6: return mixture_tok_kernel<KT, false, 1, false, 6>;
                default: return nullptr;
            }
        }
        return nullptr;
    }
    return mixture_tok_kernel<KT, false, G, false, 0, PR>;
}

static TokKernel tok_kernel_for(const MixArgs& a, int kt, int slot_g, int G, bool nll, bool x64) {
    if (x64) {
        if (nll || a.e_w) return nullptr;
        if (slot_g > 0) {
            switch (kt * 8 + slot_g) {
                case 7 * 8 + 1: return tok_variant64<7, 1, true>(a);
                case 16 * 8 + 1: return tok_variant64<16, 1, true>(a);
                case 7 * 8 + 2: return tok_variant64<7, 2, true>(a);
                case 16 * 8 + 2: return tok_variant64<16, 2, true>(a);
                case 7 * 8 + 4: return tok_variant64<7, 4, true>(a);
                case 13 * 8 + 4: return tok_variant64<13, 4, true>(a);
                case 16 * 8 + 4: return tok_variant64<16, 4, true>(a);
                default: return nullptr;
            }
        }
        if (kt == 4) return tok_variant64<4, 1, false>(a);
        if (kt == 8) return tok_variant64<8, 1, false>(a);
        if (kt == 16) return tok_variant64<16, 1, false>(a);
        return nullptr;
    }
    if (slot_g > 0) {
        switch (kt * 8 + slot_g) {
            case 7 * 8 + 1: return tok_variant<7, 1, true>(a, nll);
            case 16 * 8 + 1: return tok_variant<16, 1, true>(a, nll);
            case 7 * 8 + 2: return tok_variant<7, 2, true>(a, nll);
            case 16 * 8 + 2: return tok_variant<16, 2, true>(a, nll);
            case 7 * 8 + 4: return tok_variant<7, 4, true>(a, nll);
            case 13 * 8 + 4: return tok_variant<13, 4, true>(a, nll);
            case 16 * 8 + 4: return tok_variant<16, 4, true>(a, nll);
            default: return nullptr;
        }
    }
    if (kt == 4) return tok_variant<4, 1, false>(a, nll);
    if (kt == 8) return tok_variant<8, 1, false>(a, nll);
    if (kt == 16) return tok_variant<16, 1, false>(a, nll);
    if (G == 1) return tok_variant<0, 1, false>(a, nll);
    if (G == 2) return tok_variant<0, 2, false>(a, nll);
    return tok_variant<0, 4, false>(a, nll);
}

// workgroups of `kern` (with `lds` bytes of dynamic LDS) that are resident on the device at once: registers, LDS and
// the wave slots as the runtime's occupancy calculator sees them, times the number of CUs; cached per (kernel, LDS size)
static long resident_workgroups(TokKernel kern, size_t lds) {
    static std::mutex mu;
    static std::map<std::pair<const void*, size_t>, long> cache;
    static int cus = 0;
    std::lock_guard<std::mutex> lock(mu);
    const auto key = std::make_pair(reinterpret_cast<const void*>(kern), lds);
    const auto it = cache.find(key);
    if (it != cache.end()) return it->second;
    if (cus == 0) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess ||
            hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
            cus = 256;
    }
    int per_cu = 0;
    long n = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), kBlock, lds) == hipSuccess &&
        per_cu > 0)
        n = (long)per_cu * cus;
    (void)hipGetLastError();
    cache[key] = n;
    return n;
}

// forward (optionally with the NLL epilogue) or Newton inverse on the token-pass kernel; false = not handled
bool launch_mixture_tok(MixArgs& a, hipStream_t st, int force_g, bool x64) {
    if (x64 && (force_g != 0 || a.e_w || a.nll_out)) return false;      // fp64 arithmetic: register slots, plain coupling
    int kt = (a.K == 4 || a.K == 8 || a.K == 16) ? a.K : 0;
    // every other K: predicated register slots (force_g > 0, the sweep / test knob, keeps the rolled-loop kernel and
    // its lanes per item; the epilogue kernels exist for one lane per item only)
    int slot_g = 0;
    if (kt == 0 && force_g == 0 && !a.e_w) slots_for(a.K, kt, slot_g);
    TokGeom gm;
    int G = 1;
    size_t lds = 0;
    // first choice first: (register slots | exact K) with whole-token staging where it applies, then the same on spans only
    // (a whole token may not fit the stage where its transformed span does), then the rolled loop likewise
    bool whole = true;
    if (!make_tok_geom(a, kt, force_g, gm, G, lds, slot_g, true, 0) && (whole = false, !make_tok_geom(a, kt, force_g, gm, G, lds, slot_g, false, 0))) {
        if (slot_g == 0 || x64) return false;
        kt = 0; slot_g = 0;
        whole = true;
        if (!make_tok_geom(a, kt, force_g, gm, G, lds, 0, true, 0) && (whole = false, !make_tok_geom(a, kt, force_g, gm, G, lds, 0, false, 0)))
            return false;
    }
    // nontemporal DMA loads for launches that stage more than 64 MB (profiles/r05_mixture_nt_sweep.txt: 2-8 % from 78 MB up, -4 % at 52)
    const size_t staged = (size_t)a.B * a.N * (gm.contig ? gm.tokstride : gm.DA * a.P * 4);
    gm.nt = (g_mix_nt.load() != 0 && staged > ((size_t)g_mix_nt.load() << 20)) ? 1 : 0;
    const bool nll = a.nll_out != nullptr;
    if (a.e_w && (G != 1 || !(a.D == 2 || a.D == 3 || a.D == 4 || a.D == 6))) return false;
    const TokKernel kern = tok_kernel_for(a, kt, slot_g, G, nll, x64);
    if (!kern) return false;
    // fp64 kernels keep the per-lane DMA source offsets of a pass when they repeat from pass to pass (spans, every one at
    // the same 16-byte phase) and a pass is at most kTokPre instructions
    gm.pre = ((x64 || (CNF_TOK_PRE_F32 && !a.e_w)) && !gm.contig && (gm.tokstride & 15) == 0 && gm.stage_bytes <= kTokPre * 1024) ? 1 : 0;
    size_t extra = gm.pre ? (size_t)kWavesPerBlock * kTokPre * kWave * sizeof(uint32_t) : 0;
    if (lds + extra > 65536) { gm.pre = 0; extra = 0; }
    if (gm.split && gm.S > 1) {
        const long cap = resident_workgroups(kern, lds + extra);
        const int keep = gm.pre, keep_nt = gm.nt;
        if (cap > 0 && (long)a.B * gm.S > cap && !make_tok_geom(a, kt, force_g, gm, G, lds, slot_g, whole, cap)) return false;
        gm.nt = keep_nt;
        gm.pre = keep;          // (make_tok_geom clears it) the re-split changes S only: stage and strides stay
    }
    const dim3 block(kBlock);
    const dim3 grid(gm.split ? (unsigned)((long)a.B * gm.S) : (unsigned)((gm.ntiles + kWavesPerBlock - 1) / kWavesPerBlock));
    CNF_LAUNCH(kern, grid, block, lds + extra, st, a, gm);
    return true;
}

}  // namespace cnf
