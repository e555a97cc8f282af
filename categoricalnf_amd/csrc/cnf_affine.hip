// Row-streaming fp32 kernels: affine coupling (fwd / inv + log-det), its static-API split forms,
// ExtActNorm, the sigmoid/logit flow and the prior-log-prob + NLL assembly.
//
// All are HBM-bound (16 B/elem for the coupling: z 4 + interleaved (s,t) 8 + z' 4).  One wave owns
// a tile of whole rows (= samples), walks it in 16-byte chunks with fully coalesced loads, and
// reduces the per-row log-det through its private LDS strip — no atomics, deterministic sums.
#include "cnf_common.h"

#include <algorithm>
#include <type_traits>

namespace cnf {

// per (mask-row, channel) constants staged in LDS: keep = 1-mask, keepf = keep*e^sf, fc = max(e^sf,1)
// exact math: x2 = fc;  fast math: x2 = -2 keepf, x3 = 2 log2(e) / fc, so that
// tanh(s / fc) * keepf = keepf - 2 keepf / (2^{s x3} + 1) is mul, exp2, add, rcp, fma
struct alignas(16) ChanTab {
    float keep, keepf, x2, x3;
};

// FAST = hardware transcendental path: exp via v_exp_f32, tanh(x) = 1 - 2/(e^{2x}+1)
template <bool FAST>
__device__ __forceinline__ float exp_m(float x) { return FAST ? __builtin_amdgcn_exp2f(x * 1.4426950408889634f) : expf(x); }
template <bool FAST>
__device__ __forceinline__ float tanh_m(float x) {
    if (!FAST) return tanhf(x);
    // v_exp_f32 + v_rcp_f32 (1 ulp each); HIP's __fdividef is a full IEEE division (~10 instructions)
    const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);
    return 1.f - 2.f * __builtin_amdgcn_rcpf(e + 1.f);
}
constexpr int kMaxTab = 64;     // (mask period) x D entries, one private copy per wave

// logistic prior log-prob (distributions.py:129-136,154-163):
// softplus(v) + softplus(-v) = |v| + 2 log(1 + e^{-|v|}); one exp and one log instead of two each
// (identical to F.softplus's thresholded form to fp32 rounding: for |v| > 20 the log term is < 5e-9)
__device__ __forceinline__ float logistic_logp(float x, float mu, float sigma, float log_sigma) {
    const float v = fabsf((x - mu) / sigma);
    return -((v + 2.f * __logf(1.f + __expf(-v))) + log_sigma);
}
struct AffineArgs {
    const float* z;
    const float* nn;
    const float* sf;
    const float* mask;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int N, D, L, mr, mc, reverse;
    int P;                 // mask period x D: length of the per-channel constant table
    FastDiv div_d, div_p;
    // NLL epilogue (forward only; affine_coupling_kernel<..., NLL = true>): the prior term of z_out is
    // accumulated while z_out is still in registers, so the NLL assembly never re-reads it
    const float* pad;
    const float* length;
    float* neglog_out;
    float* nll_out;
    long long* acc;        // optional: 64 fixed-point partial sums of nll (see cnf_affine_coupling_nll_acc)
    PriorConst prior;
    // ActNorm + 1x1 convolution epilogue (affine_coupling_kernel<..., ED > 0>, cnf_affine_coupling_actconv): the pair of the NEXT
    // flow step (forward) / of the same step, inverted (reverse) applied to the coupling's output while the tile is in LDS
    const float* e_bias;   // [D]
    const float* e_scales; // [D]
    const float* e_w;      // [D,D]: the forward weight, or the inverse weight in the reverse direction
    const float* e_sldj;   // [1]
    const float* e_pad;    // [B,N] or null (the affine coupling itself ignores padding; the pair does not)
    const float* e_length; // [B] or null
};

typedef float mb_vec4 __attribute__((ext_vector_type(4)));
template <int VEC>
struct VecIO;
// Cache hints of the 16-byte row-streaming I/O.  bit 0: nontemporal loads of the conditioning values (s, t); bit 1:
// nontemporal stores; bit 2: nontemporal loads of the latents z.  Interleaved A/B on one box, bench step (forward + NLL,
// then inverse on its output) / forward-only start-to-start, us:  stores only (2): 36.0 / 19.0;  + z loads (6, the
// default): 34.8 / 18.2;  all loads (3): 38.2 / 19.6;  none (0): 38.0 / 21.5;  loads only (1): 37.2 / 19.7.
// z is read exactly once by every kernel of the path; (s, t) is read again by the inverse that follows the forward in
// the bench (and hits in the memory-side cache: the inverse takes 16.4 us after a forward, 18.0 us in a stream of its own).
#ifndef CNF_NT
#define CNF_NT 6
#endif
typedef float nt_f4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4(const float* p) {
#if (CNF_NT & 1)
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ float4 ld4_z(const float* p) {
#if (CNF_NT & 4)
    const nt_f4 v = __builtin_nontemporal_load(reinterpret_cast<const nt_f4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
#else
    return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4(float* p, float a, float b, float c, float d) {
#if (CNF_NT & 2)
    nt_f4 v = {a, b, c, d};
    __builtin_nontemporal_store(v, reinterpret_cast<nt_f4*>(p));
#else
    *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}
template <>
struct VecIO<4> {
    static __device__ __forceinline__ void load_z(const float* p, float* v) {
        const float4 a = ld4(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    // the coupling kernels' latents (three streams: z, (s, t), z'): nontemporal per CNF_NT bit 2; the two-stream kernels
    // (sigmoid flow: 10.2 -> 11.4 us with it) keep the plain load
    static __device__ __forceinline__ void load_z_stream(const float* p, float* v) {
        const float4 a = ld4_z(p);
        v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
    }
    static __device__ __forceinline__ void load_st(const float* p, float* s, float* t) {
        const float4 a = ld4(p);
        const float4 b = ld4(p + 4);
        s[0] = a.x; t[0] = a.y; s[1] = a.z; t[1] = a.w;
        s[2] = b.x; t[2] = b.y; s[3] = b.z; t[3] = b.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) { st4(p, v[0], v[1], v[2], v[3]); }
};
template <>
struct VecIO<2> {
    static __device__ __forceinline__ void load_z(const float* p, float* v) {
        const float2 a = *reinterpret_cast<const float2*>(p);
        v[0] = a.x; v[1] = a.y;
    }
    static __device__ __forceinline__ void load_z_stream(const float* p, float* v) { load_z(p, v); }
    static __device__ __forceinline__ void load_st(const float* p, float* s, float* t) {
        const float4 a = *reinterpret_cast<const float4*>(p);
        s[0] = a.x; t[0] = a.y; s[1] = a.z; t[1] = a.w;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) {
        *reinterpret_cast<float2*>(p) = make_float2(v[0], v[1]);
    }
};
template <>
struct VecIO<1> {
    static __device__ __forceinline__ void load_z(const float* p, float* v) { v[0] = *p; }
    static __device__ __forceinline__ void load_z_stream(const float* p, float* v) { load_z(p, v); }
    static __device__ __forceinline__ void load_st(const float* p, float* s, float* t) {
        const float2 a = *reinterpret_cast<const float2*>(p);
        s[0] = a.x; t[0] = a.y;
    }
    static __device__ __forceinline__ void store(float* p, const float* v) { *p = v[0]; }
};

// coupling_layer.py:53-60 + :88-98 fused.  HAS_SF: tanh bound with the learned scaling factor.
template <int VEC>
struct AffineChunk {
    float zv[VEC], sr[VEC], tr[VEC];
};

// NLLM: 0 = coupling only, 1 = + NLL epilogue, 2 = + NLL epilogue with a padding mask on the prior term
// ED > 0 (= D; whole rows per wave tile, VEC = 4): the coupling's output is not stored but kept in the wave's LDS strip; when the
// tile is through, its tokens go through ActNorm and the 1x1 convolution there (activation_normalization.py:24-48,
// permutation_layers.py:106-136: the arithmetic of actnorm_invconv_kernel, in its order) and leave as one contiguous, coalesced
// store.  The coupling itself — arithmetic, tile walk, order of the row sums — is this kernel's own: z and the log-det are the
// bits of cnf_affine_coupling followed by cnf_actnorm_invconv, and the [B,N,D] round trip between them (8 B/elem) is gone.
template <int VEC, int U, bool HAS_SF, bool REVERSE, bool FAST, int NLLM = 0, int ED = 0>
__global__ __launch_bounds__(kBlock) void affine_coupling_kernel(AffineArgs a, RowTiling tl) {
    constexpr bool NLL = NLLM != 0;
    static_assert(ED == 0 || (VEC == 4 && NLLM == 0), "the ActNorm + convolution epilogue: float4 chunks, no NLL assembly");
    using Acc = typename std::conditional<NLL, Sum2, float>::type;

    // per-wave strip of row partials, sized by the host to the tile (rw * cpr entries; unused when rw == 1)
    extern __shared__ __attribute__((aligned(16))) char part_raw[];
    Acc* part = reinterpret_cast<Acc*>(part_raw) + (size_t)(threadIdx.x >> 6) * (tl.rw * tl.cpr);
    // ED: behind the partials of the four waves, [bias | e^{+-scales} | W] and one strip of rw * L outputs per wave
    const size_t part_bytes = ((tl.rw == 1 ? 0 : (size_t)kWavesPerBlock * tl.rw * tl.cpr * sizeof(Acc)) + 15) & ~(size_t)15;
    float* etab = reinterpret_cast<float*>(part_raw + part_bytes);
    float* zs = etab + ((2 * ED + ED * ED + 3) & ~3) + (size_t)(threadIdx.x >> 6) * ((size_t)tl.rw * tl.L);
    const int e_row0 = (int)(((long)walker_block() * kWavesPerBlock + (threadIdx.x >> 6)) * tl.rw);
    // (the pair's constants through LDS: read as uniform loads into scalar registers at the start of the kernel, the way
    // actnorm_invconv_kernel keeps them, the fused forward took 23.1 instead of 20.6 us at S*)
    if (ED > 0) {
        for (int i = threadIdx.x; i < ED; i += kBlock) {
            etab[i] = a.e_bias[i];
            etab[ED + i] = expf(REVERSE ? -a.e_scales[i] : a.e_scales[i]);
        }
        for (int i = threadIdx.x; i < ED * ED; i += kBlock) etab[2 * ED + i] = a.e_w[i];
        __syncthreads();
    }
    __shared__ ChanTab tab_all[kWavesPerBlock][kMaxTab];
    ChanTab* tab = tab_all[threadIdx.x >> 6];
    // per-wave constant table: its two tiny loads are issued FIRST, the table itself is finished
    // after the first chunk loads are in flight (vmcnt is in-order, so waiting for these does not
    // wait for the chunk loads issued behind them) — see walk_row_tile_split.
    // The table holds the period P = (mask rows) x D once plus its first VEC - 1 entries again, so that the
    // VEC consecutive elements of a chunk read entries i0 .. i0 + VEC - 1 with no wrap-around logic.
    const int ntab = a.P + VEC - 1;
    const int ti = threadIdx.x & 63;
    float m_raw = 0.f, sf_raw = 0.f;
    if (ti < ntab) {
        const int tp = ti % a.P;
        const int r = tp / a.D, d = tp - r * a.D;
        if (a.mask) m_raw = a.mask[r * a.mc + (a.mc == 1 ? 0 : d)];
        if (HAS_SF) sf_raw = a.sf[d];
    }
    auto pre = [&]() {
        // opaque to the optimiser until here: keeps the wait for these two values (and the exp)
        // behind the chunk loads issued in between
        asm volatile("" : "+v"(sf_raw), "+v"(m_raw) : : "memory");
        if (ti < ntab) {
            const float f = HAS_SF ? expf(sf_raw) : 1.f;
            const float fc = fmaxf(f, 1.f);
            ChanTab t;
            t.keep = 1.f - m_raw;
            t.keepf = (1.f - m_raw) * f;
            t.x2 = FAST ? -2.f * t.keepf : fc;
            t.x3 = 2.8853900817779268f / fc;
            tab[ti] = t;
        }
        wave_lds_sync();
    };

    // row scalars of the row this lane will finish, fetched now (see first_finish_row)
    const int myrow = first_finish_row(tl);
    float my_ldj = 0.f, my_len = (float)a.N;
    if (myrow >= 0) {
        if (a.ldj_in) my_ldj = a.ldj_in[myrow];
        if (NLL && a.length) my_len = a.length[myrow];
    }

    // Batch sum (cnf_affine_coupling_nll_acc): the rows of a workgroup meet in ONE LDS word (integer adds of 31.32 fixed
    // point: order-free), every wave draws an LDS ticket when it is through, and the wave that draws the last one sends the
    // workgroup's total to the global accumulator with a single device-scope atomic.  Rounds 1-2 sent one global atomic per
    // ROW: a wave does not retire before its atomic is acknowledged (~2 us away), so every wave slot stayed occupied that
    // much longer — 18.3-18.5 us per launch against 17.1 us without the sum.  The only barrier is here, at the start, where
    // all waves are starting anyway (the per-workgroup reduction of round 1, 19.4 us, had two at the END).
    __shared__ unsigned long long wg_sum;
    __shared__ unsigned int wg_tickets;
    const bool wg_acc = NLL && a.acc != nullptr && !tl.bpr;
    if (wg_acc) {
        if (threadIdx.x == 0) {
            wg_sum = 0ull;
            wg_tickets = 0u;
        }
        __syncthreads();
    }

    bool bad = false;
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        AffineChunk<VEC> c;
        VecIO<VEC>::load_z_stream(a.z + off, c.zv);
        VecIO<VEC>::load_st(a.nn + 2 * off, c.sr, c.tr);
        return c;
    };
    auto proc = [&](const AffineChunk<VEC>& c, int row, int e0) -> Acc {
        const size_t off = (size_t)row * a.L + e0;
        float out[VEC];
        const ChanTab* tb0 = tab + (e0 - (int)fdiv((uint32_t)e0, a.div_p) * a.P);   // e0 mod P
        float acc = 0.f, lp_acc = 0.f, lp_prod = 1.f;
        const float* padrow = NLLM == 2 ? a.pad + (size_t)row * a.N : nullptr;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const ChanTab tb = tb0[j];
            float s;
            if (!HAS_SF) s = c.sr[j] * tb.keep;
            else if (FAST) s = fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(c.sr[j] * tb.x3) + 1.f), tb.x2, tb.keepf);
            else s = tanhf(c.sr[j] / tb.x2) * tb.keepf;
            const float t = c.tr[j] * tb.keep;
            out[j] = REVERSE ? c.zv[j] * exp_m<FAST>(-1.f * s) - t : (c.zv[j] + t) * exp_m<FAST>(s);
            bad |= isnan(out[j]);
            acc += s;
            if (NLLM == 2) lp_acc += prior_logp(out[j], a.prior) * padrow[fdiv((uint32_t)(e0 + j), a.div_d)];
            if (NLLM == 1) {
                // sum_j log(1 + e_j) = log prod_j (1 + e_j): one v_log_f32 per chunk (the product is in [1, 2^VEC])
                const float ax = fabsf(out[j]);
                lp_acc = fmaf(ax, -a.prior.inv_sigma, lp_acc);
                lp_prod *= 1.f + __builtin_amdgcn_exp2f(-ax * a.prior.inv_sigma_log2e);
            }
        }
        if constexpr (ED > 0) {
            *reinterpret_cast<float4*>(zs + (size_t)(row - e_row0) * a.L + e0) = make_float4(out[0], out[1], out[2], out[3]);
        } else {
            VecIO<VEC>::store(a.z_out + off, out);
        }
        if constexpr (NLL) {
            // un-padded rows: the log term and the constant log(sigma) of the chunk's elements are added once
            if (NLLM == 1)
                lp_acc = fmaf(__builtin_amdgcn_logf(lp_prod), -1.3862943611198906f, lp_acc) - (float)VEC * a.prior.log_sigma;
            return Sum2(acc, lp_acc);
        } else {
            return acc;
        }
    };
    auto ldj_of = [&](int row, float sum, float base) {
        float v = REVERSE ? base - sum : base + sum;
        if constexpr (ED > 0) {
            // the pair's log-det on top, same association as the two kernels run in sequence: ActNorm uses length | sum(pad) | N,
            // the convolution length | N
            float len_a, len_c;
            if (a.e_length) {
                len_a = len_c = a.e_length[row];
            } else {
                len_c = (float)a.N;
                len_a = (float)a.N;
                if (a.e_pad) {
                    len_a = 0.f;
                    for (int n = 0; n < a.N; ++n) len_a += a.e_pad[(size_t)row * a.N + n];
                }
            }
            float ssum = 0.f;
#pragma unroll
            for (int i = 0; i < ED; ++i) ssum += a.e_scales[i];
            const float sl = a.e_sldj[0];
            v = REVERSE ? (v - sl * len_c) + (-ssum) * len_a : (v + ssum * len_a) + sl * len_c;
        }
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
        return v;
    };
    auto emit = [&](int row, const Acc& sum, float base, float len) {
        if constexpr (NLL) {
            // task.py:96-118: nll = (-ldj + neglog) / length
            const float ldj = ldj_of(row, sum.a, base);
            const float neglog = -sum.b;
            if (a.neglog_out) a.neglog_out[row] = neglog;
            const float nll = (-ldj) / len + neglog / len;
            a.nll_out[row] = nll;
            // batch sum without a second kernel: signed 31.32 fixed point and integer atomics (associative, so the
            // result does not depend on the order the rows arrive in) into one of 64 words that sit in 64 different
            // cache lines: 256 atomics per line for B = 16384, hidden behind the streaming (64 ADJACENT words, i.e.
            // four lines, serialised them: 55 us per launch)
            // A per-sample NLL the fixed-point word cannot take (|nll| >= 4096, +-inf, NaN) goes to the slot's fp64 escape
            // word with one floating-point atomic (cnf_common.h: fix_pair_add) — the sum then follows the reference's
            // floating-point mean instead of wrapping.
            if (a.acc) {
                if (wg_acc && fabsf(nll) < (float)kAccTermMax)
                    __hip_atomic_fetch_add(&wg_sum, (unsigned long long)to_fix((double)nll), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);   // LDS
                else nll_acc_add(a.acc, row & 63, nll);
            }
        } else {
            ldj_of(row, sum, base);
        }
    };
    auto finish = [&](int row, const Acc& sum) {
        // two separate bodies: the usual one has no load in it, so nothing waits on the z' stores still in flight
        if (row == myrow) emit(row, sum, my_ldj, my_len);
        else emit(row, sum, a.ldj_in ? a.ldj_in[row] : 0.f, (NLL && a.length) ? a.length[row] : (float)a.N);
    };
    walk_row_tile_split<(U == 0 ? 1 : U), Acc, AffineChunk<VEC>, U == 0>(tl, part, load, proc, finish, pre);
    if constexpr (ED > 0) {
        // ---- the tile's tokens through ActNorm and the 1x1 convolution, one token per lane and trip, in place in the strip
        const int nrows = min(tl.rw, tl.B - e_row0);
        if (nrows > 0) {
            wave_lds_sync();
            const int lane = threadIdx.x & 63;
            const int ntok = nrows * a.N;
            for (int t = lane; t < ntok; t += kWave) {
                const float p = a.e_pad ? a.e_pad[(size_t)e_row0 * a.N + t] : 1.f;
                float xv[ED > 0 ? ED : 1], ov[ED > 0 ? ED : 1];
                float* tok = zs + (size_t)t * ED;
#pragma unroll
                for (int i = 0; i < ED; ++i) xv[i] = tok[i];
                if (!REVERSE) {
#pragma unroll
                    for (int i = 0; i < ED; ++i) {
                        float y = (xv[i] + etab[i]) * etab[ED + i];
                        if (a.e_pad) y = y * p;
                        xv[i] = y;
                    }
                }
#pragma unroll
                for (int j = 0; j < ED; ++j) {
                    float acc = 0.f;
#pragma unroll
                    for (int i = 0; i < ED; ++i) acc = fmaf(xv[i], etab[2 * ED + i * ED + j], acc);
                    if (a.e_pad) acc = acc * p;
                    if (REVERSE) {
                        acc = acc * etab[ED + j] - etab[j];
                        if (a.e_pad) acc = acc * p;
                    }
                    bad |= isnan(acc);
                    ov[j] = acc;
                }
#pragma unroll
                for (int i = 0; i < ED; ++i) tok[i] = ov[i];
            }
            wave_lds_sync();
            float4* dst = reinterpret_cast<float4*>(a.z_out + (size_t)e_row0 * a.L);
            const float4* src = reinterpret_cast<const float4*>(zs);
            const int nv = nrows * a.L / 4;
            for (int i = lane; i < nv; i += kWave) {
                const float4 q = src[i];
                __builtin_nontemporal_store(mb_vec4{q.x, q.y, q.z, q.w}, reinterpret_cast<mb_vec4*>(dst + i));
            }
        }
    }
    if (wg_acc) {
        // every wave comes through here, with or without a tile; its rows' LDS adds precede its ticket in program order
        wave_lds_sync();
        if ((threadIdx.x & 63) == 0 &&
            __hip_atomic_fetch_add(&wg_tickets, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP) == kWavesPerBlock - 1) {
            const unsigned long long total = __hip_atomic_load(&wg_sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (total != 0ull)
                atomicAdd(reinterpret_cast<unsigned long long*>(a.acc) + (size_t)(blockIdx.x & 63) * kAccStride, total);
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

template <int VEC, int U, bool FAST>
static void launch_affine_u(const AffineArgs& a, const RowTiling& tl, bool has_sf, bool reverse,
                            hipStream_t st) {
    const dim3 grid = tiling_grid(tl), block(kBlock);
    const size_t strip = (tl.bpr || tl.rw == 1) ? 0 : (size_t)tl.rw * tl.cpr;      // partials per wave
    if (a.nll_out) {   // forward + NLL epilogue
        const size_t lds = kWavesPerBlock * strip * sizeof(Sum2);
        constexpr int UN = (U == 1) ? 1 : 2;      // the epilogue variants are built for 1 and 2 chunks in flight only (prefetch and 3 measured: no gain)
        if (a.pad) {
            if (has_sf) CNF_LAUNCH((affine_coupling_kernel<VEC, UN, true, false, FAST, 2>), grid, block, lds, st, a, tl);
            else CNF_LAUNCH((affine_coupling_kernel<VEC, UN, false, false, FAST, 2>), grid, block, lds, st, a, tl);
        } else {
            if (has_sf) CNF_LAUNCH((affine_coupling_kernel<VEC, UN, true, false, FAST, 1>), grid, block, lds, st, a, tl);
            else CNF_LAUNCH((affine_coupling_kernel<VEC, UN, false, false, FAST, 1>), grid, block, lds, st, a, tl);
        }
        return;
    }
    if (a.e_w) {
        // ActNorm + 1x1 convolution epilogue (cnf_affine_coupling_actconv): float4 chunks, fast math, 2 chunks in flight (the host has
        // checked the rest); the order of the row sums does not depend on the chunks in flight
        if constexpr (VEC == 4 && FAST && U == 2) {
            const size_t part_b = ((kWavesPerBlock * strip * sizeof(float)) + 15) & ~(size_t)15;
            const size_t lds_e = part_b + (((size_t)(2 * a.D + a.D * a.D + 3) & ~(size_t)3) + (size_t)kWavesPerBlock * tl.rw * tl.L) * sizeof(float);
#define CNF_AFF_ED(ED_)                                                                                                   \
    do {                                                                                                                  \
        if (has_sf) {                                                                                                     \
            if (reverse) CNF_LAUNCH((affine_coupling_kernel<4, 2, true, true, true, 0, ED_>), grid, block, lds_e, st, a, tl);   \
            else CNF_LAUNCH((affine_coupling_kernel<4, 2, true, false, true, 0, ED_>), grid, block, lds_e, st, a, tl);          \
        } else {                                                                                                          \
            if (reverse) CNF_LAUNCH((affine_coupling_kernel<4, 2, false, true, true, 0, ED_>), grid, block, lds_e, st, a, tl);  \
            else CNF_LAUNCH((affine_coupling_kernel<4, 2, false, false, true, 0, ED_>), grid, block, lds_e, st, a, tl);         \
        }                                                                                                                 \
    } while (0)
            switch (a.D) {
                case 2: CNF_AFF_ED(2); break;
                case 3: CNF_AFF_ED(3); break;
                case 4: CNF_AFF_ED(4); break;
                case 6: CNF_AFF_ED(6); break;
                default: CNF_AFF_ED(8); break;
            }
#undef CNF_AFF_ED
        }
        return;
    }
    const size_t lds = kWavesPerBlock * strip * sizeof(float);
    if (has_sf) {
        if (reverse) CNF_LAUNCH((affine_coupling_kernel<VEC, U, true, true, FAST>), grid, block, lds, st, a, tl);
        else CNF_LAUNCH((affine_coupling_kernel<VEC, U, true, false, FAST>), grid, block, lds, st, a, tl);
    } else {
        if (reverse) CNF_LAUNCH((affine_coupling_kernel<VEC, U, false, true, FAST>), grid, block, lds, st, a, tl);
        else CNF_LAUNCH((affine_coupling_kernel<VEC, U, false, false, FAST>), grid, block, lds, st, a, tl);
    }
}

template <int VEC, bool FAST>
static void launch_affine_m(const AffineArgs& a, const RowTiling& tl, bool has_sf, bool reverse,
                            hipStream_t st) {
    // chunks per lane in one tile bound how many loads are worth issuing back to back
    const int want = unroll_target();
    const int per_lane = (int)std::min<long>(((long)tl.rw * tl.cpr + kWave - 1) / kWave, std::max(want, 1));
    if (want == 0) { launch_affine_u<VEC, 0, FAST>(a, tl, has_sf, reverse, st); return; }
    switch (per_lane) {
        case 1: launch_affine_u<VEC, 1, FAST>(a, tl, has_sf, reverse, st); break;
        case 2: launch_affine_u<VEC, 2, FAST>(a, tl, has_sf, reverse, st); break;
        case 3: launch_affine_u<VEC, 3, FAST>(a, tl, has_sf, reverse, st); break;
        default: launch_affine_u<VEC, 4, FAST>(a, tl, has_sf, reverse, st); break;
    }
}

template <int VEC>
static void launch_affine(const AffineArgs& a, const RowTiling& tl, bool has_sf, bool reverse,
                          hipStream_t st) {
    if (math_mode() == 1) launch_affine_m<VEC, true>(a, tl, has_sf, reverse, st);
    else launch_affine_m<VEC, false>(a, tl, has_sf, reverse, st);
}

// ---- static-API split forms (coupling_layer.py:76-86 and :88-98) ---------------------------------
__global__ __launch_bounds__(kBlock) void affine_params_kernel(const float* nn, const float* sf,
                                                               const float* mask, int mr, int mc,
                                                               float* s_out, float* t_out, long total,
                                                               int N, int D) {
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < total; i += (long)gridDim.x * kBlock) {
        const int d = (int)(i % D);
        const int n = (int)((i / D) % N);
        const float keep = 1.f - mask_at(mask, mr, mc, n, d);
        const float2 p = *reinterpret_cast<const float2*>(nn + 2 * i);
        float s = p.x;
        if (sf) {
            const float f = expf(sf[d]);
            s = tanhf(s / fmaxf(f, 1.f)) * f;
        }
        s_out[i] = s * keep;
        t_out[i] = p.y * keep;
    }
}

struct TransformArgs {
    const float* z;
    const float* s;
    const float* t;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int L, reverse;
};
template <int VEC>
__global__ __launch_bounds__(kBlock) void affine_transform_kernel(TransformArgs a, RowTiling tl) {
    __shared__ float part[kWavesPerBlock][kMaxTileChunks];
    bool bad = false;
    auto chunk = [&](int row, int e0) -> float {
        const size_t off = (size_t)row * a.L + e0;
        float zv[VEC], sv[VEC], tv[VEC], out[VEC];
        VecIO<VEC>::load_z(a.z + off, zv);
        VecIO<VEC>::load_z(a.s + off, sv);
        VecIO<VEC>::load_z(a.t + off, tv);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            out[j] = a.reverse ? zv[j] * expf(-1.f * sv[j]) - tv[j] : (zv[j] + tv[j]) * expf(sv[j]);
            bad |= isnan(out[j]);
            acc += sv[j];
        }
        VecIO<VEC>::store(a.z_out + off, out);
        return acc;
    };
    auto finish = [&](int row, float sum) {
        const float base = a.ldj_in ? a.ldj_in[row] : 0.f;
        a.ldj_out[row] = a.reverse ? base - sum : base + sum;
    };
    walk_row_tile<float>(tl, part[threadIdx.x >> 6], chunk, finish);
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// ---- ExtActNorm (activation_normalization.py:116-144) ------------------------------------------
struct ExtArgs {
    const float* z;
    const float* nn;    // [B,N,2D] = [bias | scales_raw] per token
    const float* pad;   // [B,N] or null
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int N, D, L, reverse;
    FastDiv div_d;
};
template <int VEC, bool FAST>
__global__ __launch_bounds__(kBlock) void ext_actnorm_kernel(ExtArgs a, RowTiling tl) {
    __shared__ float part[kWavesPerBlock][kMaxTileChunks];
    bool bad = false;
    auto chunk = [&](int row, int e0) -> float {
        const size_t off = (size_t)row * a.L + e0;
        float zv[VEC], out[VEC];
        VecIO<VEC>::load_z(a.z + off, zv);
        int n = (int)fdiv((uint32_t)e0, a.div_d);
        int d = e0 - n * a.D;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const size_t tok = (size_t)row * a.N + n;
            const float bias = a.nn[tok * 2 * a.D + d];
            const float sc = tanh_m<FAST>(a.nn[tok * 2 * a.D + a.D + d]);
            out[j] = a.reverse ? zv[j] * exp_m<FAST>(-sc) - bias : (zv[j] + bias) * exp_m<FAST>(sc);
            bad |= isnan(out[j]);
            acc += a.pad ? sc * a.pad[tok] : sc;
            if (++d == a.D) {
                d = 0;
                ++n;
            }
        }
        VecIO<VEC>::store(a.z_out + off, out);
        return acc;
    };
    auto finish = [&](int row, float sum) {
        const float base = a.ldj_in ? a.ldj_in[row] : 0.f;
        const float v = a.reverse ? base - sum : base + sum;
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    };
    walk_row_tile<float>(tl, part[threadIdx.x >> 6], chunk, finish);
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// ExtActNorm, grouped-token form: one lane owns TP consecutive tokens so that its TP*D latents and its 2*TP*D
// conditioning values ([bias D | scales D] per token) are whole 16-byte vectors: 3*NV coalescable 16-byte loads
// instead of 2 scalar gathers per element.  A "chunk" of the row walk is one token group; needs N % TP == 0.
typedef float ea_f4 __attribute__((ext_vector_type(4)));
template <int D, bool FAST>
__global__ __launch_bounds__(kBlock) void ext_actnorm_group_kernel(ExtArgs a, RowTiling tl) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    __shared__ float part[kWavesPerBlock][kMaxTileChunks];
    bool bad = false;
    auto chunk = [&](int row, int gidx) -> float {
        const size_t tok0 = (size_t)row * a.N + (size_t)gidx * TP;
        float zv[TP * D], cv[2 * TP * D], out[TP * D];
        const ea_f4* zs = reinterpret_cast<const ea_f4*>(a.z + tok0 * D);
        const ea_f4* cs = reinterpret_cast<const ea_f4*>(a.nn + tok0 * 2 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            // latents: read once -> nontemporal (21.2 -> 20.0 us at the benchmark shape); the conditioning values must NOT
            // be (29.8 us: a lane's consecutive 16-byte pieces share 128-byte lines, which then miss L1 every time)
            const ea_f4 q = __builtin_nontemporal_load(zs + v);
            zv[4 * v] = q.x; zv[4 * v + 1] = q.y; zv[4 * v + 2] = q.z; zv[4 * v + 3] = q.w;
        }
#pragma unroll
        for (int v = 0; v < 2 * NV; ++v) {
            const ea_f4 q = cs[v];
            cv[4 * v] = q.x; cv[4 * v + 1] = q.y; cv[4 * v + 2] = q.z; cv[4 * v + 3] = q.w;
        }
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            const float pv = a.pad ? a.pad[tok0 + k] : 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float bias = cv[k * 2 * D + d];
                const float sc = tanh_m<FAST>(cv[k * 2 * D + D + d]);
                const float o = a.reverse ? zv[k * D + d] * exp_m<FAST>(-sc) - bias : (zv[k * D + d] + bias) * exp_m<FAST>(sc);
                bad |= isnan(o);
                out[k * D + d] = o;
                acc += a.pad ? sc * pv : sc;
            }
        }
        ea_f4* dst = reinterpret_cast<ea_f4*>(a.z_out + tok0 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const ea_f4 q = {out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
            __builtin_nontemporal_store(q, dst + v);
        }
        return acc;
    };
    auto finish = [&](int row, float sum) {
        const float base = a.ldj_in ? a.ldj_in[row] : 0.f;
        const float v = a.reverse ? base - sum : base + sum;
        a.ldj_out[row] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    };
    walk_row_tile<float>(tl, part[threadIdx.x >> 6], chunk, finish);
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// ---- sigmoid / logit flow (sigmoid_layer.py:24-47) ---------------------------------------------
__device__ __forceinline__ float softplus_t20(float x) {   // F.softplus, beta 1, threshold 20
    return x > 20.f ? x : log1pf(expf(x));
}
struct SigArgs {
    const float* z;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int L, reverse;
    float alpha, log1ma;
};
template <int VEC, bool FAST>
__global__ __launch_bounds__(kBlock) void sigmoid_flow_kernel(SigArgs a, RowTiling tl) {
    __shared__ float part[kWavesPerBlock][kMaxTileChunks];
    bool bad = false;
    auto chunk = [&](int row, int e0) -> float {
        const size_t off = (size_t)row * a.L + e0;
        float zv[VEC], out[VEC];
        VecIO<VEC>::load_z(a.z + off, zv);
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (FAST) {
                // hardware exp2 / log2 / rcp: softplus(-z) = max(-z, 0) + log(1 + e^-|z|), sigmoid from the same e
                if (!a.reverse) {
                    const float e = __builtin_amdgcn_exp2f(-fabsf(zv[j]) * 1.4426950408889634f);
                    const float r = __builtin_amdgcn_rcpf(1.f + e);
                    const float sp = fmaxf(-zv[j], 0.f) + 0.6931471805599453f * __builtin_amdgcn_logf(1.f + e);
                    acc += -zv[j] - 2.f * sp;
                    out[j] = zv[j] >= 0.f ? r : e * r;
                } else {
                    const float u = zv[j] * (1.f - a.alpha) + a.alpha * 0.5f;
                    const float lu = 0.6931471805599453f * __builtin_amdgcn_logf(u);
                    const float l1u = 0.6931471805599453f * __builtin_amdgcn_logf(1.f - u);
                    acc += (-lu - l1u + a.log1ma);
                    out[j] = lu - l1u;
                }
            } else if (!a.reverse) {
                acc += -zv[j] - 2.f * softplus_t20(-zv[j]);
                out[j] = 1.f / (1.f + expf(-zv[j]));
            } else {
                const float u = zv[j] * (1.f - a.alpha) + a.alpha * 0.5f;
                const float lu = logf(u), l1u = logf(1.f - u);
                acc += (-lu - l1u + a.log1ma);
                out[j] = lu - l1u;
            }
            bad |= isnan(out[j]);
        }
        VecIO<VEC>::store(a.z_out + off, out);
        return acc;
    };
    auto finish = [&](int row, float sum) {
        a.ldj_out[row] = (a.ldj_in ? a.ldj_in[row] : 0.f) + sum;
    };
    walk_row_tile<float>(tl, part[threadIdx.x >> 6], chunk, finish);
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
}

// ---- logistic prior log-prob + NLL (distributions.py:129-136,154-163; set_modeling/task.py:96-118)
struct NllArgs {
    const float* z;
    const float* pad;
    const float* ldj;
    const float* length;
    float* neglog_out;
    float* nll_out;
    double* sums;
    int N, D, L;
    PriorConst prior;
    FastDiv div_d;
};
template <int VEC>
struct ZChunk {
    float v[VEC];
};
template <int VEC>
__global__ __launch_bounds__(kBlock) void prior_nll_kernel(NllArgs a, RowTiling tl) {
    __shared__ float part[kWavesPerBlock][kMaxTileChunks];
    // row scalars of the row this lane will finish, fetched now (see first_finish_row)
    const int myrow = first_finish_row(tl);
    float my_len = (float)a.N, my_ldj = 0.f;
    if (myrow >= 0) {
        if (a.length) my_len = a.length[myrow];
        if (a.ldj) my_ldj = a.ldj[myrow];
    }
    auto load = [&](int row, int e0) {
        ZChunk<VEC> c;
        VecIO<VEC>::load_z(a.z + (size_t)row * a.L + e0, c.v);
        return c;
    };
    auto proc = [&](const ZChunk<VEC>& c, int row, int e0) -> float {
        int n = (int)fdiv((uint32_t)e0, a.div_d);
        int d = e0 - n * a.D;
        float acc = 0.f;
        if (!a.pad) {
            // sum_j log(1 + e_j) = log prod_j (1 + e_j): one v_log_f32 per chunk
            float prod = 1.f;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                const float ax = fabsf(c.v[j]);
                acc = fmaf(ax, -a.prior.inv_sigma, acc);
                prod *= 1.f + __builtin_amdgcn_exp2f(-ax * a.prior.inv_sigma_log2e);
            }
            return fmaf(__builtin_amdgcn_logf(prod), -1.3862943611198906f, acc) - (float)VEC * a.prior.log_sigma;
        }
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const float lp = prior_logp(c.v[j], a.prior);
            acc += lp * a.pad[(size_t)row * a.N + n];
            if (++d == a.D) {
                d = 0;
                ++n;
            }
        }
        return acc;
    };
    auto emit = [&](int row, float sum, float len, float ldj) {
        const float neglog = -sum;
        const float nll = (-ldj) / len + neglog / len;
        if (a.neglog_out) a.neglog_out[row] = neglog;
        if (a.nll_out) a.nll_out[row] = nll;
    };
    auto finish = [&](int row, float sum) {
        if (row == myrow) emit(row, sum, my_len, my_ldj);
        else emit(row, sum, a.length ? a.length[row] : (float)a.N, a.ldj ? a.ldj[row] : 0.f);
    };
    // read-only stream: all (up to 4) chunks of a lane are loaded back to back
    walk_row_tile_split<4, float, ZChunk<VEC>>(tl, part[threadIdx.x >> 6], load, proc, finish);
}

// sums = (sum_b nll[b], B): one workgroup, fixed order (deterministic), fp64 accumulation.
// 16k waves adding into one address would serialise at ~12 ns per atomic (~200 us); this is ~3 us.
__global__ __launch_bounds__(1024) void nll_sum_kernel(const float* nll, int B, double* sums) {
    __shared__ double sh[16];
    double acc = 0.0;
    // latency-bound (the values were just written: L2 hits): every lane issues all its loads before the first
    // use — four 16-byte loads per pass when the pointer allows, i.e. one round trip for B = 16384
    int done = 0;
    if ((reinterpret_cast<uintptr_t>(nll) & 15) == 0) {
        const float4* p = reinterpret_cast<const float4*>(nll);
        const int n4 = B >> 2;
        for (int i = threadIdx.x; i < n4; i += 4 * 1024) {
            float4 v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) v[u] = p[min(i + u * 1024, n4 - 1)];
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (i + u * 1024 < n4) acc += ((double)v[u].x + (double)v[u].y) + ((double)v[u].z + (double)v[u].w);
        }
        done = n4 << 2;
    }
    for (int i = done + threadIdx.x; i < B; i += 1024) acc += (double)nll[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += sh[w];
        sums[0] = t;
        sums[1] = (double)B;
    }
}

__global__ __launch_bounds__(kBlock) void logistic_log_prob_kernel(const float* x, float* out, long n,
                                                                   float mu, float sigma,
                                                                   float log_sigma, int* flags) {
    bool bad = false;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock) {
        const float lp = logistic_logp(x[i], mu, sigma, log_sigma);
        bad |= isnan(lp);
        out[i] = lp;
    }
    if (bad) raise_flag(flags, CNF_FLAG_NAN_Z);
}

// distributions.py:139-145,117-127.  FAST = false: logit evaluated in fp64 like the reference.  FAST = true: fp32,
// logit(u) = log u - log(1 - u) with the hardware log2; u' lies in [eps/2, 1 - eps/2], 1 - u' is exact for
// u' >= 1/2 (Sterbenz) and rounded to 6e-8 relative below, so both logs keep fp32 relative accuracy
// (|x - x_fp64| <= ~2e-6 at |x| = 16); 16-byte loads and stores when the tensors allow.
template <bool FAST>
__device__ __forceinline__ float logistic_icdf(float u, float mu, float sigma, float eps) {
    const float uf = (u * (1.f - eps)) + eps / 2.f;
    float v;
    if (FAST) v = (__builtin_amdgcn_logf(uf) - __builtin_amdgcn_logf(1.f - uf)) * 0.6931471805599453f;
    else v = (float)(-log(1.0 / (double)uf - 1.0));
    return v * sigma + mu;
}
template <bool FAST>
__global__ __launch_bounds__(kBlock) void logistic_from_uniform_kernel(const float* u, float* x, long n,
                                                                       float mu, float sigma,
                                                                       float eps) {
    const bool vec = ((reinterpret_cast<uintptr_t>(u) | reinterpret_cast<uintptr_t>(x)) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        const float4 a = reinterpret_cast<const float4*>(u)[i];
        float4 o;
        o.x = logistic_icdf<FAST>(a.x, mu, sigma, eps);
        o.y = logistic_icdf<FAST>(a.y, mu, sigma, eps);
        o.z = logistic_icdf<FAST>(a.z, mu, sigma, eps);
        o.w = logistic_icdf<FAST>(a.w, mu, sigma, eps);
        reinterpret_cast<float4*>(x)[i] = o;
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock)
        x[i] = logistic_icdf<FAST>(u[i], mu, sigma, eps);
}

static inline int stream_grid(long n) {
    const long blocks = (n + kBlock - 1) / kBlock;
    return (int)std::min<long>(std::max<long>(blocks, 1), 256 * 8);
}

}  // namespace cnf

using namespace cnf;

#define DISPATCH_VEC(tl, CALL)            \
    switch ((tl).vec) {                   \
        case 4: { constexpr int V = 4; CALL; } break; \
        case 2: { constexpr int V = 2; CALL; } break; \
        default: { constexpr int V = 1; CALL; } break; \
    }

extern "C" {

static int affine_coupling_impl(const char* who, const float* z, const float* nn_out, const float* scaling_factor,
                                const float* mask, int mask_rows, int mask_cols,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                int B, int N, int D, int reverse,
                                const float* pad, const float* length, float* neglog_out, float* nll_out,
                                double* sums, float sigma, float log_sigma, int* flags, cnf_stream_t stream,
                                long long* acc = nullptr, const float* const* epi = nullptr) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out, "%s: null tensor", who);
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "%s: bad shape B=%d N=%d D=%d", who, B, N, D);
    if (B == 0) return CNF_OK;
    if (!mask) { mask_rows = 1; mask_cols = D; }
    CNF_REQUIRE(mask_rows >= 1 && (mask_cols == D || mask_cols == 1),
                "%s: mask must be [rows,%d] or [rows,1], got [%d,%d]", who, D, mask_rows, mask_cols);
    CNF_REQUIRE((long)N * D < 65536, "%s: N*D=%ld exceeds 65535", who, (long)N * D);
    if (mask_rows > N) mask_rows = N;   // coupling_layer.py:72-73 truncation
    if (mask_rows * D + 3 > kMaxTab) { set_error("%s: mask period %d x D %d too large", who, mask_rows, D); return CNF_ERR_UNSUPPORTED; }
    AffineArgs a;
    a.z = z; a.nn = nn_out; a.sf = scaling_factor; a.mask = mask; a.ldj_in = ldj_in;
    a.z_out = z_out; a.ldj_out = ldj_out; a.flags = flags;
    a.N = N; a.D = D; a.L = N * D; a.mr = mask_rows; a.mc = mask_cols; a.reverse = reverse;
    a.div_d = make_fastdiv((uint32_t)D);
    a.P = mask_rows * D;
    a.div_p = make_fastdiv((uint32_t)a.P);
    a.pad = pad; a.length = length; a.neglog_out = neglog_out; a.nll_out = nll_out; a.acc = acc;
    a.prior = make_prior_const(sigma, log_sigma);
    a.e_bias = a.e_scales = a.e_w = a.e_sldj = a.e_pad = a.e_length = nullptr;
    if (epi) {
        a.e_bias = epi[0]; a.e_scales = epi[1]; a.e_w = epi[2]; a.e_sldj = epi[3]; a.e_pad = epi[4]; a.e_length = epi[5];
    }
    // one tiling for the plain and the NLL variant: the per-row summation order depends on it, and the fused kernel's
    // log-det must stay bit-equal to the plain forward's (and exactly minus the inverse's).  A larger tile (384 chunks)
    // would save the NLL variant 0.5 us at the benchmark shape (tools/sweep_nll.py) at the price of that equality.
    const RowTiling tl = make_row_tiling(B, a.L, 0, tile_chunks_target());
    if (epi) {
        // the epilogue kernels: whole rows per wave tile in float4 chunks, fast math, D in {2, 3, 4, 6, 8}, the tile's outputs in LDS
        const size_t strip = tl.rw == 1 ? 0 : (size_t)tl.rw * tl.cpr;
        const size_t lds_e = (((size_t)kWavesPerBlock * strip * sizeof(float) + 15) & ~(size_t)15) +
                             (((size_t)(2 * D + D * D + 3) & ~(size_t)3) + (size_t)kWavesPerBlock * tl.rw * tl.L) * sizeof(float);
        if (!(D == 2 || D == 3 || D == 4 || D == 6 || D == 8) || tl.vec != 4 || tl.bpr || math_mode() != 1 || lds_e > 65536 ||
            (reinterpret_cast<uintptr_t>(z_out) & 15) != 0) {
            set_error("%s: shape / mode outside the fused kernel (B=%d N=%d D=%d): run cnf_affine_coupling and cnf_actnorm_invconv", who, B, N, D);
            return CNF_ERR_UNSUPPORTED;
        }
        launch_affine_u<4, 2, true>(a, tl, scaling_factor != nullptr, reverse != 0, (hipStream_t)stream);
        return launch_status(who);
    }
    DISPATCH_VEC(tl, launch_affine<V>(a, tl, scaling_factor != nullptr, reverse != 0, (hipStream_t)stream));
    if (sums) CNF_LAUNCH(nll_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll_out, B, sums);
    return launch_status(who);
}

int cnf_affine_coupling(const float* z, const float* nn_out, const float* scaling_factor,
                        const float* mask, int mask_rows, int mask_cols,
                        const float* ldj_in, float* z_out, float* ldj_out,
                        int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    return affine_coupling_impl("cnf_affine_coupling", z, nn_out, scaling_factor, mask, mask_rows, mask_cols,
                                ldj_in, z_out, ldj_out, B, N, D, reverse, nullptr, nullptr, nullptr, nullptr,
                                nullptr, 1.f, 0.f, flags, stream);
}

int cnf_affine_coupling_actconv(const float* z, const float* nn_out, const float* scaling_factor,
                                const float* mask, int mask_rows, int mask_cols,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                const float* an_bias, const float* an_scales, const float* conv_weight, const float* conv_sldj,
                                const float* pad, const float* length,
                                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(an_bias && an_scales && conv_weight && conv_sldj, "cnf_affine_coupling_actconv: null tensor");
    const float* epi[6] = {an_bias, an_scales, conv_weight, conv_sldj, pad, length};
    return affine_coupling_impl("cnf_affine_coupling_actconv", z, nn_out, scaling_factor, mask, mask_rows, mask_cols,
                                ldj_in, z_out, ldj_out, B, N, D, reverse, nullptr, nullptr, nullptr, nullptr,
                                nullptr, 1.f, 0.f, flags, stream, nullptr, epi);
}

int cnf_affine_coupling_nll(const float* z, const float* nn_out, const float* scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const float* ldj_in, float* z_out, float* ldj_out,
                            const float* pad, const float* length,
                            float* neglog_out, float* nll_out, double* sums,
                            int B, int N, int D, float sigma, float log_sigma,
                            int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(nll_out, "cnf_affine_coupling_nll: nll_out is required");
    return affine_coupling_impl("cnf_affine_coupling_nll", z, nn_out, scaling_factor, mask, mask_rows, mask_cols,
                                ldj_in, z_out, ldj_out, B, N, D, 0, pad, length, neglog_out, nll_out, sums,
                                sigma, log_sigma, flags, stream);
}

int cnf_affine_coupling_nll_acc(const float* z, const float* nn_out, const float* scaling_factor,
                                const float* mask, int mask_rows, int mask_cols,
                                const float* ldj_in, float* z_out, float* ldj_out,
                                const float* pad, const float* length,
                                float* neglog_out, float* nll_out, int64_t* acc,
                                int B, int N, int D, float sigma, float log_sigma,
                                int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(nll_out && acc, "cnf_affine_coupling_nll_acc: nll_out and acc are required");
    return affine_coupling_impl("cnf_affine_coupling_nll_acc", z, nn_out, scaling_factor, mask, mask_rows, mask_cols,
                                ldj_in, z_out, ldj_out, B, N, D, 0, pad, length, neglog_out, nll_out, nullptr,
                                sigma, log_sigma, flags, stream, reinterpret_cast<long long*>(acc));
}

__global__ __launch_bounds__(1024) void nll_acc_read_kernel(const long long* acc, long n, double count, double* sums) {
    // a few thousand slots (unused padding words are 0): 16 waves, independent loads, fixed summation order
    // (one wave walking 16k words paid the memory latency 256 times over: ~0.2 ms)
    __shared__ double sh[16];
    double t = 0.0;
    // word 16 k + 1 of every 128-byte line is the slot's fp64 escape word (cnf_common.h: nll_acc_add), the others fixed point
    for (long i = threadIdx.x; i < n; i += 1024)
        t += (i & (cnf::kAccStride - 1)) == 1 ? __longlong_as_double(acc[i]) : (double)acc[i] * (1.0 / 4294967296.0);
    t = cnf::wave_sum(t);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < 16; ++w) s += sh[w];
        sums[0] = s;
        sums[1] = count;
    }
}

int cnf_nll_acc_read(const int64_t* acc, int64_t n_slots, double count, double* sums, cnf_stream_t stream) {
    CNF_REQUIRE(acc && sums && n_slots > 0, "cnf_nll_acc_read: bad argument");
    CNF_LAUNCH(nll_acc_read_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream,
                       reinterpret_cast<const long long*>(acc), (long)n_slots, count, sums);
    return launch_status("cnf_nll_acc_read");
}

int cnf_affine_params(const float* nn_out, const float* scaling_factor,
                      const float* mask, int mask_rows, int mask_cols,
                      float* s_out, float* t_out, int B, int N, int D, cnf_stream_t stream) {
    CNF_REQUIRE(nn_out && s_out && t_out, "cnf_affine_params: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_affine_params: bad shape");
    if (B == 0) return CNF_OK;
    if (mask && mask_rows > N) mask_rows = N;
    const long total = (long)B * N * D;
    CNF_LAUNCH(affine_params_kernel, dim3(stream_grid(total)), dim3(kBlock), 0, (hipStream_t)stream,
                       nn_out, scaling_factor, mask, mask_rows, mask_cols, s_out, t_out, total, N, D);
    return launch_status("cnf_affine_params");
}

int cnf_affine_transform(const float* z, const float* s, const float* t,
                         const float* ldj_in, float* z_out, float* ldj_out,
                         int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && s && t && z_out && ldj_out, "cnf_affine_transform: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_affine_transform: bad shape");
    if (B == 0) return CNF_OK;
    TransformArgs a{z, s, t, ldj_in, z_out, ldj_out, flags, N * D, reverse};
    const RowTiling tl = make_row_tiling(B, a.L);
    DISPATCH_VEC(tl, CNF_LAUNCH((affine_transform_kernel<V>), tiling_grid(tl), dim3(kBlock), 0,
                                        (hipStream_t)stream, a, tl));
    return launch_status("cnf_affine_transform");
}

int cnf_ext_actnorm(const float* z, const float* nn_out, const float* pad,
                    const float* ldj_in, float* z_out, float* ldj_out,
                    int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && nn_out && z_out && ldj_out, "cnf_ext_actnorm: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && (long)N * D < 65536, "cnf_ext_actnorm: bad shape");
    if (B == 0) return CNF_OK;
    ExtArgs a{z, nn_out, pad, ldj_in, z_out, ldj_out, flags, N, D, N * D, reverse, make_fastdiv((uint32_t)D)};
    // grouped-token kernel (16-byte I/O) for the common channel counts when the tokens of a row fill whole groups
    const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    const bool aligned = ((reinterpret_cast<uintptr_t>(z) | reinterpret_cast<uintptr_t>(nn_out) | reinterpret_cast<uintptr_t>(z_out)) & 15) == 0;
    if (aligned && N % tp == 0 && N / tp >= 8 && (D == 2 || D == 3 || D == 4 || D == 6 || D == 8)) {
        const RowTiling tg = make_row_tiling(B, N / tp, /*force_vec=*/1);
        const dim3 grid = tiling_grid(tg), block(kBlock);
        hipStream_t st = (hipStream_t)stream;
#define CNF_EXTG(D_)                                                                                              \
    if (math_mode() == 1) CNF_LAUNCH((ext_actnorm_group_kernel<D_, true>), grid, block, 0, st, a, tg);   \
    else CNF_LAUNCH((ext_actnorm_group_kernel<D_, false>), grid, block, 0, st, a, tg)
        switch (D) {
            case 2: CNF_EXTG(2); break;
            case 3: CNF_EXTG(3); break;
            case 4: CNF_EXTG(4); break;
            case 6: CNF_EXTG(6); break;
            default: CNF_EXTG(8); break;
        }
#undef CNF_EXTG
        return launch_status("cnf_ext_actnorm");
    }
    const RowTiling tl = make_row_tiling(B, a.L);
    if (math_mode() == 1) {
        DISPATCH_VEC(tl, CNF_LAUNCH((ext_actnorm_kernel<V, true>), tiling_grid(tl), dim3(kBlock), 0,
                                            (hipStream_t)stream, a, tl));
    } else {
        DISPATCH_VEC(tl, CNF_LAUNCH((ext_actnorm_kernel<V, false>), tiling_grid(tl), dim3(kBlock), 0,
                                            (hipStream_t)stream, a, tl));
    }
    return launch_status("cnf_ext_actnorm");
}

int cnf_sigmoid_flow(const float* z, const float* ldj_in, float* z_out, float* ldj_out,
                     int B, int L, int reverse, float alpha, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && z_out && ldj_out, "cnf_sigmoid_flow: null tensor");
    CNF_REQUIRE(B >= 0 && L > 0, "cnf_sigmoid_flow: bad shape");
    if (B == 0) return CNF_OK;
    SigArgs a{z, ldj_in, z_out, ldj_out, flags, L, reverse, alpha, (float)log(1.0 - (double)alpha)};
    const RowTiling tl = make_row_tiling(B, L);
    if (math_mode() == 1) {
        DISPATCH_VEC(tl, CNF_LAUNCH((sigmoid_flow_kernel<V, true>), tiling_grid(tl), dim3(kBlock), 0,
                                            (hipStream_t)stream, a, tl));
    } else {
        DISPATCH_VEC(tl, CNF_LAUNCH((sigmoid_flow_kernel<V, false>), tiling_grid(tl), dim3(kBlock), 0,
                                            (hipStream_t)stream, a, tl));
    }
    return launch_status("cnf_sigmoid_flow");
}

int cnf_logistic_log_prob(const float* x, float* logp, int64_t n, float mu, float sigma,
                          float log_sigma, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(x && logp && n >= 0, "cnf_logistic_log_prob: bad argument");
    if (n == 0) return CNF_OK;
    CNF_LAUNCH(logistic_log_prob_kernel, dim3(stream_grid(n)), dim3(kBlock), 0, (hipStream_t)stream,
                       x, logp, (long)n, mu, sigma, log_sigma, flags);
    return launch_status("cnf_logistic_log_prob");
}

int cnf_logistic_from_uniform(const float* u, float* x, int64_t n, float mu, float sigma, float eps,
                              cnf_stream_t stream) {
    CNF_REQUIRE(u && x && n >= 0, "cnf_logistic_from_uniform: bad argument");
    if (n == 0) return CNF_OK;
    if (math_mode() == 1)
        CNF_LAUNCH(logistic_from_uniform_kernel<true>, dim3(stream_grid((n + 3) / 4)), dim3(kBlock), 0,
                           (hipStream_t)stream, u, x, (long)n, mu, sigma, eps);
    else
        CNF_LAUNCH(logistic_from_uniform_kernel<false>, dim3(stream_grid(n)), dim3(kBlock), 0,
                           (hipStream_t)stream, u, x, (long)n, mu, sigma, eps);
    return launch_status("cnf_logistic_from_uniform");
}

int cnf_nll_sum(const float* nll, int B, double* sums, cnf_stream_t stream) {
    CNF_REQUIRE(nll && sums && B >= 0, "cnf_nll_sum: bad argument");
    CNF_LAUNCH(nll_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll, B, sums);
    return launch_status("cnf_nll_sum");
}

int cnf_prior_nll(const float* z, const float* pad, const float* ldj, const float* length,
                  float* neglog_out, float* nll_out, double* sums,
                  int B, int N, int D, float sigma, float log_sigma, cnf_stream_t stream) {
    CNF_REQUIRE(z, "cnf_prior_nll: null tensor");
    CNF_REQUIRE(!sums || nll_out, "cnf_prior_nll: `sums` needs `nll_out` (the batch sum is taken over it)");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && (long)N * D < 65536, "cnf_prior_nll: bad shape");
    if (B == 0) return CNF_OK;
    NllArgs a{z, pad, ldj, length, neglog_out, nll_out, sums, N, D, N * D, make_prior_const(sigma, log_sigma),
              make_fastdiv((uint32_t)D)};
    const RowTiling tl = make_row_tiling(B, a.L);
    DISPATCH_VEC(tl, CNF_LAUNCH((prior_nll_kernel<V>), tiling_grid(tl), dim3(kBlock), 0,
                                        (hipStream_t)stream, a, tl));
    if (sums) CNF_LAUNCH(nll_sum_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, nll_out, B, sums);
    return launch_status("cnf_prior_nll");
}

}  // extern "C"
