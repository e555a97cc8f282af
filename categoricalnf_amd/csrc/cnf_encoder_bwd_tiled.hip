// Class-tiled backward of the mixture-model categorical encoder (any vocabulary size): d loss / d table [C, 2D] of
// cnf_encoder_forward / cnf_encoder_forward_tiled (linear_encoding.py:59-106,153-174).  Split from cnf_encoder.hip, whose
// forward / decode loops are built without the SLP vectoriser (Makefile).
#include "cnf_encoder.h"

#include <atomic>
#include <map>
#include <mutex>

namespace cnf {

// ---- backward for large vocabularies --------------------------------------------------------------------------------
//
// d loss / d table [C, 2D] of the forward above for tables beyond the LDS-resident backward kernel (2 C D > 2048).
// The work is T x C x D in both directions, but the two reductions run across different axes, so there are two
// kernels and neither needs a cross-lane reduction or a floating-point atomic:
//   (1) token lanes (encoder_bwd_token_kernel): a lane owns a token and walks the class chunks ONCE, keeping the
//       streamed log-sum-exp together with D accumulators sum_j w_j tanh(x_jd / 2 sigma) A_jd that are rescaled
//       with it whenever the running maximum moves (w_j = 2^(v_j - max)); this gives the token's log-denominator and
//       d loss / d z (the sum over all OTHER classes' reverse flows) in one sweep.  It writes one record per token:
//       [z (D) | own-class gradient w.r.t. bias and tanh(scale) (2D) | log2 denominator | beta G | class].
//   (2) class lanes (encoder_bwd_class_kernel): a lane owns a class (its 2D + 1 constants in registers) and walks a
//       range of token records broadcast from LDS, accumulating its class's 2D gradient entries in registers; the
//       lane whose class IS the token's class adds the record's own-class gradient instead.  Token ranges ("splits")
//       give the launch enough workgroups; their partial tables are summed in split order by a third tiny kernel, so
//       the result is bit-reproducible.
// Notation: G = d loss / d ldj_tok (x pad), v_j the class scores, q = softmax(v).
struct EncBwdTiledArgs {
    const int64_t* categ;
    const float* eps;
    const float* table;
    const float* prior;
    const float* pad;
    const float* g_zout;
    const float* g_ldj;
    float* rec;              // [T, 3D + 3]
    float* partials;         // [S, C, 2D]
    float* g_table;          // [C, 2D]
    long ntok;
    int N, D, C, S;
    float beta, sigma, log_sigma;
};

// Cold path of the token lanes (a token whose density sum left the fp32 range, see class_density in cnf_encoder.hip):
// the streamed base-2 log-sum-exp over ALL classes together with the D accumulators, rescaled with the running maximum,
// straight from the raw [C, 2D] table — no LDS chunk and no barrier, so a single lane can take it.  Returns log2 of the
// denominator; acc_n[d] = sum_{j != c} q_j tanh(x_jd / 2 sigma) A_jd with q = softmax over the classes.
__device__ __noinline__ float bwd_token_from_raw_table(const float* table, const float* prior, const float* z, int D, int C, int c,
                                                       float lp2, float sigma, float log_sigma, float* acc_n) {
    float m = -3e38f, ssum = 0.f;
    const float k = kLog2e / sigma;
    for (int d = 0; d < D; ++d) acc_n[d] = 0.f;
    for (int j = 0; j < C; ++j) {
        const float* row = table + (size_t)j * 2 * D;
        float acc = 0.f, prod = 1.f, tsum = 0.f, ta[kEncMaxD];
        for (int d = 0; d < D; ++d) {
            const float ts = tanhf(row[D + d]);
            const float A = expf(-ts) * k;
            const float xk = fmaf(z[d], A, -(row[d] * k));
            const float vs = fabsf(xk);
            const float e = __builtin_amdgcn_exp2f(-vs);
            acc += vs;
            prod = fmaf(prod, e, prod);
            ta[d] = copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), xk) * A;
            tsum += ts;
        }
        const bool own = j == c;
        const float v = own ? lp2 : ((prior[j] - tsum) - (float)D * log_sigma) * kLog2e - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
        const float mn = fmaxf(m, v);
        const float scale = __builtin_amdgcn_exp2f(m - mn), w = __builtin_amdgcn_exp2f(v - mn);
        ssum = fmaf(ssum, scale, w);
        const float wg = own ? 0.f : w;
        for (int d = 0; d < D; ++d) acc_n[d] = fmaf(acc_n[d], scale, wg * ta[d]);
        m = mn;
    }
    const float inv = 1.f / ssum;
    for (int d = 0; d < D; ++d) acc_n[d] *= inv;
    return m + __builtin_amdgcn_logf(ssum);
}

// PHASE as in encoder_tiled_kernel; a partial is (density sum, -, D accumulators).  The sweep sums class DENSITIES like
// the forward (class_density in cnf_encoder.hip): w_j = E_j prod_d q_jd (prod_d r_jd)^2 with q = 2^-|x|, r = 1 / (1 + q) —
// the reciprocals are the ones tanh(x / 2 sigma) = sign(x) (1 - q) r needs anyway — so a class costs D exp2 + D rcp and
// no logarithm, no exponential for a log-sum-exp and no rescaling of the D accumulators (rounds 2-3: 15 transcendental and
// ~70 plain instructions per class and token at D = 6, now 12 and ~58).  SINGLE: the whole class range is ONE chunk (every
// vocabulary up to ~585 classes at D = 6): the chunk is built once per workgroup instead of once per round of 256 tokens
// (two barriers and a serial tanhf chain per round), and a token's own-class constants come from it (e^ts = k / A,
// bias = C / k, lp2 = (init_lp + D log sigma) log2e + cst2) instead of 2D gathered table entries and D tanhf + expf.
template <int DT, int PHASE, bool SINGLE>
__global__ __launch_bounds__(kBlock) void encoder_bwd_token_kernel(EncBwdTiledArgs b, int CC, float* part, int KS, int stage_recs) {
    static_assert(!SINGLE || PHASE == 0, "one chunk = the whole class range");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* tab = reinterpret_cast<float*>(smem);
    const int D = DT > 0 ? DT : b.D;
    constexpr int DM = DT > 0 ? DT : kEncMaxD;
    const int stride = chunk_stride(D), R = 3 * D + 3;
    EncArgs a = {};
    a.table = b.table; a.prior = b.prior; a.D = D; a.C = b.C; a.sigma = b.sigma; a.log_sigma = b.log_sigma;
    const int per = (b.C + KS - 1) / KS;
    const int j_lo = PHASE == 1 ? (int)blockIdx.y * per : 0;
    const int j_hi = PHASE == 1 ? min(b.C, j_lo + per) : b.C;
    const int PS = 2 + D;
    const long rounds = (b.ntok + kBlock - 1) / kBlock;
    const float kn = kLog2e / b.sigma, inv_kn = b.sigma / kLog2e;
    // record stage behind the class chunk: [4 waves][64 tokens][R] floats (host: stage_recs; null = direct stores)
    float* rec_stage = (PHASE != 1 && stage_recs) ? tab + (size_t)CC * stride : nullptr;
    if (SINGLE) {
        build_class_chunk(a, tab, 0, b.C, D);
        __syncthreads();
    }
    for (long r = blockIdx.x; r < rounds; r += gridDim.x) {
        const long tok = r * kBlock + threadIdx.x;
        const bool live = tok < b.ntok;
        float z[DM], acc_g[DM], ets[DM];
        int c = -1;
        float lp2 = 0.f, G = 0.f, pv = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { z[d] = 0.f; acc_g[d] = 0.f; ets[d] = 1.f; }
        if (live) {
            const long long craw = b.categ[tok];        // range-checked (and reported) by the forward kernel
            c = (int)(craw < 0 ? 0 : (craw >= b.C ? b.C - 1 : craw));
            float nacc = 0.f, nprod = 1.f, ldj_f = 0.f;
            const float* row = b.table + (size_t)c * 2 * D;
            const float* tc = tab + c * stride;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float e = b.eps[tok * D + d];
                const float vs = fabsf(e) * kn;
                nacc += vs;
                nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
                if (SINGLE) {
                    ets[d] = kn * __builtin_amdgcn_rcpf(tc[2 * d]);
                    z[d] = fmaf(tc[2 * d + 1], inv_kn, e) * ets[d];
                } else {
                    const float ts = tanhf(row[D + d]);
                    ets[d] = expf(ts);
                    z[d] = (e + row[d]) * ets[d];
                    ldj_f += ts;
                }
            }
            const float init_lp = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * b.log_sigma);
            if (SINGLE) lp2 = fmaf(init_lp + (float)D * b.log_sigma, kLog2e, tc[2 * D]);
            else lp2 = ((init_lp - ldj_f) + b.prior[c]) * kLog2e;
            pv = b.pad ? b.pad[tok] : 1.f;
            G = (b.g_ldj ? b.g_ldj[tok / b.N] : 0.f) * pv;
        }
        float dsum = 0.f;
        auto sweep = [&](int j0, int cc) {
            for (int jj = 0; jj < cc; ++jj) {
                const float* t = tab + jj * stride;
                float num = t[2 * D + 1], rp = 1.f, ta[DM];
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    const float xk = fmaf(z[d], t[2 * d], -t[2 * d + 1]);
                    const float q = __builtin_amdgcn_exp2f(-fabsf(xk));
                    const float rr = __builtin_amdgcn_rcpf(1.f + q);
                    num *= q;
                    rp *= rr;
                    // tanh(x / 2 sigma) A = sign(x) (1 - q) / (1 + q) A
                    ta[d] = copysignf((1.f - q) * rr, xk) * t[2 * d];
                }
                const float w = (j0 + jj) == c ? 0.f : (num * rp) * rp;      // the true class is the 1 of the total
                dsum += w;
#pragma unroll
                for (int d = 0; d < D; ++d) acc_g[d] = fmaf(w, ta[d], acc_g[d]);
            }
        };
        if (SINGLE) {
            sweep(0, b.C);
        } else if (PHASE != 2) {
            for (int j0 = j_lo; j0 < j_hi; j0 += CC) {
                const int cc = min(CC, j_hi - j0);
                __syncthreads();
                build_class_chunk(a, tab, j0, cc, D);
                __syncthreads();
                sweep(j0, cc);
            }
        } else if (live) {
            for (int k = 0; k < KS; ++k) {                   // split order: deterministic
                const float* pk = part + ((size_t)k * b.ntok + tok) * PS;
                dsum += pk[0];
#pragma unroll
                for (int d = 0; d < D; ++d) acc_g[d] += pk[2 + d];
            }
        }
        if (live && PHASE == 1) {
            float* o = part + ((size_t)blockIdx.y * b.ntok + tok) * PS;
            o[0] = dsum;
            o[1] = 0.f;
#pragma unroll
            for (int d = 0; d < D; ++d) o[2 + d] = acc_g[d];
        }
        if (PHASE == 1) continue;
        if (live) {
            // total relative to the token's own density: tot = 1 + 2^-lp2 sum_{j != c} w_j;  q_j = 2^-lp2 w_j / tot
            const float F = __builtin_amdgcn_exp2f(-lp2);
            const float tot = fmaf(F, dsum, 1.f);
            float lse2, q_c;
            if (density_sum_ok(tot, lp2)) {
                const float inv_tot = __builtin_amdgcn_rcpf(tot);
                lse2 = lp2 + __builtin_amdgcn_logf(tot);
                q_c = inv_tot;
                const float fn = F * inv_tot;
#pragma unroll
                for (int d = 0; d < D; ++d) acc_g[d] *= fn;
            } else {
                // overflow / NaN: log domain, from the raw table (copies go to the callee: the registers stay registers)
                float zc[DM], an[DM];
#pragma unroll
                for (int d = 0; d < DM; ++d) zc[d] = z[d];
                lse2 = bwd_token_from_raw_table(b.table, b.prior, zc, D, b.C, c, lp2, b.sigma, b.log_sigma, an);
#pragma unroll
                for (int d = 0; d < DM; ++d) acc_g[d] = an[d];
                q_c = __builtin_amdgcn_exp2f(lp2 - lse2);
            }
            const float Gb = G * b.beta;
            const float g_ldjf = G - Gb * (1.f - q_c);
            const float norm = Gb * kLn2;
            // The records of a wave's 64 tokens are 64 R contiguous floats.  Written lane by lane they are R dword stores at
            // a 4 R-byte stride — 64 separate requests per instruction, 22 M of them at 10^6 tokens, which cost ~90 us (the
            // kernel took 125 us with 35 us of arithmetic) — so a wave transposes its records through its LDS strip and
            // stores them as R fully coalesced instructions.
            float* rec = rec_stage ? rec_stage + (size_t)(threadIdx.x >> 6) * kWave * R + (size_t)(threadIdx.x & 63) * R
                                   : b.rec + (size_t)tok * R;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float gz = fmaf(norm, acc_g[d], (b.g_zout ? b.g_zout[tok * D + d] : 0.f) * pv);
                rec[d] = z[d];
                rec[D + d] = gz * ets[d];
                rec[2 * D + d] = fmaf(gz, z[d], g_ldjf);
            }
            rec[3 * D] = lse2;
            rec[3 * D + 1] = Gb;
            rec[3 * D + 2] = __int_as_float(c);
        }
        if (rec_stage) {
            // all 64 lanes copy (lanes past the last token hold no record but help to store their neighbours')
            wave_lds_sync();
            const long wave_tok0 = r * kBlock + (threadIdx.x & ~63);
            const int nl = (int)max<long>(0, min<long>(kWave, b.ntok - wave_tok0));
            const float* src = rec_stage + (size_t)(threadIdx.x >> 6) * kWave * R;
            float* dst = b.rec + (size_t)wave_tok0 * R;
            for (int i = threadIdx.x & 63; i < nl * R; i += kWave) dst[i] = src[i];
            wave_lds_sync();                                 // the strip is rewritten in the next round
        }
    }
}

// token records staged per barrier: two LDS buffers, the next stage's records are in flight (in registers) while the
// current stage is scored.  (Rounds 2-3: one buffer of 64 records, load -> barrier -> score -> barrier: every stage paid a
// full memory round trip with ~1 100 cycles of arithmetic to hide it; 74 us at 10^6 tokens x 16 classes, 234 at 51.)
constexpr int enc_bwd_stage(int dt) { return (dt > 0 && dt <= 8) ? 128 : 64; }

template <int DT>
__global__ __launch_bounds__(kBlock) void encoder_bwd_class_kernel(EncBwdTiledArgs b, int cl_shift) {
    // 256 lanes = CL class lanes x TL token lanes (CL = 2^cl_shift >= min(C, 256)): small vocabularies keep the whole
    // workgroup busy by giving every class TL lanes that take every TL-th token record of a stage
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int D = DT > 0 ? DT : b.D;
    constexpr int DM = DT > 0 ? DT : kEncMaxD;
    constexpr int STAGE = enc_bwd_stage(DT);
    constexpr int NPF = (STAGE * (3 * DM + 3) + kBlock - 1) / kBlock;      // staged floats per thread
    const int R = 3 * D + 3;
    float* stage = reinterpret_cast<float*>(smem);                          // [2][STAGE * R]
    const int CL = 1 << cl_shift, TL = kBlock >> cl_shift;
    const int jl = threadIdx.x & (CL - 1), tl = threadIdx.x >> cl_shift;
    const int j = blockIdx.x * CL + jl;
    const bool live = j < b.C;
    float A[DM], Cb[DM], gb[DM], gt[DM], dts[DM];
    float cst2 = 0.f;
    {
        const float k = kLog2e / b.sigma;
        float ssum = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float* row = b.table + (size_t)(live ? j : 0) * 2 * D;
            const float ts = tanhf(row[D + d]);
            A[d] = expf(-ts) * k;
            Cb[d] = row[d] * k;
            dts[d] = 1.f - ts * ts;
            ssum += ts;
            gb[d] = 0.f;
            gt[d] = 0.f;
        }
        cst2 = ((b.prior[live ? j : 0] - ssum) - (float)D * b.log_sigma) * kLog2e;
    }
    // token range of this split, in whole stages
    const long per = ((b.ntok + b.S - 1) / b.S + STAGE - 1) / STAGE * STAGE;
    const long t0 = (long)blockIdx.y * per, t1 = min(t0 + per, b.ntok);
    const float inv_sigma = 1.f / b.sigma;
    float s1[DM], s2[DM], sgv = 0.f;               // sum of tanh gv, of tanh gv z, of gv over this lane's (token, class) pairs
#pragma unroll
    for (int d = 0; d < DM; ++d) { s1[d] = 0.f; s2[d] = 0.f; }
    float pf[NPF];
    auto fetch = [&](long ts0) {
        const int n = (int)min<long>(STAGE, t1 - ts0) * R;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = threadIdx.x + k * kBlock;
            pf[k] = i < n ? b.rec[(size_t)ts0 * R + i] : 0.f;
        }
    };
    auto put = [&](float* buf, long ts0) {
        const int n = (int)min<long>(STAGE, t1 - ts0) * R;
#pragma unroll
        for (int k = 0; k < NPF; ++k) {
            const int i = threadIdx.x + k * kBlock;
            if (i < n) buf[i] = pf[k];
        }
    };
    if (t0 < t1) {
        fetch(t0);
        put(stage, t0);
    }
    __syncthreads();
    int sidx = 0;
    for (long ts0 = t0; ts0 < t1; ts0 += STAGE, sidx ^= 1) {
        const int nt = (int)min<long>(STAGE, t1 - ts0);
        const float* cur = stage + (size_t)sidx * STAGE * R;
        const bool more = ts0 + STAGE < t1;
        if (more) fetch(ts0 + STAGE);                    // in flight while this stage is scored
        if (live) {
        for (int t = tl; t < nt; t += TL) {
            const float* rec = cur + t * R;              // one address per token lane: LDS broadcast within it
            const int c = __float_as_int(rec[3 * D + 2]);
            if (c == j) {                                // exactly one lane of the whole grid per token
#pragma unroll
                for (int d = 0; d < D; ++d) { gb[d] += rec[D + d]; gt[d] += rec[2 * D + d]; }
                continue;
            }
            float acc = 0.f, prod = 1.f, th[DM];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float xk = fmaf(rec[d], A[d], -Cb[d]);
                const float vs = fabsf(xk);
                const float e = __builtin_amdgcn_exp2f(-vs);
                acc += vs;
                prod = fmaf(prod, e, prod);
                th[d] = copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), xk);      // tanh(x / 2 sigma)
            }
            const float v = cst2 - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
            const float gv = -rec[3 * D + 1] * __builtin_amdgcn_exp2f(v - rec[3 * D]);  // d loss / d v_j = -beta G q_j
            // d v_j / d bias = tanh / sigma;  d v_j / d ts = tanh z e^{-ts} / sigma - 1  (A ln2 = e^{-ts} / sigma): the
            // per-class constants (1 / sigma, ln2 A_d, the -1) are applied ONCE to the sums, after the token loop
            sgv += gv;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float tg = th[d] * gv;
                s1[d] += tg;
                s2[d] = fmaf(tg, rec[d], s2[d]);
            }
        }
        }
        if (more) put(stage + (size_t)(sidx ^ 1) * STAGE * R, ts0 + STAGE);
        __syncthreads();                                 // the other buffer is complete; this one may be overwritten next time
    }
#pragma unroll
    for (int d = 0; d < D; ++d) {
        gb[d] = fmaf(s1[d], inv_sigma, gb[d]);
        gt[d] += fmaf(s2[d] * kLn2, A[d], -sgv);
    }
    // combine the token lanes of a class in lane order, then through tanh to the raw scale
    if (TL > 1) {
        __syncthreads();
        float* red = stage;                              // [TL][CL][2D]
#pragma unroll
        for (int d = 0; d < D; ++d) {
            red[((size_t)tl * CL + jl) * 2 * D + d] = gb[d];
            red[((size_t)tl * CL + jl) * 2 * D + D + d] = gt[d];
        }
        __syncthreads();
        if (tl == 0) {
            for (int o = 1; o < TL; ++o) {
#pragma unroll
                for (int d = 0; d < D; ++d) {
                    gb[d] += red[((size_t)o * CL + jl) * 2 * D + d];
                    gt[d] += red[((size_t)o * CL + jl) * 2 * D + D + d];
                }
            }
        }
    }
    if (!live || tl != 0) return;
    float* out = b.partials + ((size_t)blockIdx.y * b.C + j) * 2 * D;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        out[d] = gb[d];
        out[D + d] = gt[d] * dts[d];
    }
}

// g_table[p] = sum over the splits, fixed order: a workgroup owns 16 table entries, 16 lanes per entry each add every
// 16th split (fp64), then the 16 lane sums are combined in lane order.  (One lane per entry walking all S splits was
// a chain of S dependent-latency loads: 380 us of the 430 us backward at S = 1024.)
__global__ __launch_bounds__(kBlock) void encoder_bwd_splits_kernel(const float* partials, int S, long P, float* out) {
    __shared__ double red[16][17];
    const int pl = threadIdx.x & 15, sl = threadIdx.x >> 4;
    const long p = (long)blockIdx.x * 16 + pl;
    double acc = 0.0;
    if (p < P) {
        int s = sl;
        for (; s + 48 < S; s += 64) {                   // four independent loads in flight per lane
            const float v0 = partials[(size_t)s * P + p], v1 = partials[(size_t)(s + 16) * P + p];
            const float v2 = partials[(size_t)(s + 32) * P + p], v3 = partials[(size_t)(s + 48) * P + p];
            acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
        }
        for (; s < S; s += 16) acc += (double)partials[(size_t)s * P + p];
    }
    red[sl][pl] = acc;
    __syncthreads();
    if (sl == 0 && p < P) {
        double t = 0.0;
        for (int k = 0; k < 16; ++k) t += red[k][pl];
        out[p] = (float)t;
    }
}


// Workgroup barrier for data exchanged through LDS only: the wave's LDS queue is drained (lgkmcnt) and the waves meet, but
// the vector-memory counter is left alone.  __syncthreads() carries a release fence that on gfx9 also waits for every
// outstanding global LOAD — here the next stage's inputs, issued one stage ahead precisely so that nobody waits for them
// (measured with it: 5.8 us per stage, one memory round trip each; the whole backward 119 us at 10^6 tokens x 16 classes).
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// -DCNF_ENC_BWD_PROBE (tools/build_variant.sh): every wave of the pair kernel adds up the 100 MHz ticks it spends in X, at
// barrier a, in Y and at barrier b and stores the sums behind the workgroup tables of the workspace
// (tools/encoder_bwd_phases.py reads them).
#ifdef CNF_ENC_BWD_PROBE
#define PROBE_DECL unsigned long long pt_[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, pl_ = __builtin_amdgcn_s_memrealtime()
#define PROBE(i) do { const unsigned long long n_ = __builtin_amdgcn_s_memrealtime(); pt_[i] += n_ - pl_; pl_ = n_; } while (0)
#define PROBE_STORE do { if ((threadIdx.x & 63) == 0) { unsigned long long* o_ = reinterpret_cast<unsigned long long*>(b.partials + (size_t)gridDim.x * b.C * 2 * DT) + ((size_t)blockIdx.x * (NT / 64) + (threadIdx.x >> 6)) * 10; for (int i_ = 0; i_ < 10; ++i_) o_[i_] = pt_[i_]; } } while (0)
#else
#define PROBE_DECL
#define PROBE(i)
#define PROBE_STORE
#endif

// ---- round 5: ONE pass over the (token, class) pairs, token denominators KNOWN -----------------------------------------
//
// The two passes above evaluate the T x C x D tensor of reverse-flow terms twice (once per reduction axis: over the classes
// for d loss / d z, over the tokens for the class rows), 12 transcendental instructions per pair each time.  Here a pair
// lane owns a CLASS (its 2D + 1 constants and 2D + 1 gradient sums in registers) and U tokens of a STAGE of ST = TL x U
// tokens (TL token lanes: pair lane = TL index x C + class).  With log q_c = class_prob_log from the forward, a token's
// factor -beta G 2^-lse2 (lse2 = lp2 - log2 q_c) is known before its pairs are scored, so a pair's terms
// g_jd = -beta G q_j tanh(x_jd / 2 sigma) A_jd go straight into the class sums and nothing is kept in registers across a
// barrier; the reduction over the classes (through LDS: [token][channel][class], class contiguous) is needed only for the
// token's OWN-class gradient, which may lag behind.  256 lanes = 3 waves of pair lanes + ONE token wave (which wave it is
// rotates with the workgroup index: see the kernel) that runs ahead of / behind them:
//   X(s)  pair waves : pairs of stage s -> class sums (registers), terms g_jd -> LDS; own-class gradient of stage s - 2
//         token wave : own-class gradients of stage s - 1 (from the class-summed terms), records of stage s + 1 (z, the
//                      factor, the class), global loads of stage s + 2
//   Y(s)  all        : terms of stage s summed over the classes, class order -> record
// two barriers per stage (LDS-only: lds_barrier), none of them behind a lone wave's serial work.  Workgroup sums go to
// partials[workgroup], added in workgroup order by encoder_bwd_splits64_kernel: bit-reproducible, no floating-point atomics.
// A token whose own density is outside the fp32 range of the density sum (density_sum_ok, as the forward decided) has its
// pairs scored in the log domain.
// What was measured on the way (1 048 576 tokens x 16 classes, D = 6; the two passes: 117-120 us):
//   * denominators computed in the kernel (phase A keeps w, w ta_d of U tokens in registers, a class reduction and a token
//     phase between A and B, four barriers per stage): 116 us — every phase is a short latency-bound section behind a barrier;
//   * __syncthreads() in the stage loop: its release fence waits for the global loads of the NEXT stage: +5 us per stage;
//   * 4 pair waves + token wave (320 lanes): two workgroups per CU instead of the three the occupancy query promises: 143 us;
//   * own-class adds as a separate phase after the class sums: 0.5 us per stage of exposed LDS latency;
//   * sign-free tanh / density from r = 1 / (1 + e^-y): 92 us but 4e-3 off (see the pair loop);
//   * a branch-free copy of the pair loop for full stages without log-domain tokens (no per-token tests, four slots in one
//     basic block): 141 registers -> three workgroups per CU, 105 us; capped at 128 registers (10 spilled): 97 us;
// this form: 94-99 us (the pair arithmetic alone is 31 us of VALU issue: tools/isa_cost.py, profiles/r05_op_rates.txt).
struct PairsGeom {
    int TL, ST, rs, RT;
    long nstages;
    int per_wg;
    int rec_off, ctr_off;
};

template <int DT, int U, int PW>          // PW pair waves + one token wave
__global__ __launch_bounds__(PW * 64 + 64) void encoder_bwd_pairs_kernel(EncBwdTiledArgs b, const float* cpl, PairsGeom g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = DT, NT = PW * 64 + 64;
    // record fields
    constexpr int F_GFAC = D, F_C = D + 1, F_GZP = D + 2, F_GLDJF = 2 * D + 2, F_SUMS = 2 * D + 3, F_OWNB = 3 * D + 3, F_LSE = 4 * D + 3;
    float* tab = reinterpret_cast<float*>(smem);
    float* rec0 = tab + g.rec_off;
    float* ctr = tab + g.ctr_off;
    const int stride = chunk_stride(D), RT = g.RT, rs = g.rs, ST = g.ST, TL = g.TL;
    const int L = threadIdx.x, lane = L & 63, wv = L >> 6;
    // which wave is the token wave rotates with the workgroup index: the token waves of the workgroups that share a CU then
    // sit on different SIMDs (wave w of every workgroup lands on the same one) instead of leaving one SIMD without pair work
    const int tw = (int)((blockIdx.x >> 3) % (unsigned)(PW + 1));
    const bool token_wave = wv == tw;
    const int PL = (wv - (wv > tw ? 1 : 0)) * 64 + lane;            // pair-lane index
    {
        EncArgs a = {};
        a.table = b.table; a.prior = b.prior; a.D = D; a.C = b.C; a.sigma = b.sigma; a.log_sigma = b.log_sigma;
        build_class_chunk(a, tab, 0, b.C, D);
        for (int i = L; i < ST * D * rs; i += NT) ctr[i] = 0.f;     // the padding columns [C, rs) stay zero
        for (int i = L; i < 3 * ST * RT; i += NT) rec0[i] = 0.f;    // own-class fields are read (and masked) before they are first written
    }
    __syncthreads();
    const int tl = PL / b.C, j = PL - tl * b.C;
    const bool live = !token_wave && tl < TL;
    float A[D], Cb[D], E = 0.f, cst2 = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) {
        A[d] = live ? tab[j * stride + 2 * d] : 0.f;
        Cb[d] = live ? tab[j * stride + 2 * d + 1] : 0.f;
    }
    if (live) { cst2 = tab[j * stride + 2 * D]; E = tab[j * stride + 2 * D + 1]; }
    float s1[D], s2[D], sgv = 0.f;
#pragma unroll
    for (int d = 0; d < D; ++d) { s1[d] = 0.f; s2[d] = 0.f; }
    const float kn = kLog2e / b.sigma, inv_kn = b.sigma / kLog2e;
    const long sg0 = (long)blockIdx.x * g.per_wg;
    const int ns = (int)max<long>(0, min<long>(g.nstages - sg0, g.per_wg));   // this workgroup's stages, sg0 + [0, ns)
    const int nch = (ST + kWave - 1) / kWave;                        // token-wave chunks of 64 tokens per stage (1 unless C is tiny)
    const int recsz = ST * RT;

    // ---- token wave: loads, records, own-class gradients ---------------------------------------------------------------
    // `load` is unconditional (a lane past the end re-reads the last token) and nothing in it USES a loaded value: a
    // conditional or consumed load makes the wave wait for memory on the spot.  An absent optional input is read from `eps`.
    float pe[D], pg[D], ppv, pgl, pcp;
    long long pcraw;
    const float* gz_src = b.g_zout ? b.g_zout : b.eps;
    const float* pad_src = b.pad ? b.pad : b.eps;
    const float* gl_src = b.g_ldj ? b.g_ldj : b.eps;
    const bool has_gz = b.g_zout != nullptr, has_pad = b.pad != nullptr, has_gl = b.g_ldj != nullptr;
    auto load = [&](int s, int ch) {
        const long tok = min((sg0 + s) * ST + ch * kWave + lane, b.ntok - 1);
        pcraw = b.categ[tok];                           // range-checked (and reported) by the forward kernel
#pragma unroll
        for (int d = 0; d < D; ++d) {
            pe[d] = b.eps[tok * D + d];
            pg[d] = gz_src[tok * D + d];
        }
        ppv = pad_src[tok];
        pgl = gl_src[(unsigned)tok / (unsigned)b.N];
        pcp = cpl[tok];
    };
    auto record = [&](int s, int ch, float* rbuf) {     // consumes the loaded values
        const int t = ch * kWave + lane;
        if (t < ST && (sg0 + s) * ST + t < b.ntok) {
            float* r = rbuf + t * RT;
            const int c = (int)(pcraw < 0 ? 0 : (pcraw >= b.C ? b.C - 1 : pcraw));
            const float pv = has_pad ? ppv : 1.f;
            const float* tc = tab + c * stride;
            float nacc = 0.f, nprod = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float e = pe[d];
                const float vs = fabsf(e) * kn;
                nacc += vs;
                nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
                const float ets = kn * __builtin_amdgcn_rcpf(tc[2 * d]);
                r[d] = fmaf(tc[2 * d + 1], inv_kn, e) * ets;
                r[F_GZP + d] = has_gz ? pg[d] * pv : 0.f;
            }
            const float init_lp = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * b.log_sigma);
            const float lp2 = fmaf(init_lp + (float)D * b.log_sigma, kLog2e, tc[2 * D]);
            const float cp2 = pcp * kLog2e;                                  // log2 q_c
            const float q_c = __builtin_amdgcn_exp2f(cp2);
            const float lse2 = lp2 - cp2;
            const float H = __builtin_amdgcn_exp2f(-lse2);
            const bool cold = !(lp2 > -90.f && H < 3e38f);                  // density_sum_ok's territory, as the forward saw it
            const float G = has_gl ? pgl * pv : 0.f;
            const float Gb = G * b.beta;
            r[F_GFAC] = cold ? -Gb : -Gb * H;
            r[F_C] = __int_as_float(c | (cold ? (1 << 30) : 0));
            r[F_GLDJF] = G - Gb * (1.f - q_c);
            r[F_LSE] = lse2;
        }
    };
    auto own_grads = [&](int s, int ch, float* rbuf) {  // stage s: class-summed terms -> own-class gradient in the pair sums' units
        const int t = ch * kWave + lane;
        if (t < ST && (sg0 + s) * ST + t < b.ntok) {
            float* r = rbuf + t * RT;
            const float g_ldjf = r[F_GLDJF];
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float gz = fmaf(-kLn2, r[F_SUMS + d], r[F_GZP + d]);       // sum_j g_jd = -beta G sum_j q_j ta_jd
                r[F_OWNB + d] = gz * kLog2e;                                      // g_bias sigma A_c = gz e^ts sigma A_c = gz log2e
                r[F_SUMS + d] = fmaf(gz, r[d], g_ldjf) * kLog2e;                  // g_ts log2e
            }
        }
    };
    // record buffers of stages s - 2, s - 1, s, s + 1 (three buffers: s + 1 and s - 2 share one)
    float *r_m2 = rec0 + recsz, *r_m1 = rec0 + 2 * recsz, *r_0 = rec0, *r_p1 = rec0 + recsz;
    if (token_wave && ns > 0) {
        for (int ch = 0; ch < nch; ++ch) { load(0, ch); record(0, ch, r_0); }
        if (nch == 1 && ns > 1) load(1, 0);
    }
    lds_barrier();

    unsigned own_p1 = 0, own_p2 = 0;
    PROBE_DECL;
    for (int s = 0; s < ns; ++s) {
        const int nt = (int)min<long>(ST, b.ntok - (sg0 + s) * ST);
        unsigned own = 0;
        // ---- X ------------------------------------------------------------------------------------------------------------
        if (token_wave) {
            if (s > 0)
                for (int ch = 0; ch < nch; ++ch) own_grads(s - 1, ch, r_m1);
            if (s + 1 < ns) {
                if (nch == 1) {
                    record(s + 1, 0, r_p1);
                    if (s + 2 < ns) load(s + 2, 0);
                } else {
                    for (int ch = 0; ch < nch; ++ch) { load(s + 1, ch); record(s + 1, ch, r_p1); }
                }
            }
        } else if (live) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = tl + u * TL;
                // the own-class gradient of the token this lane scored two stages ago (zero unless its class is this lane's):
                // read now, added at the end of the slot, so that the LDS round trip hides behind the pair arithmetic
                const float* ro = r_m2 + t * RT;
                const float om = ((own_p2 >> u) & 1u) ? 1.f : 0.f;
                float ob[D], ot[D];
#pragma unroll
                for (int d = 0; d < D; ++d) { ob[d] = ro[F_OWNB + d]; ot[d] = ro[F_SUMS + d]; }
                if (t < nt) {
                    const float* r = r_0 + t * RT;
                    const int cf = __float_as_int(r[F_C]);
                    const float gfac = r[F_GFAC];
                    const bool is_own = (cf & ~(1 << 30)) == j;
                    own |= is_own ? (1u << u) : 0u;
                    float ta[D], wg;
                    if (!(cf >> 30)) {
                        // (measured and dropped: the sign-free forms tanh(y / 2) = 2 r - 1, density r (1 - r) with r = 1 / (1 + e^-y)
                        // save 3 ns of 9 per channel, but v_rcp_f32's error in r is amplified by 1 / (1 - r): 4e-3 in the gradient)
                        float num = E;
                        float rp = 1.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const float xk = fmaf(r[d], A[d], -Cb[d]);
                            const float q = __builtin_amdgcn_exp2f(-fabsf(xk));
                            const float rr = __builtin_amdgcn_rcpf(1.f + q);
                            num *= q;
                            rp *= rr;
                            // tanh(x / 2 sigma) A = sign(x) (1 - q) / (1 + q) A = sign(x) (2 / (1 + q) - 1) A
                            ta[d] = copysignf(fmaf(2.f, rr, -1.f), xk) * A[d];
                        }
                        num = (num * rp) * rp;
                        wg = is_own ? 0.f : num * gfac;                           // -beta G q_j
                    } else {
                        // a token outside the fp32 range of the densities: q_j = 2^(v_j - lse2) in the log domain
                        float acc = 0.f, prod = 1.f;
#pragma unroll
                        for (int d = 0; d < D; ++d) {
                            const float xk = fmaf(r[d], A[d], -Cb[d]);
                            const float vs = fabsf(xk);
                            const float e = __builtin_amdgcn_exp2f(-vs);
                            acc += vs;
                            prod = fmaf(prod, e, prod);
                            ta[d] = copysignf((1.f - e) * __builtin_amdgcn_rcpf(1.f + e), xk) * A[d];
                        }
                        const float v = cst2 - fmaf(2.f, __builtin_amdgcn_logf(prod), acc);
                        wg = is_own ? 0.f : gfac * __builtin_amdgcn_exp2f(v - r[F_LSE]);
                    }
                    sgv += wg;
                    float* cp = ctr + (size_t)t * D * rs + j;
#pragma unroll
                    for (int d = 0; d < D; ++d) {
                        const float gd = wg * ta[d];
                        s1[d] += gd;
                        s2[d] = fmaf(gd, r[d], s2[d]);
                        cp[d * rs] = gd;
                    }
                }
#pragma unroll
                for (int d = 0; d < D; ++d) { s1[d] = fmaf(om, ob[d], s1[d]); s2[d] = fmaf(om, ot[d], s2[d]); }
            }
        }
        PROBE(0);
        lds_barrier();
        PROBE(1);
        // ---- Y: the stage's terms summed over the classes, class order -------------------------------------------------------
        {
            // 8-byte reads: rows an odd multiple of 8 bytes apart are conflict-free and need 2 (not 4) floats of padding.
            // (Measured and dropped: a cell's classes dealt to 2 or 4 adjacent lanes with quad-permute sums, so that 288 cells fill
            // 256 lanes evenly / 192 cells use all 512 — 97 -> 106 / 116 us at 16 classes, 266 -> 273 / 282 at 51: the cells
            // are not what the barrier behind this phase waits for.)
            const int c2 = (b.C + 1) >> 1;
            for (int cell = L; cell < nt * D; cell += NT) {
                const float2* p = reinterpret_cast<const float2*>(ctr + (size_t)cell * rs);
                float2 a0 = p[0], a1 = make_float2(0.f, 0.f);
                int i = 1;
                for (; i + 1 < c2; i += 2) {
                    const float2 v = p[i], w2 = p[i + 1];
                    a1.x += v.x; a1.y += v.y;
                    a0.x += w2.x; a0.y += w2.y;
                }
                if (i < c2) { const float2 v = p[i]; a1.x += v.x; a1.y += v.y; }
                const int t = cell / D, d = cell - t * D;
                r_0[t * RT + F_SUMS + d] = (a0.x + a0.y) + (a1.x + a1.y);
            }
        }
        own_p2 = own_p1;
        own_p1 = own;
        {   // rotate the record buffers
            float* f = r_m2;
            r_m2 = r_m1; r_m1 = r_0; r_0 = r_p1; r_p1 = r_m2;
            (void)f;
        }
        PROBE(2);
        lds_barrier();
        PROBE(3);
    }
    PROBE_STORE;
    // the last two stages' own-class gradients (after the rotation: stage ns - 1 is r_m1, ns - 2 is r_m2)
    if (ns > 0) {
        if (token_wave)
            for (int ch = 0; ch < nch; ++ch) own_grads(ns - 1, ch, r_m1);
        lds_barrier();
        if (live) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int t = tl + u * TL;
                if ((own_p2 >> u) & 1u) {
#pragma unroll
                    for (int d = 0; d < D; ++d) { s1[d] += r_m2[t * RT + F_OWNB + d]; s2[d] += r_m2[t * RT + F_SUMS + d]; }
                }
                if ((own_p1 >> u) & 1u) {
#pragma unroll
                    for (int d = 0; d < D; ++d) { s1[d] += r_m1[t * RT + F_OWNB + d]; s2[d] += r_m1[t * RT + F_SUMS + d]; }
                }
            }
        }
    }
    // ---- the workgroup's table: token lanes of a class in lane order, then through tanh to the raw scale ----------------
    __syncthreads();
    float* red = ctr;                                          // [TL][C][2D + 1]
    if (live) {
        float* o = red + ((size_t)tl * b.C + j) * (2 * D + 1);
#pragma unroll
        for (int d = 0; d < D; ++d) { o[d] = s1[d]; o[D + d] = s2[d]; }
        o[2 * D] = sgv;
    }
    __syncthreads();
    if (!token_wave && PL < b.C) {
        float t1[D], t2[D], tg = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) { t1[d] = 0.f; t2[d] = 0.f; }
        for (int o = 0; o < TL; ++o) {
            const float* p = red + ((size_t)o * b.C + PL) * (2 * D + 1);
#pragma unroll
            for (int d = 0; d < D; ++d) { t1[d] += p[d]; t2[d] += p[D + d]; }
            tg += p[2 * D];
        }
        float* out = b.partials + ((size_t)blockIdx.x * b.C + PL) * 2 * D;
        const float* row = b.table + (size_t)PL * 2 * D;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float ts = tanhf(row[D + d]);
            out[d] = t1[d] / (b.sigma * A[d]);
            out[D + d] = fmaf(t2[d], kLn2, -tg) * (1.f - ts * ts);
        }
    }
}

// log q_c per token for callers that did not keep the forward's class_prob_log: the forward's density sum and nothing else
// (cnf_encoder.hip: class_density, density_sum_ok), one token per lane, the whole class range as ONE chunk in LDS.
template <int DT>
__global__ __launch_bounds__(kBlock) void encoder_bwd_cpl_kernel(EncBwdTiledArgs b, float* cpl) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int D = DT;
    float* tab = reinterpret_cast<float*>(smem);
    const int stride = chunk_stride(D);
    EncArgs a = {};
    a.table = b.table; a.prior = b.prior; a.D = D; a.C = b.C; a.sigma = b.sigma; a.log_sigma = b.log_sigma;
    build_class_chunk(a, tab, 0, b.C, D);
    __syncthreads();
    const float kn = kLog2e / b.sigma, inv_kn = b.sigma / kLog2e;
    for (long tok = (long)blockIdx.x * kBlock + threadIdx.x; tok < b.ntok; tok += (long)gridDim.x * kBlock) {
        const long long craw = b.categ[tok];
        const int c = (int)(craw < 0 ? 0 : (craw >= b.C ? b.C - 1 : craw));
        const float* tc = tab + c * stride;
        float z[D], nacc = 0.f, nprod = 1.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float e = b.eps[tok * D + d];
            const float vs = fabsf(e) * kn;
            nacc += vs;
            nprod = fmaf(nprod, __builtin_amdgcn_exp2f(-vs), nprod);
            z[d] = fmaf(tc[2 * d + 1], inv_kn, e) * (kn * __builtin_amdgcn_rcpf(tc[2 * d]));
        }
        const float init_lp = -(kLn2 * fmaf(2.f, __builtin_amdgcn_logf(nprod), nacc) + (float)D * b.log_sigma);
        const float lp2 = fmaf(init_lp + (float)D * b.log_sigma, kLog2e, tc[2 * D]);
        float dsum = 0.f;
        for (int jj = 0; jj < b.C; ++jj) {
            const float* t = tab + jj * stride;
            float num = t[2 * D + 1], den = 1.f;
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const float q = __builtin_amdgcn_exp2f(-fabsf(fmaf(z[d], t[2 * d], -t[2 * d + 1])));
                num *= q;
                den = fmaf(den, q, den);
            }
            const float r = __builtin_amdgcn_rcpf(den);
            dsum += jj == c ? 0.f : (num * r) * r;
        }
        const float tot = fmaf(__builtin_amdgcn_exp2f(-lp2), dsum, 1.f);
        float cp;
        if (density_sum_ok(tot, lp2)) {
            cp = -kLn2 * __builtin_amdgcn_logf(tot);
        } else {
            float zc[kEncMaxD], an[kEncMaxD];
#pragma unroll
            for (int d = 0; d < D; ++d) zc[d] = z[d];
            cp = (lp2 - bwd_token_from_raw_table(b.table, b.prior, zc, D, b.C, c, lp2, b.sigma, b.log_sigma, an)) * kLn2;
        }
        cpl[tok] = cp;
    }
}

// The same sum with 64 lanes per table entry (a workgroup owns 4 entries): each lane adds every 64th split in fp64, the 64
// lane sums are combined in lane order.  For the S <= 2048 workgroup tables of the pair kernel: 3 us instead of 7.
__global__ __launch_bounds__(kBlock) void encoder_bwd_splits64_kernel(const float* partials, int S, long P, float* out) {
    __shared__ double red[4][kWave + 1];
    const int sl = threadIdx.x & 63, pl = threadIdx.x >> 6;
    const long p = (long)blockIdx.x * 4 + pl;
    double acc = 0.0;
    if (p < P) {
        int s = sl;
        for (; s + 192 < S; s += 256) {                 // four independent loads in flight per lane
            const float v0 = partials[(size_t)s * P + p], v1 = partials[(size_t)(s + 64) * P + p];
            const float v2 = partials[(size_t)(s + 128) * P + p], v3 = partials[(size_t)(s + 192) * P + p];
            acc += (double)v0; acc += (double)v1; acc += (double)v2; acc += (double)v3;
        }
        for (; s < S; s += 64) acc += (double)partials[(size_t)s * P + p];
    }
    red[pl][sl] = acc;
    __syncthreads();
    if (sl == 0 && p < P) {
        double t = 0.0;
        for (int k = 0; k < kWave; ++k) t += red[pl][k];
        out[p] = (float)t;
    }
}

}  // namespace cnf

using namespace cnf;

extern "C" {

static int bwd_class_shift(int C) {
    int sh = 0;
    while ((1 << sh) < std::min(C, kBlock)) ++sh;
    return sh;
}
static int bwd_tiled_splits(long ntok, int C) {
    const int groups = (C + kBlock - 1) / kBlock;
    const long by_tokens = std::max<long>(1, ntok / 256);           // at least 256 tokens per split
    return (int)std::max<long>(1, std::min<long>(by_tokens, (1024 + groups - 1) / groups));
}

// ---- the pair kernel's geometry --------------------------------------------------------------------------------------
static std::atomic<int> g_enc_bwd_kernel{0};
constexpr int kPairWaves = 3, kPairTokens = 4;          // 3 pair waves + the token wave = a 256-lane workgroup; 4 tokens per pair lane
constexpr int kPairWavesWide = 7;                        // the wide workgroup: 7 pair waves + the token wave = 512 lanes
constexpr int kPairMaxC = kPairWaves * 64, kPairMaxCWide = kPairWavesWide * 64;
constexpr int kPairMaxWgs = 2048;
static bool pair_shape_ok(long ntok, int D, int C, int max_c = kPairMaxCWide) {
    return C <= max_c && ntok < (1l << 31) && (D <= 4 || D == 6 || D == 8);
}
// Which route (profiles/r05_encoder_bwd_sweep.txt, r05_encoder_bwd_sweep_wide.txt, one MI355X): 0 = the two passes, kPairWaves = the
// 256-lane pair workgroup, kPairWavesWide = the 512-lane one.  With the forward's class_prob_log: the narrow workgroup at 9 ... 27
// classes (16 classes: 94 vs 117 us at 10^6 tokens), the wide one from 28 classes on, where a stage of 448 / C x 4 tokens still fits
// the token wave's 64 lanes and the class count fills more of 448 lanes than of 192 (51 classes: 265 vs 316 us for the two passes and
// 328 for the narrow workgroup; 42: 214 vs 296; 64: 314 vs 343) — up to 64 classes at any size, up to 200 on batches of at most
// 131 072 tokens (96 classes: 53 vs 68 us at 65 536 tokens); on small batches (<= 16 384 tokens) at every class count the workgroups
// can hold (4 096 tokens: 14-17 us up to 64 classes, 31-58 us at 200-448, against 28-168 us).  Below 9 classes the class sums in the
// token lanes' registers are cheaper; beyond these limits a stage holds too few tokens for the workgroup.  Without class_prob_log (the
// pre-pass repeats the forward's density sum): small batches of 16 classes or more only.
static int pair_kernel_choice(long ntok, int C, bool have_cpl) {
    const int small = C <= 48 ? kPairWaves : kPairWavesWide;
    if (ntok <= 16384) return (have_cpl || C >= 16) ? small : 0;
    if (!have_cpl || C < 9) return 0;
    if (C <= 27) return kPairWaves;
    if (C <= 64) return kPairWavesWide;
    if (ntok <= 131072 && C <= 200) return kPairWavesWide;
    if (C <= 96) return kPairWaves;
    return 0;
}
static size_t make_pairs_geom(long ntok, int D, int C, int U, int pair_lanes, PairsGeom& g, int resident_per_cu = 0) {
    int rs = (C + 1) & ~1;
    if (((rs >> 1) & 1) == 0) rs += 2;                  // rows an odd multiple of 8 bytes apart (8-byte reads)
    g.rs = rs;
    g.RT = 4 * D + 5;
    auto up4 = [](size_t v) { return (v + 3) & ~(size_t)3; };
    const size_t tab = up4((size_t)C * (2 * D + 2));
    size_t rec = 0, ctr = 0, lds = 0;
    for (g.TL = std::max(1, std::min(pair_lanes / C, pair_lanes / U));; --g.TL) {
        g.ST = g.TL * U;
        rec = up4((size_t)3 * g.ST * g.RT);
        ctr = up4(std::max((size_t)g.ST * D * rs, (size_t)g.TL * C * (2 * D + 1)));
        lds = (tab + rec + ctr) * sizeof(float);
        if (lds <= 64 * 1024 || g.TL == 1) break;
    }
    g.nstages = (ntok + g.ST - 1) / g.ST;
    g.rec_off = (int)tab;
    g.ctr_off = (int)(tab + rec);
    // workgroups: what the chip holds at once (the runtime's occupancy figure for this kernel and LDS size; before it is
    // known: by LDS alone), consecutive stages each.  More, shorter workgroups do not balance the CUs' finishing times, they add
    // their start-up: 2 / 3 / 4 / 8 times as many at 1 048 576 tokens: 97 -> 105 / 112 / 122 / 156 us at 16 classes, 266 -> 269 /
    // 281 / 281 / 313 at 51
    const int per_cu = resident_per_cu > 0 ? resident_per_cu : (int)std::max<size_t>(1, std::min<size_t>(6, (160 * 1024) / lds));
    const long wgs = std::min<long>(std::min<long>(g.nstages, 256l * per_cu), kPairMaxWgs);
    g.per_wg = (int)((g.nstages + wgs - 1) / wgs);
    return lds;
}

}  // extern "C"

// false = the stage does not fit 64 KB of LDS at this class count (the caller takes the two passes)
template <int DT, int U, int PW>
static bool launch_pairs(EncBwdTiledArgs& b, const float* cpl, int D, int C, hipStream_t st) {
    constexpr int kPairsNT = PW * 64 + 64;
    PairsGeom g;
    size_t lds = make_pairs_geom(b.ntok, D, C, U, PW * 64, g);
    if (lds > 64 * 1024) return false;
    // resident workgroups per CU for this instantiation and LDS size (registers and wave slots included), cached
    static std::mutex mu;
    static std::map<size_t, int> cache;
    int per_cu = 0;
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = cache.find(lds);
        if (it != cache.end()) {
            per_cu = it->second;
        } else {
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&encoder_bwd_pairs_kernel<DT, U, PW>), kPairsNT, lds) != hipSuccess || per_cu <= 0)
                per_cu = 1;
            (void)hipGetLastError();
            cache[lds] = per_cu;
        }
    }
    lds = make_pairs_geom(b.ntok, D, C, U, PW * 64, g, per_cu);
    const int wgs = (int)((g.nstages + g.per_wg - 1) / g.per_wg);
    b.S = wgs;
    CNF_LAUNCH((encoder_bwd_pairs_kernel<DT, U, PW>), dim3(wgs), dim3(kPairsNT), lds, st, b, cpl, g);
    return true;
}

extern "C" {

void cnf_set_encoder_bwd_kernel(int which) { g_enc_bwd_kernel.store(which, std::memory_order_relaxed); }

int64_t cnf_encoder_bwd_tiled_workspace_floats(int B, int N, int D, int C) {
    const long ntok = (long)B * N;
    const int ks = tiled_class_splits(C);
    const int64_t passes = (int64_t)ntok * (3 * D + 3) + (int64_t)bwd_tiled_splits(ntok, C) * C * 2 * D + (ks > 1 ? (int64_t)ks * ntok * (2 + D) : 0);
    const int64_t pair = pair_shape_ok(ntok, D, C) ? (int64_t)kPairMaxWgs * C * 2 * D + ntok : 0;      // workgroup tables + log q_c per token
    return std::max(passes, pair);
}

static int encoder_bwd_dispatch(const char* what, const int64_t* categ, const float* eps, const float* table,
                                const float* category_prior, const float* pad, float beta, const float* class_prob_log,
                                const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream);

int cnf_encoder_forward_bwd_tiled(const int64_t* categ, const float* eps, const float* table,
                                  const float* category_prior, const float* pad, float beta,
                                  const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                  int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream) {
    return encoder_bwd_dispatch("cnf_encoder_forward_bwd_tiled", categ, eps, table, category_prior, pad, beta, nullptr, g_zout, g_ldj,
                                g_table, workspace, B, N, D, C, sigma, log_sigma, stream);
}

int cnf_encoder_forward_bwd_cpl(const int64_t* categ, const float* eps, const float* table,
                                const float* category_prior, const float* pad, float beta, const float* class_prob_log,
                                const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream) {
    return encoder_bwd_dispatch("cnf_encoder_forward_bwd_cpl", categ, eps, table, category_prior, pad, beta, class_prob_log, g_zout, g_ldj,
                                g_table, workspace, B, N, D, C, sigma, log_sigma, stream);
}

static int encoder_bwd_dispatch(const char* what, const int64_t* categ, const float* eps, const float* table,
                                const float* category_prior, const float* pad, float beta, const float* class_prob_log,
                                const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream) {
    CNF_REQUIRE(categ && eps && table && category_prior && g_table && workspace, "%s: null tensor", what);
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0 && D <= kEncMaxD, "%s: bad shape", what);
    if (B == 0) {       // the forward accepts an empty batch; its gradient is a zero table
        cnf::zero_fill_async(g_table, (size_t)C * 2 * D * sizeof(float), (hipStream_t)stream);
        return launch_status(what);
    }
    EncBwdTiledArgs b = {};
    b.categ = categ; b.eps = eps; b.table = table; b.prior = category_prior; b.pad = pad;
    b.g_zout = g_zout; b.g_ldj = g_ldj; b.g_table = g_table;
    b.ntok = (long)B * N; b.N = N; b.D = D; b.C = C; b.beta = beta; b.sigma = sigma; b.log_sigma = log_sigma;
    hipStream_t st = (hipStream_t)stream;
    const int which = g_enc_bwd_kernel.load(std::memory_order_relaxed);
    const long P = (long)C * 2 * D;
    // which: 0 = by shape, 1 = the two passes, 2 / 3 = the narrow / wide pair workgroup wherever its lanes hold the classes
    int pw = 0;
    if (which == 2 && pair_shape_ok(b.ntok, D, C, kPairMaxC)) pw = kPairWaves;
    else if (which == 3 && pair_shape_ok(b.ntok, D, C)) pw = kPairWavesWide;
    else if (which == 0 && pair_shape_ok(b.ntok, D, C)) pw = pair_kernel_choice(b.ntok, C, class_prob_log != nullptr);
    if (pw) {                                           // ... if its stage fits 64 KB of LDS at this class count
        PairsGeom fit;
        if (make_pairs_geom(b.ntok, D, C, kPairTokens, pw * 64, fit) > 64 * 1024) pw = 0;
    }
    if (pw) {
        // the pairs walked once with the token denominators known: the forward's class_prob_log, or a pre-pass that repeats
        // the forward's density sum
        b.partials = workspace;
        const float* cpl = class_prob_log;
        if (!cpl) {
            float* own_cpl = workspace + (size_t)kPairMaxWgs * P;
            const size_t lds_c = (size_t)C * (2 * D + 2) * sizeof(float);
            const int grid_c = (int)std::min<long>((b.ntok + kBlock - 1) / kBlock, 256 * 8);
            DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_cpl_kernel<(DT > 0 ? DT : 1)>), dim3(grid_c), dim3(kBlock), lds_c, st, b, own_cpl));
            cpl = own_cpl;
        }
        bool done = false;
        if (pw == kPairWavesWide) { DISPATCH_D(D, done = (launch_pairs<(DT > 0 ? DT : 1), kPairTokens, kPairWavesWide>(b, cpl, D, C, st))); }
        else { DISPATCH_D(D, done = (launch_pairs<(DT > 0 ? DT : 1), kPairTokens, kPairWaves>(b, cpl, D, C, st))); }
        if (done) {
            CNF_LAUNCH(encoder_bwd_splits64_kernel, dim3((unsigned)((P + 3) / 4)), dim3(kBlock), 0, st, (const float*)b.partials, b.S, P, g_table);
            return launch_status(what);
        }
    }
    b.S = bwd_tiled_splits(b.ntok, C);
    b.rec = workspace;
    b.partials = workspace + (size_t)b.ntok * (3 * D + 3);
    // LDS of the token lanes: the class chunk (at most 16 KB here, so that four workgroups share a CU next to the record
    // stage) + the record stage of 256 tokens; D > 8 (records of up to 51 floats) stores its records directly
    const bool stage_recs = D <= 8;
    const size_t stage_bytes = stage_recs ? (size_t)kBlock * (3 * D + 3) * sizeof(float) : 0;
    const int CC = std::min(C, std::max(1, tiled_chunk_classes(D) / (stage_recs ? 2 : 1)));
    const size_t smem_a = (size_t)CC * (2 * D + 2) * sizeof(float) + stage_bytes;
    const int grid_a = (int)std::min<long>((b.ntok + kBlock - 1) / kBlock, 256 * 8);
    const int KS = tiled_class_splits(C);
    float* part = b.partials + (size_t)b.S * C * 2 * D;
    const int sr = stage_recs ? 1 : 0;
    if (KS == 1 && C <= CC) {
        DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_token_kernel<DT, 0, true>), dim3(grid_a), dim3(kBlock), smem_a, st, b, CC, (float*)nullptr, 1, sr));
    } else if (KS == 1) {
        DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_token_kernel<DT, 0, false>), dim3(grid_a), dim3(kBlock), smem_a, st, b, CC, (float*)nullptr, 1, sr));
    } else {
        DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_token_kernel<DT, 1, false>), dim3(grid_a, KS), dim3(kBlock), smem_a, st, b, CC, part, KS, 0));
        DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_token_kernel<DT, 2, false>), dim3(grid_a), dim3(kBlock), stage_bytes, st, b, 0, part, KS, sr));
    }
    const int sh = bwd_class_shift(C);
    const size_t smem_b = std::max((size_t)2 * enc_bwd_stage(D <= 8 && (D <= 4 || D == 6 || D == 8) ? D : 0) * (3 * D + 3), (size_t)(sh < 8 ? kBlock * 2 * D : 0)) * sizeof(float);
    const dim3 grid_b((C + (1 << sh) - 1) >> sh, b.S);
    DISPATCH_D(D, CNF_LAUNCH((encoder_bwd_class_kernel<DT>), grid_b, dim3(kBlock), smem_b, st, b, sh));
    CNF_LAUNCH(encoder_bwd_splits_kernel, dim3((unsigned)((P + 15) / 16)), dim3(kBlock), 0, st,
               (const float*)b.partials, b.S, P, g_table);
    return launch_status(what);
}

}  // extern "C"
