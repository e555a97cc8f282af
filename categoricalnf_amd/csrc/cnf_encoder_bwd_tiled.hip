// Class-tiled backward of the mixture-model categorical encoder (any vocabulary size): d loss / d table [C, 2D] of
// cnf_encoder_forward / cnf_encoder_forward_tiled (linear_encoding.py:59-106,153-174).  Split from cnf_encoder.hip, whose
// forward / decode loops are built without the SLP vectoriser (Makefile).
#include "cnf_encoder.h"

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
// Notation as in cnf_encoder_bwd.hip: G = d loss / d ldj_tok (x pad), v_j the class scores, q = softmax(v).
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

int64_t cnf_encoder_bwd_tiled_workspace_floats(int B, int N, int D, int C) {
    const long ntok = (long)B * N;
    const int ks = tiled_class_splits(C);
    return (int64_t)ntok * (3 * D + 3) + (int64_t)bwd_tiled_splits(ntok, C) * C * 2 * D + (ks > 1 ? (int64_t)ks * ntok * (2 + D) : 0);
}

int cnf_encoder_forward_bwd_tiled(const int64_t* categ, const float* eps, const float* table,
                                  const float* category_prior, const float* pad, float beta,
                                  const float* g_zout, const float* g_ldj, float* g_table, float* workspace,
                                  int B, int N, int D, int C, float sigma, float log_sigma, cnf_stream_t stream) {
    CNF_REQUIRE(categ && eps && table && category_prior && g_table && workspace, "cnf_encoder_forward_bwd_tiled: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && C > 0 && D <= kEncMaxD, "cnf_encoder_forward_bwd_tiled: bad shape");
    if (B == 0) {       // the forward accepts an empty batch; its gradient is a zero table
        cnf::zero_fill_async(g_table, (size_t)C * 2 * D * sizeof(float), (hipStream_t)stream);
        return launch_status("cnf_encoder_forward_bwd_tiled");
    }
    EncBwdTiledArgs b = {};
    b.categ = categ; b.eps = eps; b.table = table; b.prior = category_prior; b.pad = pad;
    b.g_zout = g_zout; b.g_ldj = g_ldj; b.g_table = g_table;
    b.ntok = (long)B * N; b.N = N; b.D = D; b.C = C; b.beta = beta; b.sigma = sigma; b.log_sigma = log_sigma;
    b.S = bwd_tiled_splits(b.ntok, C);
    b.rec = workspace;
    b.partials = workspace + (size_t)b.ntok * (3 * D + 3);
    hipStream_t st = (hipStream_t)stream;
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
    const long P = (long)C * 2 * D;
    CNF_LAUNCH(encoder_bwd_splits_kernel, dim3((unsigned)((P + 15) / 16)), dim3(kBlock), 0, st,
               (const float*)b.partials, b.S, P, g_table);
    return launch_status("cnf_encoder_forward_bwd_tiled");
}

}  // extern "C"
