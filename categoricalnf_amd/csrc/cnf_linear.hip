// ActNorm (per-channel affine) and the invertible 1x1 convolution: 8 B/elem streaming kernels whose
// log-det is analytic (no reduction over elements), plus the data-dependent-init statistics.
#include "cnf_common.h"

#include <atomic>

#include <algorithm>

namespace cnf {

constexpr int kMaxD = 64;   // channels staged in LDS

struct ActNormArgs {
    const float* z;
    const float* bias;
    const float* scales;
    const float* pad;
    const float* length;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int B, N, D, reverse;
    long total;         // B*N*D
};

// activation_normalization.py:24-48.  Flat float4 streaming; channel = element index mod D.
template <int VEC>
__global__ __launch_bounds__(kBlock) void actnorm_kernel(ActNormArgs a) {
    __shared__ float sb[kMaxD], se[kMaxD];
    __shared__ float ssum;
    for (int d = threadIdx.x; d < a.D; d += kBlock) {
        sb[d] = a.bias[d];
        se[d] = a.reverse ? expf(-a.scales[d]) : expf(a.scales[d]);
    }
    if (threadIdx.x == 0) {
        float s = 0.f;
        for (int d = 0; d < a.D; ++d) s += a.scales[d];   // scales.sum(dim=[1,2])
        ssum = a.reverse ? -s : s;
    }
    __syncthreads();

    bool bad = false;
    const long nvec = a.total / VEC;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < nvec; i += (long)gridDim.x * kBlock) {
        const long e0 = i * VEC;
        float v[VEC];
        if (VEC == 4) {
            const float4 q = *reinterpret_cast<const float4*>(a.z + e0);
            v[0] = q.x; v[1] = q.y; v[VEC > 2 ? 2 : 0] = q.z; v[VEC > 3 ? 3 : 0] = q.w;
        } else {
            for (int j = 0; j < VEC; ++j) v[j] = a.z[e0 + j];
        }
        long tok = e0 / a.D;
        int d = (int)(e0 - tok * a.D);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float o = a.reverse ? v[j] * se[d] - sb[d] : (v[j] + sb[d]) * se[d];
            if (a.pad) o = o * a.pad[tok];
            bad |= isnan(o);
            v[j] = o;
            if (++d == a.D) {
                d = 0;
                ++tok;
            }
        }
        if (VEC == 4) {
            *reinterpret_cast<float4*>(a.z_out + e0) = make_float4(v[0], v[1], v[VEC > 2 ? 2 : 0], v[VEC > 3 ? 3 : 0]);
        } else {
            for (int j = 0; j < VEC; ++j) a.z_out[e0 + j] = v[j];
        }
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);

    // analytic log-det: one thread per sample
    for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
        float len;
        if (a.length) {
            len = a.length[b];
        } else if (a.pad) {
            len = 0.f;
            for (int n = 0; n < a.N; ++n) len += a.pad[b * a.N + n];
        } else {
            len = (float)a.N;
        }
        const float v = (a.ldj_in ? a.ldj_in[b] : 0.f) + ssum * len;
        a.ldj_out[b] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    }
}

// activation_normalization.py:55-67 statistics (fp64 accumulation, one atomic per wave and channel)
__global__ __launch_bounds__(kBlock) void actnorm_stats_kernel(const float* z, const float* pad,
                                                               const double* mean, double* out,
                                                               long ntok, int D, int pass) {
    // each thread owns channel d = threadIdx.x % D of tokens strided by the block; D <= kMaxD
    const int lanes_per_tok = D;
    const int toks_per_block = kBlock / lanes_per_tok;
    const int slot = threadIdx.x / lanes_per_tok, d = threadIdx.x % lanes_per_tok;
    double acc = 0.0, cnt = 0.0;
    if (slot < toks_per_block) {
        const double m = pass == 1 ? mean[d] : 0.0;
        for (long t = (long)blockIdx.x * toks_per_block + slot; t < ntok; t += (long)gridDim.x * toks_per_block) {
            const double w = pad ? (double)pad[t] : 1.0;
            const double x = (double)z[t * D + d];
            acc += pass == 0 ? x * w : (x - m) * (x - m) * w;
            cnt += w;
        }
    }
    __shared__ double sacc[kBlock], scnt[kBlock];
    sacc[threadIdx.x] = acc;
    scnt[threadIdx.x] = cnt;
    __syncthreads();
    if (threadIdx.x < D) {
        double s = 0.0, c = 0.0;
        for (int k = 0; k < toks_per_block; ++k) {
            s += sacc[k * D + threadIdx.x];
            c += scnt[k * D + threadIdx.x];
        }
        atomicAdd(&out[threadIdx.x], s);
        if (pass == 0 && threadIdx.x == 0) atomicAdd(&out[D], c);
    }
}

struct ConvArgs {
    const float* x;
    const float* w;      // [D,D] row-major, z' = x @ W
    const float* sldj;   // device scalar
    const float* pad;
    const float* length;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int B, N, D, reverse;
    long ntok;
};

// ActNorm followed by the 1x1 convolution (the first two layers of every flow step of the reference's models),
// one pass instead of two: forward  z' = (((z + b) e^{sc}) pad) @ W pad ; reverse  z' = (((z @ W^-1) pad) e^{-sc} - b) pad.
// The arithmetic is the two kernels' arithmetic in the same order, so results are identical to running them
// back to back; only the intermediate [B,N,D] round trip through HBM (8 B/elem) disappears.
struct ActConvArgs {
    const float* z;
    const float* bias;
    const float* scales;
    const float* w;
    const float* sldj;
    const float* pad;
    const float* length;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    int* flags;
    int B, N, D, reverse;
    long ntok;
};
// One lane owns TP consecutive tokens so that its TP*D floats are whole 16-byte vectors (D = 6: two tokens = three
// float4): 16-byte loads and nontemporal 16-byte stores, every byte of a line used by neighbouring lanes.  The
// per-channel constants (bias, e^{+-scales}, W) are read ONCE into registers before the loop — inside it the
// compiler could not keep them, the stores to z_out may alias them.
typedef float ac_f4 __attribute__((ext_vector_type(4)));
// ACT / CONV: round 4 — the same kernel with one of the two layers compiled out IS the stand-alone ActNorm / 1x1 convolution for
// D in {1..6, 8} (cnf_actnorm, cnf_invconv): their own kernels (a flat float4 stream with a 64-bit division per vector; one token per
// lane at a D*4-byte stride) took 12.5 / 12.3 us at the benchmark shape, this one 9.7 for both layers together.
template <int D, bool ACT = true, bool CONV = true>
__global__ __launch_bounds__(kBlock) void actnorm_invconv_kernel(ActConvArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    float wk[CONV ? D * D : 1], bk[D], ek[D];
    if (CONV) {
#pragma unroll
        for (int i = 0; i < D * D; ++i) wk[i] = a.w[i];
    }
    if (ACT) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            bk[i] = a.bias[i];
            ek[i] = expf(a.reverse ? -a.scales[i] : a.scales[i]);
        }
    }
    bool bad = false;
    auto token = [&](const float* xin, float* out, float p) {
        float xv[D];
#pragma unroll
        for (int i = 0; i < D; ++i) xv[i] = xin[i];
        if (ACT && !a.reverse) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float y = (xv[i] + bk[i]) * ek[i];
                if (a.pad) y = y * p;
                xv[i] = y;
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) {
            float acc;
            if (CONV) {
                acc = 0.f;
#pragma unroll
                for (int i = 0; i < D; ++i) acc = fmaf(xv[i], wk[i * D + j], acc);
                if (a.pad) acc = acc * p;
            } else {
                acc = xv[j];
            }
            if (ACT && a.reverse) {
                acc = acc * ek[j] - bk[j];
                if (a.pad) acc = acc * p;
            }
            bad |= isnan(acc);
            out[j] = acc;
        }
    };
    const bool aligned = ((reinterpret_cast<uintptr_t>(a.z) | reinterpret_cast<uintptr_t>(a.z_out)) & 15) == 0;
    const long ngroups = aligned ? a.ntok / TP : 0;
    // Wave tiles of 64 groups: the tile is ONE contiguous span of 64*NV 16-byte vectors, loaded fully coalesced (lane i
    // takes vectors i, i+64, ...) into the wave's LDS strip, read back group-wise (lane i owns floats [i*TP*D, ...):
    // stride TP*D words, conflict-free for the 128-bit reads), computed, and written out the same way in reverse.
    // (Per-lane 16-byte loads at a TP*D*4-byte stride, the previous form, cost 12.7 us at the benchmark shape.)
    __shared__ ac_f4 strip_all[kWavesPerBlock][kWave * NV];
    ac_f4* strip = strip_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const ac_f4* src = reinterpret_cast<const ac_f4*>(a.z + tile * (kWave * TP * D));
        ac_f4 q[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) q[v] = src[v * kWave + lane];
#pragma unroll
        for (int v = 0; v < NV; ++v) strip[v * kWave + lane] = q[v];
        wave_lds_sync();
        float xin[TP * D], out[TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const ac_f4 r = strip[lane * NV + v];
            xin[4 * v] = r.x; xin[4 * v + 1] = r.y; xin[4 * v + 2] = r.z; xin[4 * v + 3] = r.w;
        }
        const long g = tile * kWave + lane;
#pragma unroll
        for (int k = 0; k < TP; ++k) token(xin + k * D, out + k * D, a.pad ? a.pad[g * TP + k] : 1.f);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const ac_f4 r = {out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
            strip[lane * NV + v] = r;               // the lane's own group: nobody else reads these words
        }
        wave_lds_sync();
        ac_f4* dst = reinterpret_cast<ac_f4*>(a.z_out + tile * (kWave * TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) __builtin_nontemporal_store(strip[v * kWave + lane], dst + v * kWave + lane);
        wave_lds_sync();                            // the strip is refilled by the next tile
    }
    // groups that do not fill a wave tile: one group per lane with direct 16-byte I/O
    for (long g = ntiles * kWave + (long)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (long)gridDim.x * kBlock) {
        float xin[TP * D], out[TP * D];
        const ac_f4* src = reinterpret_cast<const ac_f4*>(a.z + g * (TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const ac_f4 r = src[v];
            xin[4 * v] = r.x; xin[4 * v + 1] = r.y; xin[4 * v + 2] = r.z; xin[4 * v + 3] = r.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) token(xin + k * D, out + k * D, a.pad ? a.pad[g * TP + k] : 1.f);
        ac_f4* dst = reinterpret_cast<ac_f4*>(a.z_out + g * (TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const ac_f4 r = {out[4 * v], out[4 * v + 1], out[4 * v + 2], out[4 * v + 3]};
            dst[v] = r;
        }
    }
    // tokens that do not fill a group (or everything, for unaligned tensors): one token per lane
    for (long t = ngroups * TP + (long)blockIdx.x * kBlock + threadIdx.x; t < a.ntok; t += (long)gridDim.x * kBlock) {
        float xin[D], out[D];
#pragma unroll
        for (int i = 0; i < D; ++i) xin[i] = a.z[t * D + i];
        token(xin, out, a.pad ? a.pad[t] : 1.f);
#pragma unroll
        for (int i = 0; i < D; ++i) a.z_out[t * D + i] = out[i];
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    // log-det of both layers: ActNorm uses length | sum(pad) | N, the convolution length | N
    float ssum = 0.f;
    if (ACT) {
#pragma unroll
        for (int i = 0; i < D; ++i) ssum += a.scales[i];
    }
    const float sl = CONV ? a.sldj[0] : 0.f;
    for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
        float len_a, len_c;
        if (a.length) {
            len_a = len_c = a.length[b];
        } else {
            len_c = (float)a.N;
            if (a.pad) {
                len_a = 0.f;
                for (int n = 0; n < a.N; ++n) len_a += a.pad[b * a.N + n];
            } else len_a = (float)a.N;
        }
        const float base = a.ldj_in ? a.ldj_in[b] : 0.f;
        // same association as the two layers run in sequence
        float v;
        if (ACT && CONV) v = a.reverse ? (base - sl * len_c) + (-ssum) * len_a : (base + ssum * len_a) + sl * len_c;
        else if (ACT) v = base + (a.reverse ? -ssum : ssum) * len_a;                   // the stand-alone kernels' expressions
        else v = a.reverse ? base - sl * len_c : base + sl * len_c;
        a.ldj_out[b] = v;
        if (isnan(v)) raise_flag(a.flags, CNF_FLAG_NAN_LDJ);
    }
}

// ---- the LU-parametrised weight of the 1x1 convolution (permutation_layers.py:61-71) -------------------------------------------
// W = P (L o tril(-1) + I)(U o triu(1) + diag(sign_s e^log_s)), sldj = sum log_s — D x D parameter preparation that the reference
// (and rounds 1-3 here) assembles from ~11 tiny torch ops whose autograd adds ~15 more: 26 launches per flow step and training
// pass, 208 of the 283 launches of an 8-step affine flow's forward + backward at the benchmark shape (4.0 ms of host time for
// 1.9 ms of kernels, profiles/r04_flow_autograd_overhead.txt).  One workgroup, thread (r, c) owns entry [r][c]; D <= 16.
constexpr int kLuMax = 16;
__device__ __forceinline__ void lu_factors(const float* l, const float* u, const float* log_s, const float* sign_s, int D, int r, int c,
                                           float (*Lo)[kLuMax + 1], float (*Up)[kLuMax + 1]) {
    if (r < D && c < D) {
        Lo[r][c] = (c < r ? l[r * D + c] : 0.f) + (r == c ? 1.f : 0.f);
        Up[r][c] = (c > r ? u[r * D + c] : 0.f) + (r == c ? sign_s[r] * expf(log_s[r]) : 0.f);
    }
}
// w_inv_out (optional, D <= 8): W^-1 as cnf_actnorm_invconv_bwd would compute it in a launch of its own (small_inverse_wg:
// the same bits) — the backward of a fused group that kept only its OUTPUT takes it from here
__global__ __launch_bounds__(kLuMax * kLuMax) void lu_weight_kernel(const float* p, const float* l, const float* u, const float* log_s,
                                                                    const float* sign_s, float* w_out, float* sldj_out, float* w_inv_out, int D) {
    __shared__ float Lo[kLuMax][kLuMax + 1], Up[kLuMax][kLuMax + 1], M[kLuMax][kLuMax + 1];
    __shared__ float Wd[64];
    __shared__ double Ainv[8][17];
    const int r = threadIdx.x / kLuMax, c = threadIdx.x % kLuMax;
    const bool live = r < D && c < D;
    lu_factors(l, u, log_s, sign_s, D, r, c, Lo, Up);
    __syncthreads();
    if (live) {
        float m = 0.f;
        for (int k = 0; k < D; ++k) m = fmaf(Lo[r][k], Up[k][c], m);
        M[r][c] = m;
    }
    __syncthreads();
    if (live) {
        float w = 0.f;
        for (int k = 0; k < D; ++k) w = fmaf(p[r * D + k], M[k][c], w);
        w_out[r * D + c] = w;
        if (w_inv_out) Wd[r * D + c] = w;
    }
    if (threadIdx.x == 0) {
        float t = 0.f;
        for (int i = 0; i < D; ++i) t += log_s[i];
        sldj_out[0] = t;
    }
    if (w_inv_out) {                        // workgroup-uniform
        __syncthreads();
        small_inverse_wg(Wd, w_inv_out, D, threadIdx.x, Ainv);
    }
}
// g_l = (P^T g_W U^T) o tril(-1), g_u = (L^T P^T g_W) o triu(1), g_log_s[i] = (L^T P^T g_W)[i][i] sign_s[i] e^log_s[i] + g_sldj
__global__ __launch_bounds__(kLuMax * kLuMax) void lu_weight_bwd_kernel(const float* p, const float* l, const float* u, const float* log_s,
                                                                        const float* sign_s, const float* g_w, const float* g_sldj,
                                                                        float* g_l, float* g_u, float* g_log_s, int D) {
    __shared__ float Lo[kLuMax][kLuMax + 1], Up[kLuMax][kLuMax + 1], G[kLuMax][kLuMax + 1];
    const int r = threadIdx.x / kLuMax, c = threadIdx.x % kLuMax;
    const bool live = r < D && c < D;
    lu_factors(l, u, log_s, sign_s, D, r, c, Lo, Up);
    if (live) {
        float g = 0.f;                                    // (P^T g_W)[r][c]
        if (g_w)
            for (int k = 0; k < D; ++k) g = fmaf(p[k * D + r], g_w[k * D + c], g);
        G[r][c] = g;
    }
    __syncthreads();
    if (live) {
        float gl = 0.f, gu = 0.f;
        for (int k = 0; k < D; ++k) {
            gl = fmaf(G[r][k], Up[c][k], gl);             // (G U^T)[r][c]
            gu = fmaf(Lo[k][r], G[k][c], gu);             // (L^T G)[r][c]
        }
        g_l[r * D + c] = c < r ? gl : 0.f;
        g_u[r * D + c] = c > r ? gu : 0.f;
        if (r == c) g_log_s[r] = gu * (sign_s[r] * expf(log_s[r])) + (g_sldj ? g_sldj[0] : 0.f);
    }
}

// any D: one lane per output element
__global__ __launch_bounds__(kBlock) void invconv_generic_kernel(ConvArgs a) {
    const int D = a.D;
    bool bad = false;
    const long total = a.ntok * D;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const long t = e / D;
        const int j = (int)(e - t * D);
        float acc = 0.f;
        for (int i = 0; i < D; ++i) acc = fmaf(a.x[t * D + i], a.w[i * D + j], acc);
        if (a.pad) acc = acc * a.pad[t];
        bad |= isnan(acc);
        a.z_out[e] = acc;
    }
    if (bad) raise_flag(a.flags, CNF_FLAG_NAN_Z);
    const float sl = a.sldj[0];
    for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
        const float s = sl * (a.length ? a.length[b] : (float)a.N);
        const float base = a.ldj_in ? a.ldj_in[b] : 0.f;
        a.ldj_out[b] = a.reverse ? base - s : base + s;
    }
}

// one vector per lane (no grid-stride loop): every load of the launch can be in flight at once, which is
// what keeps HBM busy for these single-pass 8 B/elem kernels (measured against a 2048-workgroup cap)
static inline int stream_grid(long n) {
    const long blocks = (n + kBlock - 1) / kBlock;
    return (int)std::min<long>(std::max<long>(blocks, 1), 1 << 22);
}

}  // namespace cnf

using namespace cnf;

// 1 (default): cnf_actnorm / cnf_invconv take the fused pair's token-owner kernel for D in {1..6, 8}; 0 = their own older kernels (A/B, tests)

// launches actnorm_invconv_kernel<D, ACT, CONV> for D in {1..6, 8}; false = another D (the callers keep their generic kernels)
template <bool ACT, bool CONV>
static bool launch_act_conv(const ActConvArgs& a, hipStream_t st) {
    const int D = a.D;
    // one lane per group of 1, 2 or 4 tokens (16-byte I/O); uncapped grid: every lane makes a single trip
    const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    const long lanes = std::max<long>((a.ntok + tp - 1) / tp, 1);      // one group per lane, 64 groups per wave tile
    const dim3 grid((unsigned)std::min<long>((lanes + kBlock - 1) / kBlock, 1 << 22)), block(kBlock);
    switch (D) {
        case 1: CNF_LAUNCH((actnorm_invconv_kernel<1, ACT, CONV>), grid, block, 0, st, a); break;
        case 2: CNF_LAUNCH((actnorm_invconv_kernel<2, ACT, CONV>), grid, block, 0, st, a); break;
        case 3: CNF_LAUNCH((actnorm_invconv_kernel<3, ACT, CONV>), grid, block, 0, st, a); break;
        case 4: CNF_LAUNCH((actnorm_invconv_kernel<4, ACT, CONV>), grid, block, 0, st, a); break;
        case 5: CNF_LAUNCH((actnorm_invconv_kernel<5, ACT, CONV>), grid, block, 0, st, a); break;
        case 6: CNF_LAUNCH((actnorm_invconv_kernel<6, ACT, CONV>), grid, block, 0, st, a); break;
        case 8: CNF_LAUNCH((actnorm_invconv_kernel<8, ACT, CONV>), grid, block, 0, st, a); break;
        default: return false;
    }
    return true;
}

extern "C" {


int cnf_actnorm(const float* z, const float* bias, const float* scales,
                const float* pad, const float* length,
                const float* ldj_in, float* z_out, float* ldj_out,
                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && bias && scales && z_out && ldj_out, "cnf_actnorm: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && D <= kMaxD, "cnf_actnorm: bad shape B=%d N=%d D=%d (D<=%d)", B, N, D, kMaxD);
    if (B == 0) return CNF_OK;
    {   // D in {1..6, 8}: the token-owner kernel of the fused pair with the convolution compiled out
        ActConvArgs f{z, bias, scales, nullptr, nullptr, pad, length, ldj_in, z_out, ldj_out, flags, B, N, D, reverse, (long)B * N};
        if (launch_act_conv<true, false>(f, (hipStream_t)stream)) return launch_status("cnf_actnorm");
    }
    ActNormArgs a{z, bias, scales, pad, length, ldj_in, z_out, ldj_out, flags, B, N, D, reverse, (long)B * N * D};
    const long work = std::max<long>(a.total / 4, B);
    if (a.total % 4 == 0)
        CNF_LAUNCH((actnorm_kernel<4>), dim3(stream_grid(work)), dim3(kBlock), 0, (hipStream_t)stream, a);
    else
        CNF_LAUNCH((actnorm_kernel<1>), dim3(stream_grid(a.total)), dim3(kBlock), 0, (hipStream_t)stream, a);
    return launch_status("cnf_actnorm");
}

int cnf_actnorm_stats(const float* z, const float* pad, const double* mean, double* out,
                      int B, int N, int D, int pass, cnf_stream_t stream) {
    CNF_REQUIRE(z && out && (pass == 0 || mean), "cnf_actnorm_stats: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && D <= kMaxD, "cnf_actnorm_stats: bad shape");
    const long ntok = (long)B * N;
    const int tpb = kBlock / D;
    const int grid = (int)std::min<long>((ntok + tpb - 1) / tpb, 1024);
    CNF_LAUNCH(actnorm_stats_kernel, dim3(std::max(grid, 1)), dim3(kBlock), 0, (hipStream_t)stream,
                       z, pad, mean, out, ntok, D, pass);
    return launch_status("cnf_actnorm_stats");
}

int cnf_invconv(const float* x, const float* weight, const float* sldj,
                const float* pad, const float* length,
                const float* ldj_in, float* z_out, float* ldj_out,
                int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(x && weight && sldj && z_out && ldj_out, "cnf_invconv: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_invconv: bad shape");
    if (B == 0) return CNF_OK;
    hipStream_t st = (hipStream_t)stream;
    {   // D in {1..6, 8}: the token-owner kernel of the fused pair with ActNorm compiled out
        ActConvArgs f{x, nullptr, nullptr, weight, sldj, pad, length, ldj_in, z_out, ldj_out, flags, B, N, D, reverse, (long)B * N};
        if (launch_act_conv<false, true>(f, st)) return launch_status("cnf_invconv");
    }
    ConvArgs a{x, weight, sldj, pad, length, ldj_in, z_out, ldj_out, flags, B, N, D, reverse, (long)B * N};
    // every other D: the generic kernel (one lane per output element)
    CNF_LAUNCH(invconv_generic_kernel, dim3(stream_grid(a.ntok * D)), dim3(kBlock), 0, st, a);
    return launch_status("cnf_invconv");
}

int cnf_invconv_lu_weight(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                          float* weight_out, float* sldj_out, int D, cnf_stream_t stream) {
    CNF_REQUIRE(p && l && u && log_s && sign_s && weight_out && sldj_out, "cnf_invconv_lu_weight: null tensor");
    CNF_REQUIRE(D > 0, "cnf_invconv_lu_weight: bad shape");
    if (D > kLuMax) {
        set_error("cnf_invconv_lu_weight: built for D <= %d (got %d)", kLuMax, D);
        return CNF_ERR_UNSUPPORTED;
    }
    CNF_LAUNCH(lu_weight_kernel, dim3(1), dim3(kLuMax * kLuMax), 0, (hipStream_t)stream, p, l, u, log_s, sign_s, weight_out, sldj_out, (float*)nullptr, D);
    return launch_status("cnf_invconv_lu_weight");
}

int cnf_invconv_lu_weight_inv(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                              float* weight_out, float* sldj_out, float* weight_inv_out, int D, cnf_stream_t stream) {
    CNF_REQUIRE(p && l && u && log_s && sign_s && weight_out && sldj_out && weight_inv_out, "cnf_invconv_lu_weight_inv: null tensor");
    CNF_REQUIRE(D > 0, "cnf_invconv_lu_weight_inv: bad shape");
    if (D > 8) {
        set_error("cnf_invconv_lu_weight_inv: the inverse is built for D <= 8 (got %d)", D);
        return CNF_ERR_UNSUPPORTED;
    }
    CNF_LAUNCH(lu_weight_kernel, dim3(1), dim3(kLuMax * kLuMax), 0, (hipStream_t)stream, p, l, u, log_s, sign_s, weight_out, sldj_out, weight_inv_out, D);
    return launch_status("cnf_invconv_lu_weight_inv");
}

int cnf_invconv_lu_weight_bwd(const float* p, const float* l, const float* u, const float* log_s, const float* sign_s,
                              const float* g_weight, const float* g_sldj, float* g_l, float* g_u, float* g_log_s, int D,
                              cnf_stream_t stream) {
    CNF_REQUIRE(p && l && u && log_s && sign_s && g_l && g_u && g_log_s, "cnf_invconv_lu_weight_bwd: null tensor");
    CNF_REQUIRE(D > 0, "cnf_invconv_lu_weight_bwd: bad shape");
    if (D > kLuMax) {
        set_error("cnf_invconv_lu_weight_bwd: built for D <= %d (got %d)", kLuMax, D);
        return CNF_ERR_UNSUPPORTED;
    }
    CNF_LAUNCH(lu_weight_bwd_kernel, dim3(1), dim3(kLuMax * kLuMax), 0, (hipStream_t)stream, p, l, u, log_s, sign_s, g_weight, g_sldj,
               g_l, g_u, g_log_s, D);
    return launch_status("cnf_invconv_lu_weight_bwd");
}

int cnf_actnorm_invconv(const float* z, const float* bias, const float* scales, const float* weight,
                        const float* sldj, const float* pad, const float* length,
                        const float* ldj_in, float* z_out, float* ldj_out,
                        int B, int N, int D, int reverse, int* flags, cnf_stream_t stream) {
    CNF_REQUIRE(z && bias && scales && weight && sldj && z_out && ldj_out, "cnf_actnorm_invconv: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_actnorm_invconv: bad shape");
    if (B == 0) return CNF_OK;
    ActConvArgs a{z, bias, scales, weight, sldj, pad, length, ldj_in, z_out, ldj_out, flags, B, N, D, reverse, (long)B * N};
    if (!launch_act_conv<true, true>(a, (hipStream_t)stream)) {
        set_error("cnf_actnorm_invconv: fused kernel is built for D in {1,2,3,4,5,6,8}; run the two layers separately for D=%d", D);
        return CNF_ERR_UNSUPPORTED;
    }
    return launch_status("cnf_actnorm_invconv");
}

}  // extern "C"
