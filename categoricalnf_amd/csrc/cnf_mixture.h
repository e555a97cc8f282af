// Shared by the mixture-CDF coupling kernels (cnf_mixture.hip: fp64 kernels and the C entry points;
// cnf_mixture_tok.hip: the fp32 token-pass kernel on DMA-staged parameter rows).
#pragma once
#include "cnf_common.h"

namespace cnf {

constexpr int kMaxAct = 64;          // channels per token
constexpr double kLn10 = 2.302585092994045684;

struct MixArgs {
    // fused (module) form: fp32 z / nn_out
    const float* z;
    const float* nn;
    const float* sf;
    const float* msf;
    // split (static API) form: fp64 tensors
    const double* z64;
    const double* p_t;
    const double* p_log_s;
    const double* p_log_pi;
    const double* p_mu;
    const double* p_ls;
    const float* mask;
    const float* pad;
    const float* ldj_in;
    float* z_out;
    float* ldj_out;
    float* reg_out;
    double* z_out64;
    double* ldj_out64;
    double* reg_elem64;
    int* flags;
    int B, N, D, K, P, L;
    // layout of nn (and of g_nn in the backward): nn_D parameter blocks per token, the first one belonging to channel nn_c0.
    // Reference layout: nn_D = D, nn_c0 = 0 (blocks for ALL channels, half of them multiplied by the zero mask).  Compact layout
    // (cnf_mixture_coupling_compact*): nn_D = DA, nn_c0 = first transformed channel — blocks of the transformed channels only.
    int nn_D, nn_c0;
    int compact;            // entry point asked for the compact layout (launch_mixture derives nn_D / nn_c0 from the channel list)
    int mr, mc;
    int DA;                 // transformed channels per token (channel mask) or D (per-item test)
    unsigned long long act_bits;   // bit d set = channel d is transformed (a list cannot be indexed per lane)
    int per_item_mask;      // 1: mask varies along N (chess) -> test every item
    int cst_lds;            // inverse with run-time K: 1 = per-mixture constants cached in LDS
    int reverse, pad_in_transform, pad_output, use_reg;
    double reg_max, reg_factor;
    FastDiv div_d, div_da;
    FastDiv div_upi;        // fp32 forward kernel: staging units per item
    // token-pass kernel (cnf_mixture_tok.hip): NLL epilogue and the split-row workspace
    const float* length;    // [B] or null (= N)
    float* neglog_out;      // [B] or null
    float* nll_out;         // [B]; non-null selects the NLL epilogue
    long long* nll_acc;     // optional: 64 fixed-point batch-sum words (as cnf_affine_coupling_nll_acc)
    long long* ws_acc;      // [2B] fixed-point row sums, zero before and after every launch (or null)
    double* ws_big;         // [2B] fp64 escape words of the row sums (terms the fixed-point words cannot take), zero likewise
    int* ws_cnt;            // [B] arrival tickets, zero before and after every launch (or null)
    PriorConst prior;
    // optional epilogue of the forward kernel: ActNorm and 1x1 convolution of the NEXT flow step applied to z' before
    // it is written (cnf_mixture_coupling_actconv); e_w non-null selects it
    const float* e_bias;    // [D]
    const float* e_scales;  // [D]
    const float* e_w;       // [D, D], z'' = z' @ W
    const float* e_sldj;    // [1]
    const float* e_length;  // [B] or null: the `length` the two layers receive
};

// cnf_mixture_tok.hip
bool launch_mixture_tok(MixArgs& a, hipStream_t st, int force_g, bool x64 = false);
void set_mixture_split_waves(int w);
void set_mixture_whole_tokens(int on);
void set_mixture_nt_mb(int mb);
int mixture_nt_mb();
// cnf_mixture_tok_bwd.hip
bool launch_mixture_tok_bwd(MixArgs& a, const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                            float* g_sf, float* g_msf, float* workspace, hipStream_t st, int force_g);

__device__ __forceinline__ double safe_log(double x) { return log(fmax(x, 1e-22)); }

// F.softplus (beta 1, threshold 20) and F.logsigmoid in fp64, as torch evaluates them
__device__ __forceinline__ double softplus64(double x) { return x > 20.0 ? x : log1p(exp(x)); }

constexpr float kLog2eF = 1.4426950408889634f;
constexpr float kLn2F = 0.6931471805599453f;

struct BoundTab {       // tanh bound of one raw parameter: v -> f tanh(v / max(f,1)) = f - 2f / (2^{v x3} + 1)
    float x3, m2f, f;
};
__device__ __forceinline__ BoundTab make_bound(float raw_sf) {
    const float f = expf(raw_sf);
    BoundTab b;
    b.x3 = 2.8853900817779268f / fmaxf(f, 1.f);
    b.m2f = -2.f * f;
    b.f = f;
    return b;
}
__device__ __forceinline__ float apply_bound(float v, const BoundTab& b) {
    return fmaf(__builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v * b.x3) + 1.f), b.m2f, b.f);
}
// the same bound with the library tanh, as the fp64 kernels apply it (get_mixt_params :156-162: fp32, then widened)
__device__ __forceinline__ float apply_bound_exact(float v, const BoundTab& b) { return tanhf(v / fmaxf(b.f, 1.f)) * b.f; }

// G = lanes per item: 1, or 4 for a run-time K too large to stage 64 rows per wave (the language model's K = 51
// rows are 620 bytes): a pass then covers 16 items, lane g of an item takes the mixtures k = g (mod 4) and the four
// partial sums are combined by xor-shuffles (symmetric, so the four lanes stay bit-identical and take the same
// branches).
template <int G>
__device__ __forceinline__ float gsum(float v) {
    if (G >= 2) v += __shfl_xor(v, 1, kWave);
    if (G >= 4) v += __shfl_xor(v, 2, kWave);
    return v;
}
template <int G>
__device__ __forceinline__ float gmax(float v) {
    if (G >= 2) v = fmaxf(v, __shfl_xor(v, 1, kWave));
    if (G >= 4) v = fmaxf(v, __shfl_xor(v, 2, kWave));
    return v;
}
template <int G>
__device__ __forceinline__ float gmin(float v) {
    if (G >= 2) v = fminf(v, __shfl_xor(v, 1, kWave));
    if (G >= 4) v = fminf(v, __shfl_xor(v, 2, kWave));
    return v;
}

// The same reductions for groups of G consecutive lanes that start on a multiple of G (the token-pass kernels: lane =
// token * DA * G + channel * G + share): quad-permute DPP moves instead of ds_bpermute, i.e. no LDS round trip (the
// Newton iterations of the inverse reduce three sums per evaluation).  All lanes of a group are active together.
template <int CTRL>
__device__ __forceinline__ float quad_dpp(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, 0xf, 0xf, true));
}
constexpr int kQuadXor1 = 0xB1;     // quad_perm [1,0,3,2]
constexpr int kQuadXor2 = 0x4E;     // quad_perm [2,3,0,1]
template <int G>
__device__ __forceinline__ float qsum(float v) {
    static_assert(G == 1 || G == 2 || G == 4, "quad-permute reductions cover groups of 1, 2 or 4 lanes");
    if (G >= 2) v += quad_dpp<kQuadXor1>(v);
    if (G >= 4) v += quad_dpp<kQuadXor2>(v);
    return v;
}
template <int CTRL>
__device__ __forceinline__ double quad_dpp64(double v) {
    const long long b = __double_as_longlong(v);
    const int lo = __builtin_amdgcn_update_dpp(0, (int)b, CTRL, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_update_dpp(0, (int)(b >> 32), CTRL, 0xf, 0xf, true);
    return __longlong_as_double(((long long)hi << 32) | (unsigned int)lo);
}
template <int G>
__device__ __forceinline__ double qsum64(double v) {
    if (G >= 2) v += quad_dpp64<kQuadXor1>(v);
    if (G >= 4) v += quad_dpp64<kQuadXor2>(v);
    return v;
}
template <int G>
__device__ __forceinline__ float qmax(float v) {
    if (G >= 2) v = fmaxf(v, quad_dpp<kQuadXor1>(v));
    if (G >= 4) v = fmaxf(v, quad_dpp<kQuadXor2>(v));
    return v;
}
template <int G>
__device__ __forceinline__ float qmin(float v) {
    if (G >= 2) v = fminf(v, quad_dpp<kQuadXor1>(v));
    if (G >= 4) v = fminf(v, quad_dpp<kQuadXor2>(v));
    return v;
}

}  // namespace cnf
