// Shared host/device helpers for the gfx950 kernels of libcnf_hip.so.
// CDNA4 only: wave = 64 lanes, no CUDA compatibility paths.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/cnf_hip.h"
#include "../../include/cnf_tuning.h"

namespace cnf {

constexpr int kWave = 64;
constexpr int kBlock = 256;              // 4 waves per workgroup
constexpr int kWavesPerBlock = kBlock / kWave;
constexpr int kMaxTileChunks = 512;      // LDS partials one wave may own (floats)

// ---- host-side status ------------------------------------------------------------------------
void set_error(const char* fmt, ...);
int launch_status(const char* what);     // CNF_OK or CNF_ERR_LAUNCH (+ message)
int tile_chunks_target();
int unroll_target();
int math_mode();
int inverse_mode();
int mixture_tile_items();
// kernel timing (cnf_prof_arm / cnf_prof_collect): true = the launch that asked takes this event pair
bool prof_take(hipEvent_t* start, hipEvent_t* stop);

// Every kernel of the library is launched through this: an armed launch goes out with the dispatch's own start /
// stop timestamps bound to an event pair (hipExtLaunchKernelGGL; measured cost ~4 us of queue time per timed launch).
#define CNF_LAUNCH(kernel, grid, block, lds, st, ...)                                                        \
    do {                                                                                                     \
        hipEvent_t cnf_ps_, cnf_pe_;                                                                         \
        if (cnf::prof_take(&cnf_ps_, &cnf_pe_))                                                              \
            hipExtLaunchKernelGGL(kernel, grid, block, lds, st, cnf_ps_, cnf_pe_, 0, __VA_ARGS__);           \
        else                                                                                                 \
            hipLaunchKernelGGL(kernel, grid, block, lds, st, __VA_ARGS__);                                   \
    } while (0)

// Zero `bytes` bytes (a multiple of 4) at `p` with a KERNEL.  The library never issues hipMemsetAsync: on this stack a
// hipGraph memset node of 16 B ... 4 KiB writes garbage from the second launch of the graph on (tools/
// graph_linear_probe.py, profiles/r03_graph_train_root_cause.txt), and every entry point may be captured.
__global__ void cnf_zero_fill_kernel(uint32_t* p, size_t words);
inline void zero_fill_async(void* p, size_t bytes, hipStream_t st) {
    const size_t words = bytes / 4;
    if (words == 0) return;
    const unsigned grid = (unsigned)std::min<size_t>((words + 4 * 256 - 1) / (4 * 256), 4096);
    hipLaunchKernelGGL(cnf_zero_fill_kernel, dim3(grid), dim3(256), 0, st, reinterpret_cast<uint32_t*>(p), words);
}

#define CNF_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            cnf::set_error(__VA_ARGS__);      \
            return CNF_ERR_ARG;               \
        }                                     \
    } while (0)

// Exact floor(n / d) for n, d < 65536 with one v_mul_hi_u32.
struct FastDiv {
    uint32_t magic;
    uint32_t d;
};
inline FastDiv make_fastdiv(uint32_t d) {
    FastDiv f;
    f.d = d;
    f.magic = d <= 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d);
    return f;
}
__device__ __forceinline__ uint32_t fdiv(uint32_t n, const FastDiv f) {
    return f.d <= 1 ? n : __umulhi(n, f.magic);
}

// How the rows (= samples, L contiguous elements each) of a [B, L] tensor are dealt to waves:
// one wave owns `rw` consecutive rows per tile and walks them in chunks of VEC elements.
struct RowTiling {
    int B;          // rows
    int L;          // elements per row
    int vec;        // 4, 2 or 1 — largest of those dividing L, so a chunk never straddles rows
    int cpr;        // chunks per row = L / vec
    int rw;         // rows per wave-tile
    int p2;         // power of two >= rw, capped at 64 (row slots reduced per pass)
    int bpr;        // 1: small batch of long rows -> the four waves of a workgroup share ONE row (4x the waves)
    long ntiles;
    FastDiv div_cpr;
};
RowTiling make_row_tiling(int B, int L, int force_vec = 0, int target_chunks = 256, int gran = kWave);
inline dim3 tiling_grid(const RowTiling& t) {
    if (t.bpr) return dim3((unsigned)t.B);
    return dim3((unsigned)((t.ntiles + kWavesPerBlock - 1) / kWavesPerBlock));
}

// ---- device helpers ---------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// sum inside aligned groups of g lanes (g a power of two <= 64); every lane gets its group's sum
template <typename T>
__device__ __forceinline__ T group_sum(T v, int g) {
    for (int m = g >> 1; m >= 1; m >>= 1) v += __shfl_xor(v, m, kWave);
    return v;
}
// two sums carried through one row walk (log-det and prior log-prob of the fused coupling + NLL kernel)
struct Sum2 {
    float a, b;
    __device__ __forceinline__ Sum2() {}
    __device__ __forceinline__ Sum2(int) : a(0.f), b(0.f) {}
    __device__ __forceinline__ Sum2(float x, float y) : a(x), b(y) {}
    __device__ __forceinline__ Sum2& operator+=(const Sum2& o) {
        a += o.a;
        b += o.b;
        return *this;
    }
};
__device__ __forceinline__ Sum2 wave_sum(Sum2 v) { return Sum2(wave_sum(v.a), wave_sum(v.b)); }
__device__ __forceinline__ Sum2 group_sum(Sum2 v, int g) { return Sum2(group_sum(v.a, g), group_sum(v.b, g)); }
// make this wave's LDS writes visible to its own later reads (lanes exchange data through LDS)
__device__ __forceinline__ void wave_lds_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}
__device__ __forceinline__ void raise_flag(int* flags, int bit) {
    if (flags) atomicOr(flags, bit);
}

// Walk one wave-tile.  For each chunk of VEC elements starting at element e0 of `row`:
//   load_fn(row, e0) -> Data        issues the global loads (U chunks are issued back to back so
//                                   that U x 48 B per lane are in flight before the first use);
//   proc_fn(data, row, e0) -> T     computes, stores and returns the chunk's log-det contribution;
//   finish_fn(row, sum)             stores the row's sum.
// `part` is this wave's private LDS scratch (kMaxTileChunks values of T).
struct NoPre {
    __device__ __forceinline__ void operator()() const {}
};

// Logical workgroup index of the row walkers = the hardware's (round-robin over the 8 XCDs).  Giving each XCD one
// contiguous eighth of the workgroups instead (-DCNF_XCD_EIGHTHS) shortens the dispatch-timestamp duration of the affine
// forward + NLL kernel in a forward-only stream (20.2 -> 19.2 us) but changes neither its start-to-start time nor the
// bench step (36.0 us either way, interleaved A/B on one box): what shrinks is the overlap of a launch's timestamps
// with the drain of the launch before it, not the work.  Kept as a build option for A/B only.
__device__ __forceinline__ unsigned xcd_block(unsigned bid, unsigned nb) {
#if defined(CNF_XCD_EIGHTHS)
    const unsigned xcd = bid & 7u, idx = bid >> 3;
    const unsigned per = nb >> 3, rem = nb & 7u;
    return xcd * per + (xcd < rem ? xcd : rem) + idx;
#else
    (void)nb;
    return bid;
#endif
}
__device__ __forceinline__ unsigned walker_block() { return xcd_block(blockIdx.x, gridDim.x); }

// pre_fn() runs once per wave AFTER the first group of loads has been issued and before the first
// proc_fn: per-wave setup (LDS tables) hides behind the HBM latency of those loads.
template <int U, typename T, typename Data, bool PREFETCH = false, typename LoadFn, typename ProcFn, typename FinishFn,
          typename PreFn = NoPre>
__device__ __forceinline__ void walk_row_tile_split(const RowTiling& tl, T* part, LoadFn&& load_fn,
                                                    ProcFn&& proc_fn, FinishFn&& finish_fn,
                                                    PreFn&& pre_fn = NoPre()) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (tl.bpr) {
        // block-per-row mode (training-sized batches: B ~ 10^2 rows of 10^3 elements would otherwise
        // occupy B waves only): the workgroup's waves interleave over the chunks of one row and combine
        // their partial sums through LDS in wave order
        __shared__ T bpr_sh[kWavesPerBlock];
        const int row = blockIdx.x;
        T acc = 0;
        bool first = true;
        for (int c0 = threadIdx.x; c0 < tl.cpr || first; c0 += kBlock * U) {
            Data dat[U];
#pragma unroll
            for (int u = 0; u < U; ++u) dat[u] = load_fn(row, min(c0 + kBlock * u, tl.cpr - 1) * tl.vec);
            if (first) {
                pre_fn();
                first = false;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (c0 + kBlock * u < tl.cpr) acc += proc_fn(dat[u], row, (c0 + kBlock * u) * tl.vec);
        }
        acc = wave_sum(acc);
        if (lane == 0) bpr_sh[wave] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            T t = 0;
            for (int w = 0; w < kWavesPerBlock; ++w) t += bpr_sh[w];
            finish_fn(row, t);
        }
        return;
    }
    const long tile = (long)walker_block() * kWavesPerBlock + wave;
    if (tile >= tl.ntiles) return;
    const int row0 = (int)(tile * tl.rw);
    const int nrows = min(tl.rw, tl.B - row0);
    const int nch = nrows * tl.cpr;
    T acc1 = 0;
    if (PREFETCH && U == 1) {
        // software pipeline: the loads of chunk i+1 are in flight while chunk i is computed
        int c = lane;
        int r = tl.rw == 1 ? 0 : (int)fdiv((uint32_t)c, tl.div_cpr);
        int row = row0 + r, e0 = (c - r * tl.cpr) * tl.vec;
        Data cur;
        if (c < nch) cur = load_fn(row, e0);
        pre_fn();
        while (c < nch) {
            const int cn = c + kWave;
            const int rn = tl.rw == 1 ? 0 : (int)fdiv((uint32_t)cn, tl.div_cpr);
            const int rown = row0 + rn, en = (cn - rn * tl.cpr) * tl.vec;
            Data nxt;
            if (cn < nch) nxt = load_fn(rown, en);
            const T v = proc_fn(cur, row, e0);
            if (tl.rw == 1) acc1 += v;
            else part[c] = v;
            cur = nxt;
            c = cn; row = rown; e0 = en;
        }
    } else {
        // one group = U chunks per lane: all loads first, then (first group only) pre_fn, then compute
        auto group = [&](int c0, auto&& between) {
            Data dat[U];
            int rr[U], ee[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // lanes past the end of the tile re-load its last chunk (unconditional loads keep the
                // data out of control-flow phis, so no wait is forced before `between`)
                const int c = min(c0 + kWave * u, nch - 1);
                const int r = tl.rw == 1 ? 0 : (int)fdiv((uint32_t)c, tl.div_cpr);
                rr[u] = row0 + r;
                ee[u] = (c - r * tl.cpr) * tl.vec;
                dat[u] = load_fn(rr[u], ee[u]);
            }
            between();
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int c = c0 + kWave * u;
                if (c < nch) {
                    const T v = proc_fn(dat[u], rr[u], ee[u]);
                    if (tl.rw == 1) acc1 += v;
                    else part[c] = v;
                }
            }
        };
        group(lane, pre_fn);
        for (int c0 = lane + kWave * U; c0 < nch; c0 += kWave * U) group(c0, NoPre());
    }
    if (tl.rw == 1) {
        acc1 = wave_sum(acc1);
        if (lane == 0) finish_fn(row0, acc1);
        return;
    }
    wave_lds_sync();
    const int g = kWave / tl.p2;          // lanes cooperating on one row
    const int sub = lane & (g - 1);
    for (int r0 = 0; r0 < nrows; r0 += tl.p2) {
        const int r = r0 + lane / g;
        T acc = 0;
        if (r < nrows)
            for (int i = sub; i < tl.cpr; i += g) acc += part[r * tl.cpr + i];
        acc = group_sum(acc, g);
        if (sub == 0 && r < nrows) finish_fn(row0 + r, acc);
    }
}

// The row whose finish_fn this lane will run first in walk_row_tile_split (-1: none).  Kernels use it to fetch the
// row's scalars (incoming log-det, length) at the START of the wave: loaded inside finish_fn they cost every wave one
// more serial memory round trip at its very end, with nothing else in flight.
__device__ __forceinline__ int first_finish_row(const RowTiling& tl) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    if (tl.bpr) return threadIdx.x == 0 ? (int)blockIdx.x : -1;
    const long tile = (long)walker_block() * kWavesPerBlock + wave;
    if (tile >= tl.ntiles) return -1;
    const int row0 = (int)(tile * tl.rw);
    if (tl.rw == 1) return lane == 0 ? row0 : -1;
    const int g = kWave / tl.p2;
    const int r = lane / g;
    return ((lane & (g - 1)) == 0 && r < min(tl.rw, tl.B - row0)) ? row0 + r : -1;
}

// Single-functor form: chunk_fn(row, e0) loads, computes, stores and returns the contribution.
template <typename T, typename ChunkFn, typename FinishFn>
__device__ __forceinline__ void walk_row_tile(const RowTiling& tl, T* part, ChunkFn&& chunk_fn,
                                              FinishFn&& finish_fn) {
    struct Nothing {};
    walk_row_tile_split<1, T, Nothing, false>(
        tl, part, [](int, int) { return Nothing{}; },
        [&](const Nothing&, int row, int e0) { return chunk_fn(row, e0); }, finish_fn);
}

constexpr int kAccStride = 16;  // int64 words between two of the 64 batch-sum accumulators (128 B: one cache line each)

// ---- order-free sums: 64-bit fixed point with an fp64 escape word ------------------------------
// Sums whose terms arrive in no fixed order (per-row log-det sums of the token-pass mixture kernels, the batch NLL
// accumulator, the fp64 mixture backward's parameter-gradient words) are integer sums of v * 2^32: integer addition is
// associative, so the result is bit-reproducible.  The word holds |sum| < 2^31 only.  A term at or above the caller's limit
// (chosen so that the largest possible number of smaller terms stays below 2^31), and every +-inf / NaN term, goes to an
// fp64 word BESIDE the integer word with a floating-point atomic instead; the value of the pair is fix / 2^32 + big.  The
// reference sums these quantities in floating point (mixture_cdf_layer.py:95-123 `.sum(dim=[1,2])`, task.py:96-118), so a
// finite term of 1e10 gives a finite sum there, +-inf gives +-inf, NaN gives NaN — and the same here.  Reproducibility to
// the last bit is kept wherever no such term exists (the big word stays +0.0).
constexpr double kFix32 = 4294967296.0;
__device__ __forceinline__ long long to_fix(double v) { return __double2ll_rn(v * kFix32); }
__device__ __forceinline__ double fix_pair_value(long long fix, double big) { return (double)fix * (1.0 / kFix32) + big; }
__device__ __forceinline__ void fix_pair_add(unsigned long long* fix, double* big, double v, double lim) {
    if (fabs(v) < lim) atomicAdd(fix, (unsigned long long)to_fix(v));
    else unsafeAtomicAdd(big, v);           // ds_add_f64 / global_atomic_add_f64 (rare)
}
// The batch NLL accumulator (cnf_affine_coupling_nll_acc, cnf_mixture_coupling_nll): word 16 k is the fixed-point sum of
// slot k, word 16 k + 1 of the same 128-byte line its fp64 escape word.  Per-sample NLLs below 4096 (bits per dimension are
// O(1)) leave room for 2^19 samples per slot, 3.3e7 per accumulator, before the integer word could wrap.
constexpr double kAccTermMax = 4096.0;
__device__ __forceinline__ void nll_acc_add(long long* acc, int slot, float nll) {
    unsigned long long* w = reinterpret_cast<unsigned long long*>(acc) + (size_t)slot * kAccStride;
    fix_pair_add(w, reinterpret_cast<double*>(w + 1), (double)nll, kAccTermMax);
}

// logistic prior log-prob (distributions.py:129-136,154-163) for mu = 0:
// softplus(v) + softplus(-v) = |v| + 2 log(1 + e^{-|v|}), one exp and one log instead of two each,
// with the constants folded on the host: a = 1/sigma, a2 = log2(e)/sigma; the hardware
// exp2 / log2 are used directly (no range-scaling code, no fp32 division): 8 VALU instructions per element
// W^-1 of a D x D matrix, D <= 8, in fp64 (Gauss-Jordan, partial pivoting): threads tid < 64 of the workgroup work — lane
// (r, c) = (tid >> 3, tid & 7) owns entry [r][c] of W and of the inverse being built — and EVERY thread of the workgroup must
// call it (barriers inside).  A: [8][17] doubles of LDS.  The reference inverts in double as well (permutation_layers.py:76:
// torch.inverse(weight.double()).float()).  One body for cnf_actnorm_invconv_bwd's own inverse launch and for the LU weight
// assembly's by-product (cnf_invconv_lu_weight_inv): the same bits.
__device__ __forceinline__ void small_inverse_wg(const float* w, float* w_inv, int D, int tid, double (*A)[17]) {
    const int r = tid >> 3, c = tid & 7;
    const bool live = tid < 64 && r < D && c < D;
    if (live) {
        A[r][c] = (double)w[r * D + c];
        A[r][8 + c] = r == c ? 1.0 : 0.0;
    }
    __syncthreads();
    for (int k = 0; k < D; ++k) {
        int piv = k;
        double best = fabs(A[k][k]);
        for (int row = k + 1; row < D; ++row) {
            const double v = fabs(A[row][k]);
            if (v > best) {
                best = v;
                piv = row;
            }
        }
        const double k0 = live ? A[k][c] : 0.0, k1 = live ? A[k][8 + c] : 0.0;
        const double p0 = live ? A[piv][c] : 0.0, p1 = live ? A[piv][8 + c] : 0.0;
        __syncthreads();
        if (tid < 64 && r == 0 && c < D && piv != k) {
            A[k][c] = p0; A[k][8 + c] = p1;
            A[piv][c] = k0; A[piv][8 + c] = k1;
        }
        __syncthreads();
        double n0 = 0.0, n1 = 0.0;
        if (live) {
            const double pv = A[k][k];
            const double q0 = A[k][c] / pv, q1 = A[k][8 + c] / pv, f = A[r][k];
            n0 = r == k ? q0 : A[r][c] - f * q0;
            n1 = r == k ? q1 : A[r][8 + c] - f * q1;
        }
        __syncthreads();
        if (live) {
            A[r][c] = n0;
            A[r][8 + c] = n1;
        }
        __syncthreads();
    }
    if (live) w_inv[r * D + c] = (float)A[r][8 + c];
}

struct PriorConst {
    float inv_sigma, inv_sigma_log2e, log_sigma;
};
inline PriorConst make_prior_const(float sigma, float log_sigma) {
    PriorConst c;
    c.inv_sigma = (float)(1.0 / (double)sigma);
    c.inv_sigma_log2e = (float)(1.4426950408889634 / (double)sigma);
    c.log_sigma = log_sigma;
    return c;
}
__device__ __forceinline__ float prior_logp(float x, const PriorConst& c) {
    const float ax = fabsf(x);
    const float l2 = __builtin_amdgcn_logf(1.f + __builtin_amdgcn_exp2f(-ax * c.inv_sigma_log2e));
    return -((ax * c.inv_sigma + 1.3862943611198906f * l2) + c.log_sigma);
}

// expanded coupling mask value (coupling_layer.py:67-74): rows tile along N, cols broadcast over D
__device__ __forceinline__ float mask_at(const float* mask, int mr, int mc, int n, int d) {
    if (!mask) return 0.f;
    const int r = mr == 1 ? 0 : n % mr;
    return mask[r * mc + (mc == 1 ? 0 : d)];
}

}  // namespace cnf
