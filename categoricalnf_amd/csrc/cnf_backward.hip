// Backward (vector-Jacobian) kernels of the fp32 flow layers: affine coupling (both directions, and its static
// split forms), ExtActNorm, ActNorm, 1x1 convolution, logistic prior / NLL and the sigmoid flow.
//
// The reference has no backward code: it differentiates its eager op chains with autograd (train loop
// general/train.py:144-155).  These kernels compute the same gradients in one streaming pass per layer:
//   inputs   saved forward tensors + upstream grads g_zout [B,N,D], g_ldj [B]
//   outputs  g_z, g_nn (element-wise, streamed) and the parameter gradients (batch reductions).
//
// Round 4 rewrite on the forward path's tiling.  The element streams are walked in FLAT TILES: a tile is tc = 64 * U * G
// consecutive 16-byte chunks of the [B, L] tensors (no row alignment: a training batch of 64 long rows fills the chip
// like 16384 short ones), one wave per tile, U chunks per lane loaded back to back (16-byte, fully coalesced: lane i of
// load u reads bytes [16 i, 16 i + 16) of one contiguous 1 KiB span) before the first use; the row of a chunk (for
// g_ldj / length) and its position in the row (for the mask / channel) come from one mul_hi each (FastDiv) — there is no
// 64-bit division anywhere; transcendentals are v_exp_f32 / v_rcp_f32 in math mode 1.  The grid covers every tile with
// one wave (up to kBwdMaxBlocks workgroups; beyond that the waves stride over the tiles).
//
// Parameter gradients.  A lane adds its contributions to LANE-PRIVATE words of LDS (`acc[p * 256 + threadIdx.x]`: plain
// read-add-write, consecutive lanes hit consecutive banks, no two lanes ever share a word, a chunk's words are read
// together and written together: one LDS round trip per chunk) or keeps them in registers (1x1 convolution: a lane owns
// whole tokens there).  When a WAVE is through it sums its 64 words per parameter in a fixed order on the DPP network
// (quad_perm / row_half_mirror / row_mirror inside the rows of 16 lanes, row_bcast across them) and writes ONE row of
// `partials[waves][P]` — no barrier, no ticket, no atomic; bwd_reduce_partials_kernel sums the rows in fp64 in row order.
// No floating-point atomic anywhere: the gradients are bit-reproducible run to run.  (LDS float atomics — ds_add_f32 to
// lane-private words — were measured first: ~125 cycles per wave instruction, 54 / 79 us for the affine / ActNorm kernels.)
#include "cnf_common.h"

#include <algorithm>
#include <atomic>
#include <type_traits>

namespace cnf {

constexpr int kBwdMaxBlocks = 4096;     // workgroups; every WAVE writes one row of the partials buffer (cnf_bwd_workspace_floats)
constexpr int kBwdMaxRows = kBwdMaxBlocks * kWavesPerBlock;
constexpr int kBwdMaxRowP = 160;        // widest row of these kernels (1x1 convolution at D = 12: 145 entries)
constexpr int kBwdMaxTab = 64;          // (mask period x D) + VEC - 1 per-channel constants (one lane builds one entry), one private copy per wave
constexpr int kBwdMaxD = 64;            // channels of the per-workgroup constant tables

// chunks in flight per lane (1..3) and chunk groups per wave tile, 0 = every kernel's own default; process-wide, set
// before use (cnf_set_bwd_tile)
static std::atomic<int> g_bwd_u{0}, g_bwd_g{0};
// ActNorm backward: 1 = the token-owner tile kernel where it applies (default), 0 = always the flat-tile kernel (A/B, tests)
static std::atomic<int> g_act_bwd_tiles{1}, g_aff_bwd_tiles{1};

// partials [P, nrows <= kBwdMaxRows] (column-major rows of the waves) -> column sums in fp64 in a fixed order; columns [0, split) go to out_a, the rest to
// out_b (either may be null): the parameter-gradient tensors are written directly.  One workgroup of 1024 threads per
// column; a thread's (up to 16) loads are all issued before the first add — the launch is one memory round trip long.
constexpr int kReduceBlock = 1024;
// one output column: extra >= 0: column `extra` is added to every column >= extra_from (ActNorm's log-det term belongs to
// every d scales[d])
__device__ __forceinline__ void reduce_column(const float* partials, int nrows, float* out_a, float* out_b, int split, int extra,
                                              int extra_from, int p) {
    constexpr int K = kBwdMaxRows / kReduceBlock;
    const bool with_extra = extra >= 0 && p >= extra_from;
    float v[K], w[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const size_t r = (size_t)min((int)threadIdx.x + k * kReduceBlock, nrows - 1);
        v[k] = partials[(size_t)p * nrows + r];
        w[k] = with_extra ? partials[(size_t)extra * nrows + r] : 0.f;
    }
    double acc = 0.0;
#pragma unroll
    for (int k = 0; k < K; ++k)
        if ((int)threadIdx.x + k * kReduceBlock < nrows) acc += (double)v[k] + (double)w[k];
    __shared__ double sh[kReduceBlock / kWave];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < kReduceBlock / kWave; ++w) t += sh[w];
        if (p < split) {
            if (out_a) out_a[p] = (float)t;
        } else if (out_b) {
            out_b[p - split] = (float)t;
        }
    }
}
__global__ __launch_bounds__(kReduceBlock) void bwd_reduce_partials_kernel(const float* partials, int nrows, int P, float* out_a,
                                                                          float* out_b, int split, int extra, int extra_from) {
    (void)P;
    reduce_column(partials, nrows, out_a, out_b, split, extra, extra_from, blockIdx.x);
}

// ---- deferred reductions (cnf_bwd_defer_begin / cnf_bwd_defer_flush) ------------------------------------------------------
// Every streaming backward entry point ends with the small dependent launch above: ~4.7 us as the caller sees it (two
// launch gaps and one memory round trip for a few KB of partials), 18.6 instead of 14.8 us for the ActNorm backward.  A host
// that owns a whole backward pass can collect those reductions and run them as ONE launch: between begin and flush the entry
// points only write their partial rows (each call needs its OWN workspace, alive until the flush) and queue a job; the flush
// launches one workgroup per output column of every job.  Same kernel body, same summation order: the gradients are the
// bits of the immediate reductions.  The job table travels in the kernel arguments.
struct ReduceJob {
    const float* partials;
    float* out_a;
    float* out_b;
    int nrows, split, extra, extra_from;
    int col0, cols;                     // this job's workgroups are [col0, col0 + cols)
};
constexpr int kMaxReduceJobs = 24;
struct ReduceJobs {
    int n;
    ReduceJob j[kMaxReduceJobs];
};
__global__ __launch_bounds__(kReduceBlock) void bwd_reduce_jobs_kernel(ReduceJobs jobs) {
    int k = 0;
    while (k + 1 < jobs.n && (int)blockIdx.x >= jobs.j[k + 1].col0) ++k;
    const ReduceJob& jb = jobs.j[k];
    reduce_column(jb.partials, jb.nrows, jb.out_a, jb.out_b, jb.split, jb.extra, jb.extra_from, (int)blockIdx.x - jb.col0);
}
struct DeferState {
    bool on = false;
    hipStream_t stream = nullptr;       // the stream the queued jobs' kernels were launched on: their reduction goes there and nowhere else
    ReduceJobs jobs;
};
static thread_local DeferState g_defer;
static void flush_reduce_jobs() {
    ReduceJobs& q = g_defer.jobs;
    if (q.n > 0) {
        const int total = q.j[q.n - 1].col0 + q.j[q.n - 1].cols;
        CNF_LAUNCH(bwd_reduce_jobs_kernel, dim3(total), dim3(kReduceBlock), 0, g_defer.stream, q);
        q.n = 0;
    }
}
// the closing reduction of an entry point: now, or queued for cnf_bwd_defer_flush
static void enqueue_reduce(int cols, const float* partials, int nrows, int P, float* out_a, float* out_b, int split, int extra,
                           int extra_from, hipStream_t st) {
    if (!g_defer.on) {
        CNF_LAUNCH(bwd_reduce_partials_kernel, dim3(cols), dim3(kReduceBlock), 0, st, partials, nrows, P, out_a, out_b, split, extra, extra_from);
        return;
    }
    ReduceJobs& q = g_defer.jobs;
    // a full table, or a call on ANOTHER stream than the queued ones: what is queued is reduced first, on its own stream
    if (q.n == kMaxReduceJobs || (q.n > 0 && st != g_defer.stream)) flush_reduce_jobs();
    g_defer.stream = st;
    ReduceJob& jb = q.j[q.n];
    jb.partials = partials; jb.out_a = out_a; jb.out_b = out_b;
    jb.nrows = nrows; jb.split = split; jb.extra = extra; jb.extra_from = extra_from;
    jb.col0 = q.n ? q.j[q.n - 1].col0 + q.j[q.n - 1].cols : 0;
    jb.cols = cols;
    ++q.n;
}

// ---- fixed-order cross-lane sums on the DPP network (no LDS round trip, no ds_bpermute) --------------------------------
// lanes without a source, and rows outside ROW_MASK, read 0
template <int CTRL, int ROW_MASK = 0xf>
__device__ __forceinline__ float dpp0(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), CTRL, ROW_MASK, 0xf, false));
}
// every lane gets the sum over the 16 lanes of its DPP row: lane pairs, quads, half rows (i <-> 7 - i), rows (i <-> 15 - i)
__device__ __forceinline__ float row16_sum(float v) {
    v += dpp0<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp0<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp0<0x141>(v);     // row_half_mirror
    v += dpp0<0x140>(v);     // row_mirror
    return v;
}
// input: row sums (row16_sum); lane 63 gets ((r3 + r2) + (r1 + r0))
__device__ __forceinline__ float rows_total_in_lane63(float v) {
    v += dpp0<0x142, 0xa>(v);    // row_bcast15: lane 15 of rows 0 / 2 -> rows 1 / 3
    v += dpp0<0x143, 0xc>(v);    // row_bcast31: lane 31 -> rows 2, 3
    return v;
}

// Lanes of ONE wave exchange data through LDS: the LDS queue of a wave is in order, so a read issued after a write sees
// it; all that is needed is that the compiler keeps the order ("memory") — and NOT cnf_common.h's wave_lds_sync, whose
// release fence also drains the vector-memory counter: at the end of a streaming wave that is a wait for every gradient
// store still in flight (one memory round trip) in front of the parameter-gradient reduction, whose own row store then
// pays a second one (measured: affine backward 33.2 -> 31 us forward direction, 35.9 -> 30.3 inverse direction).
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- a wave's parameter-gradient sums -> its row of the partials buffer --------------------------------------------------
// No barrier, no ticket, no atomics: a wave reduces what its own lanes hold and writes row (blockIdx.x * 4 + wave).
// (One row per WORKGROUP was measured first: a closing barrier or a ticket drawn by the last wave — with the barrier the
// ticket word needs at the start — costs the affine kernel 2 us at the benchmark shape, the serial per-lane fold into
// lane-private words another 3.)
// The partials buffer is COLUMN-major, partials[p * nrows + row] with nrows = waves of the launch: the reduction launch
// then reads one contiguous run per parameter (row-major made every one of its workgroups touch every line of the buffer:
// 7 us instead of 4 for the 37 columns of the 1x1 convolution).
struct WaveRow {
    float* base;        // &partials[row]
    size_t stride;      // nrows
    __device__ __forceinline__ float& operator[](int p) const { return base[(size_t)p * stride]; }
};
__device__ __forceinline__ WaveRow wave_partials_row(float* partials) {
    return WaveRow{partials + ((size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)), (size_t)gridDim.x * kWavesPerBlock};
}

// Per-element sums in LANE-PRIVATE LDS words my_acc[slot] (slot = column * kBlock; my_acc = acc + threadIdx.x): plain
// read-add-write, no other lane ever touches the word.  `distinct`: the VEC slots of a chunk are different words (at least
// VEC channels) -> one LDS round trip for the chunk instead of VEC dependent ones.
template <int VEC>
__device__ __forceinline__ void lane_private_add(float* my_acc, const int (&slots)[VEC], const float (&vals)[VEC], bool distinct) {
    if (distinct) {
        float r[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) r[j] = my_acc[slots[j]];
        asm volatile("" ::: "memory");
#pragma unroll
        for (int j = 0; j < VEC; ++j) my_acc[slots[j]] = r[j] + vals[j];
    } else {
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            my_acc[slots[j]] += vals[j];
            asm volatile("" ::: "memory");
        }
    }
}
// lane-private words acc[p * kBlock + threadIdx.x], p < P: this wave's 64 lanes per column.  Eight columns at a time: the
// eight reads go out together, the eight DPP chains interleave, column k's total lands in lane k and the eight totals
// leave in ONE store instruction.
__device__ __forceinline__ void wave_lane_private_reduce(const float* acc, int P, const WaveRow out_row) {
    wave_lds_order();
    const int lane = threadIdx.x & 63;
    for (int p0 = 0; p0 < P; p0 += 8) {
        float r[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = acc[(size_t)min(p0 + k, P - 1) * kBlock + threadIdx.x];
        float mine = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float v = rows_total_in_lane63(row16_sum(r[k]));
            const float t = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), kWave - 1));
            if (lane == k) mine = t;
        }
        if (lane < 8 && p0 + lane < P) out_row[p0 + lane] = mine;
    }
}

// register sums (R per lane, the same parameter entry in every lane): DPP row sums, the four rows of a column meet in
// the wave's LDS strip `wsh` (4 R floats) and lane p adds them in row order
template <int R>
__device__ __forceinline__ void wave_register_reduce(const float (&acc)[R], float* wsh, const WaveRow out_row) {
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int p = 0; p < R; ++p) {
        const float v = row16_sum(acc[p]);
        if ((lane & 15) == 0) wsh[(lane >> 4) * R + p] = v;
    }
    wave_lds_order();
    for (int p = lane; p < R; p += kWave) out_row[p] = (wsh[p] + wsh[R + p]) + (wsh[2 * R + p] + wsh[3 * R + p]);
}

// ---- flat tiles -------------------------------------------------------------------------------------------------------------
struct FlatTiling {
    long nchunks;       // B * cpr
    long ntiles;
    int vec;            // floats per chunk: 4, 2 or 1 (the largest the row length and the pointers allow)
    int cpr;            // chunks per row
    int L;              // elements per row
    int tc;             // chunks per tile: 64 * U * G
    int fast_rows;      // every chunk index of the launch < 2^32 / cpr: a tile's first row is one mul_hi
    FastDiv div_cpr;    // exact for n < 2^32 / cpr
};

static inline bool aligned_to(size_t bytes, std::initializer_list<const void*> ptrs) {
    for (const void* p : ptrs)
        if (p && (reinterpret_cast<uintptr_t>(p) & (bytes - 1))) return false;
    return true;
}

static FlatTiling make_flat_tiling(long B, int L, int U, int vec, int G) {
    FlatTiling t;
    t.vec = vec;
    t.cpr = L / vec;
    t.L = L;
    t.nchunks = B * t.cpr;
    const long limit = (1l << 32) / std::max(t.cpr, 1);
    // in-tile chunk offsets go through FastDiv: (tc + cpr) must stay below 2^32 / cpr
    while (G > 1 && (long)kWave * U * G + t.cpr >= limit) --G;
    t.tc = kWave * U * G;
    t.ntiles = (t.nchunks + t.tc - 1) / t.tc;
    t.fast_rows = t.nchunks < limit ? 1 : 0;
    t.div_cpr = make_fastdiv((uint32_t)t.cpr);
    return t;
}
static inline int vec_for(int L, std::initializer_list<const void*> ptrs) {
    if (L % 4 == 0 && aligned_to(16, ptrs)) return 4;
    if (L % 2 == 0 && aligned_to(8, ptrs)) return 2;
    return 1;
}
static inline dim3 flat_grid(const FlatTiling& t) {
    const long blocks = (t.ntiles + kWavesPerBlock - 1) / kWavesPerBlock;
    return dim3((unsigned)std::min<long>(std::max<long>(blocks, 1), kBwdMaxBlocks));
}
// the row walkers need (tc + cpr) * cpr < 2^32 (cpr < ~60 000 chunks: rows of up to 240 000 elements)
static inline bool flat_ok(int L, int vec) {
    const long cpr = L / vec;
    return (3l * kWave + cpr) * cpr < (1l << 32);
}

// Walk this wave's tiles.  For each chunk of VEC elements starting at element e0 of `row`:
//   load_fn(row, e0) -> Data     issues the global loads (U chunks back to back);
//   proc_fn(data, row, e0, u)    computes and stores (u = which of the lane's U chunks in flight).
// pre_fn() runs once per wave after the first group of loads has been issued (per-wave LDS tables hide behind them); a
// wave without a tile runs it too (it may contain what the workgroup's closing barrier needs).
template <int VEC, int U, typename Data, typename LoadFn, typename ProcFn, typename PreFn>
__device__ __forceinline__ void walk_flat_tiles(const FlatTiling& tl, LoadFn&& load_fn, ProcFn&& proc_fn, PreFn&& pre_fn) {
    const int lane = threadIdx.x & 63;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    bool first = true;
    for (long tile = wave_id; tile < tl.ntiles; tile += nwaves) {
        const long c_base = tile * tl.tc;
        const uint32_t row0 = tl.fast_rows ? fdiv((uint32_t)c_base, tl.div_cpr) : (uint32_t)(c_base / tl.cpr);
        const uint32_t off0 = (uint32_t)(c_base - (long)row0 * tl.cpr);
        const long left = tl.nchunks - c_base;
        const int nch = left < (long)tl.tc ? (int)left : tl.tc;
        for (int g0 = 0; g0 < nch; g0 += kWave * U) {
            Data dat[U];
            int rr[U], ee[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                // lanes past the end of the tile re-load its last chunk: unconditional loads, no control flow around them
                const uint32_t q = off0 + (uint32_t)min(g0 + kWave * u + lane, nch - 1);
                const uint32_t r = fdiv(q, tl.div_cpr);
                rr[u] = (int)(row0 + r);
                ee[u] = (int)(q - r * tl.cpr) * VEC;
                dat[u] = load_fn(rr[u], ee[u]);
            }
            if (first) {
                pre_fn();
                first = false;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (g0 + kWave * u + lane < nch) proc_fn(dat[u], rr[u], ee[u], u);
        }
    }
    if (first) pre_fn();
}

// 16 / 8 / 4-byte I/O of a chunk.  NT = nontemporal (tensors this kernel is the only reader of: the saved forward
// tensors; the gradients it writes are read by the next backward kernel and keep the plain store).
typedef float bw_f4 __attribute__((ext_vector_type(4)));
typedef float bw_f2 __attribute__((ext_vector_type(2)));
template <int VEC, bool NT>
__device__ __forceinline__ void ld_chunk(const float* p, float* v) {
    if constexpr (VEC == 4) {
        const bw_f4 q = NT ? __builtin_nontemporal_load(reinterpret_cast<const bw_f4*>(p)) : *reinterpret_cast<const bw_f4*>(p);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if constexpr (VEC == 2) {
        const bw_f2 q = NT ? __builtin_nontemporal_load(reinterpret_cast<const bw_f2*>(p)) : *reinterpret_cast<const bw_f2*>(p);
        v[0] = q.x; v[1] = q.y;
    } else {
        v[0] = NT ? __builtin_nontemporal_load(p) : *p;
    }
}
template <int VEC, bool NT>
__device__ __forceinline__ void st_chunk(float* p, const float* v) {
    if constexpr (VEC == 4) {
        const bw_f4 q = {v[0], v[1], v[2], v[3]};
        if (NT) __builtin_nontemporal_store(q, reinterpret_cast<bw_f4*>(p));
        else *reinterpret_cast<bw_f4*>(p) = q;
    } else if constexpr (VEC == 2) {
        const bw_f2 q = {v[0], v[1]};
        if (NT) __builtin_nontemporal_store(q, reinterpret_cast<bw_f2*>(p));
        else *reinterpret_cast<bw_f2*>(p) = q;
    } else {
        if (NT) __builtin_nontemporal_store(v[0], p);
        else *p = v[0];
    }
}
// Cache hints of the backward streams.  bit 0: nontemporal loads of the saved forward tensors (z_out / x, nn_out);
// bit 1: nontemporal loads of the upstream gradient; bit 2: nontemporal stores of the gradients.
#ifndef CNF_BWD_NT
#define CNF_BWD_NT 5
#endif
// timing experiments only (results are wrong with any bit set): 1 = no fold / combine at the end of the affine kernel,
// 2 = no zero fill / ticket barrier at its start, 4 = no per-element contribution arithmetic
#ifndef CNF_BWD_ABLATE
#define CNF_BWD_ABLATE 0
#endif
constexpr bool kNtSaved = (CNF_BWD_NT & 1) != 0, kNtUp = (CNF_BWD_NT & 2) != 0, kNtOut = (CNF_BWD_NT & 4) != 0;

template <bool FAST>
__device__ __forceinline__ float bexp(float x) { return FAST ? __builtin_amdgcn_exp2f(x * 1.4426950408889634f) : expf(x); }
// 1 / (e^x + 1): tanh(x / 2) = 1 - 2 r, sigmoid(x) = 1 - r
template <bool FAST>
__device__ __forceinline__ float rcp_exp_p1(float x) {
    return FAST ? __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(x * 1.4426950408889634f) + 1.f) : 1.f / (expf(x) + 1.f);
}

// ---- affine coupling (coupling_layer.py:53-63, 88-98) ------------------------------------------------------------------------
struct AffBwdArgs {
    const float* z_out;     // forward OUTPUT of the direction being differentiated
    const float* nn;
    const float* sf;        // nullable
    const float* mask;
    const float* g_zout;    // nullable (= zeros)
    const float* g_ldj;     // nullable
    float* g_z;
    float* g_nn;
    float* partials;        // [gridDim.x, D] (d scaling_factor), used when sf != null
    int N, D, L, mr, mc;
    int P;                  // mask period x D: length of the per-channel constant table
    FastDiv div_p, div_d;
};
// per (mask row, channel): keep = 1 - mask, keepf = keep e^sf, kscale = keepf / max(e^sf, 1),
// x3 = 2 log2(e) / max(e^sf, 1) (fast math) or max(e^sf, 1) (exact), nfl = -[e^sf >= 1] (clamp(min=1) passes the
// gradient there, torch semantics), slot = word offset of the channel's accumulator column
struct alignas(16) AffTabA {
    float keep, keepf, x3, kscale;
};
struct alignas(8) AffTabB {
    float nfl;
    int slot;
};
template <int VEC>
struct AffBwdChunk {
    float zo[VEC], sr[VEC], tr[VEC], gzo[VEC];
    float gl;
};

// The scaling-factor gradient: a lane adds every element's contribution to its LANE-PRIVATE LDS word of the element's
// channel (lane_private_add: plain read-add-write, one LDS round trip per chunk), and at the end the wave sums its 64
// words per channel on the DPP network and writes one row of the partials buffer (wave_lane_private_reduce).
// Measured against per-lane REGISTER sums (64 lanes x 3 chunks, or 63 lanes, so that a lane's chunks always start at
// the same channel; the sums then have to be sorted by channel through LDS at the end of every wave): 32.7 / 32.0 us
// (forward / inverse direction) against 33.4 / 36.3 at the benchmark shape (profiles/r04_bwd_ab.txt) — the LDS words
// cost the inner loop 12 instructions per chunk, the sorting cost every wave as much as its whole inner loop.
template <int VEC, int U, bool HAS_SF, bool REVERSE, bool FAST>
__global__ __launch_bounds__(kBlock) void affine_bwd_kernel(AffBwdArgs a, FlatTiling tl) {
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [D][kBlock] lane-private words (HAS_SF)
    __shared__ AffTabA tabA_all[kWavesPerBlock][kBwdMaxTab];
    __shared__ AffTabB tabB_all[kWavesPerBlock][kBwdMaxTab];
    AffTabA* tabA = tabA_all[threadIdx.x >> 6];
    AffTabB* tabB = tabB_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    float* my_acc = acc + threadIdx.x;
    if (HAS_SF)
        for (int d = 0; d < a.D; ++d) my_acc[d * kBlock] = 0.f;
    // the table's two tiny loads go out first, the table is finished behind the first chunk loads (walk_flat_tiles)
    const int ntab = a.P + VEC - 1;
    float m_raw = 0.f, sf_raw = 0.f;
    int d_tab = 0;
    if (lane < ntab) {
        const int tp = lane % a.P;
        const int r = tp / a.D;
        d_tab = tp - r * a.D;
        if (a.mask) m_raw = a.mask[r * a.mc + (a.mc == 1 ? 0 : d_tab)];
        if (HAS_SF) sf_raw = a.sf[d_tab];
    }
    auto pre = [&]() {
        asm volatile("" : "+v"(sf_raw), "+v"(m_raw) : : "memory");
        if (lane < ntab) {
            const float f = HAS_SF ? expf(sf_raw) : 1.f;
            const float fc = fmaxf(f, 1.f);
            AffTabA ta;
            ta.keep = 1.f - m_raw;
            ta.keepf = (1.f - m_raw) * f;
            ta.x3 = FAST ? 2.8853900817779268f / fc : fc;
            ta.kscale = ta.keepf / fc;
            tabA[lane] = ta;
            AffTabB tb;
            tb.nfl = f >= 1.f ? -1.f : 0.f;
            tb.slot = d_tab * kBlock;
            tabB[lane] = tb;
        }
        wave_lds_order();
    };
    // a missing upstream gradient is zero: the load goes to a valid address anyway and its result is dropped
    const float* gz_src = a.g_zout ? a.g_zout : a.z_out;
    const bool has_gz = a.g_zout != nullptr, has_gl = a.g_ldj != nullptr;
    const float* gl_src = has_gl ? a.g_ldj : a.z_out;
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        AffBwdChunk<VEC> c;
        ld_chunk<VEC, kNtSaved>(a.z_out + off, c.zo);
        float st[2 * VEC];
        ld_chunk<VEC, kNtSaved>(a.nn + 2 * off, st);
        if (VEC > 1) ld_chunk<VEC, kNtSaved>(a.nn + 2 * off + VEC, st + VEC);
        else st[1] = a.nn[2 * off + 1];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            c.sr[j] = st[2 * j];
            c.tr[j] = st[2 * j + 1];
        }
        ld_chunk<VEC, kNtUp>(gz_src + off, c.gzo);
        c.gl = gl_src[has_gl ? row : 0];
        return c;
    };
    auto proc = [&](const AffBwdChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        const int ti0 = e0 - (int)fdiv((uint32_t)e0, a.div_p) * a.P;       // e0 mod P
        const float gl = has_gl ? c.gl : 0.f;
        float gz[VEC], gn[2 * VEC], contribs[VEC];
        int slots[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const AffTabA ta = tabA[ti0 + j];
            const float gzo = has_gz ? c.gzo[j] : 0.f;
            float s, sech2 = 0.f;
            if (!HAS_SF) {
                s = c.sr[j] * ta.keep;
            } else if (FAST) {
                // r = 1 / (e^{2u} + 1), u = s_raw / max(f, 1): tanh u = 1 - 2 r, sech^2 u = 4 r (1 - r)
                const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(c.sr[j] * ta.x3) + 1.f);
                s = fmaf(r, -2.f * ta.keepf, ta.keepf);
                sech2 = 4.f * r * (1.f - r);
            } else {
                const float th = tanhf(c.sr[j] / ta.x3);
                s = th * ta.keepf;
                sech2 = 1.f - th * th;
            }
            float gs, gt;
            if (!REVERSE) {                         // z' = (z + t) e^s ; ldj += s
                gz[j] = gzo * bexp<FAST>(s);
                gt = gz[j];
                gs = fmaf(gzo, c.zo[j], gl);
            } else {                                // z' = z e^-s - t ; ldj -= s
                gz[j] = gzo * bexp<FAST>(-s);
                gt = -gzo;
                gs = -fmaf(gzo, c.zo[j] + c.tr[j] * ta.keep, gl);
            }
            float g_sr;
            if (HAS_SF) {
                // d s / d s_raw = keepf sech^2 / fc;  d s / d scaling_factor = keepf (th - [f >= 1] u sech^2), so the
                // channel's gradient collects gs s - [f >= 1] s_raw g_sr
                g_sr = gs * sech2 * ta.kscale;
                const AffTabB tb = tabB[ti0 + j];
                const float contrib = fmaf(tb.nfl * c.sr[j], g_sr, gs * s);
                contribs[j] = contrib;
                slots[j] = tb.slot;
            } else {
                g_sr = gs * ta.keep;
            }
            gn[2 * j] = g_sr;
            gn[2 * j + 1] = gt * ta.keep;
        }
        st_chunk<VEC, kNtOut>(a.g_z + off, gz);
        if (VEC > 1) {
            st_chunk<VEC, kNtOut>(a.g_nn + 2 * off, gn);
            st_chunk<VEC, kNtOut>(a.g_nn + 2 * off + VEC, gn + VEC);
        } else {
            st_chunk<2, kNtOut>(a.g_nn + 2 * off, gn);
        }
        if (HAS_SF && !(CNF_BWD_ABLATE & 4)) lane_private_add<VEC>(my_acc, slots, contribs, a.D >= VEC);
    };
    walk_flat_tiles<VEC, U, AffBwdChunk<VEC>>(tl, load, proc, pre);
    if (HAS_SF && !(CNF_BWD_ABLATE & 1)) wave_lane_private_reduce(acc, a.D, wave_partials_row(a.partials));
}

// ---- affine coupling, token-owner wave tiles (channel masks, D in {2, 3, 4, 6, 8}) -------------------------------------------------
// The form of ext_actnorm_bwd_tile_kernel / invconv_bwd_kernel below: a wave takes spans of 64 token groups with coalesced
// 16-byte loads, transposes them through its LDS strips so that a lane owns whole tokens — every per-channel constant (mask,
// e^sf, clamp flag) and the D scaling-factor sums then live in REGISTERS: no per-element LDS table read, no lane-private LDS
// read-add-write.  Measured against the flat-tile kernel above at the benchmark shape: profiles/r04_bwd_probe.txt.
struct AffTileArgs {
    const float* z_out;
    const float* nn;        // [B,N,2D] interleaved (s_raw, t) pairs
    const float* sf;        // nullable
    const float* mask;      // [D] channel mask or null (nothing kept)
    const float* g_zout;
    const float* g_ldj;     // nullable
    float* g_z;
    float* g_nn;
    float* partials;
    long ntok;
    int N, D;
    FastDiv div_n;
    int fast_rows;
};
template <int D, bool HAS_SF, bool REVERSE, bool FAST>
__global__ __launch_bounds__(kBlock) void affine_bwd_tile_kernel(AffTileArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4, NC = 2 * NV;
    __shared__ bw_f4 strip_all[kWavesPerBlock][4][kWave * NV];
    __shared__ float comb[kWavesPerBlock][4 * D];
    bw_f4* sz = strip_all[threadIdx.x >> 6][0];
    bw_f4* sg = strip_all[threadIdx.x >> 6][1];
    bw_f4* sc0 = strip_all[threadIdx.x >> 6][2];        // vectors [0, NV) of every group's nn row
    bw_f4* sc1 = strip_all[threadIdx.x >> 6][3];        // vectors [NV, 2 NV)
    const int lane = threadIdx.x & 63;
    float keep[D], keepf[D], x3[D], kscale[D], nfl[D], acc[D];
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float m = a.mask ? a.mask[d] : 0.f;
        const float f = HAS_SF ? expf(a.sf[d]) : 1.f;
        const float fc = fmaxf(f, 1.f);
        keep[d] = 1.f - m;
        keepf[d] = (1.f - m) * f;
        x3[d] = FAST ? 2.8853900817779268f / fc : fc;
        kscale[d] = keepf[d] / fc;
        nfl[d] = f >= 1.f ? -1.f : 0.f;
        acc[d] = 0.f;
    }
    // a missing upstream gradient is zero: the loads go to a valid address anyway and their result is dropped
    const bool has_gz = a.g_zout != nullptr;
    const float* gz_src = has_gz ? a.g_zout : a.z_out;
    // one token: zo / gzo [D], nv = (s_raw, t) pairs [2D] in; gz [D], gn [2D] out (the arithmetic of affine_bwd_kernel)
    auto token = [&](const float* zo, const float* gin, const float* nv, float gl, float* gz, float* gn) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
            const float sr = nv[2 * d], tr = nv[2 * d + 1], gzo = has_gz ? gin[d] : 0.f;
            float s, sech2 = 0.f;
            if (!HAS_SF) {
                s = sr * keep[d];
            } else if (FAST) {
                const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(sr * x3[d]) + 1.f);
                s = fmaf(r, -2.f * keepf[d], keepf[d]);
                sech2 = 4.f * r * (1.f - r);
            } else {
                const float th = tanhf(sr / x3[d]);
                s = th * keepf[d];
                sech2 = 1.f - th * th;
            }
            float gs, gt;
            if (!REVERSE) {                         // z' = (z + t) e^s ; ldj += s
                gz[d] = gzo * bexp<FAST>(s);
                gt = gz[d];
                gs = fmaf(gzo, zo[d], gl);
            } else {                                // z' = z e^-s - t ; ldj -= s
                gz[d] = gzo * bexp<FAST>(-s);
                gt = -gzo;
                gs = -fmaf(gzo, zo[d] + tr * keep[d], gl);
            }
            float g_sr;
            if (HAS_SF) {
                g_sr = gs * sech2 * kscale[d];
                acc[d] += fmaf(nfl[d] * sr, g_sr, gs * s);
            } else {
                g_sr = gs * keep[d];
            }
            gn[2 * d] = g_sr;
            gn[2 * d + 1] = gt * keep[d];
        }
    };
    auto row_gl = [&](long tok) {
        if (!a.g_ldj) return 0.f;
        const long b = a.fast_rows ? (long)fdiv((uint32_t)tok, a.div_n) : tok / a.N;
        return a.g_ldj[b];
    };
    const long ngroups = a.ntok / TP;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const bw_f4* srcz = reinterpret_cast<const bw_f4*>(a.z_out + tile * (kWave * TP * D));
        const bw_f4* srcg = reinterpret_cast<const bw_f4*>(gz_src + tile * (kWave * TP * D));
        const bw_f4* srcc = reinterpret_cast<const bw_f4*>(a.nn + tile * (kWave * TP * 2 * D));
        bw_f4 qz[NV], qg[NV], qc[NC];
#pragma unroll
        for (int v = 0; v < NV; ++v) qz[v] = __builtin_nontemporal_load(srcz + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NC; ++v) qc[v] = __builtin_nontemporal_load(srcc + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) qg[v] = srcg[v * kWave + lane];
        const long tok0 = (tile * kWave + lane) * TP;
        float gl[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) gl[k] = row_gl(tok0 + k);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sz[v * kWave + lane] = qz[v];
            sg[v * kWave + lane] = qg[v];
        }
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int i = v * kWave + lane, grp = i / NC, j = i - grp * NC;
            (j < NV ? sc0 : sc1)[grp * NV + (j < NV ? j : j - NV)] = qc[v];
        }
        wave_lds_order();
        float zo[TP * D], gzo[TP * D], cv[2 * TP * D], gz[TP * D], gc[2 * TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = sz[lane * NV + v];
            zo[4 * v] = r.x; zo[4 * v + 1] = r.y; zo[4 * v + 2] = r.z; zo[4 * v + 3] = r.w;
            const bw_f4 q = sg[lane * NV + v];
            gzo[4 * v] = q.x; gzo[4 * v + 1] = q.y; gzo[4 * v + 2] = q.z; gzo[4 * v + 3] = q.w;
            const bw_f4 c0 = sc0[lane * NV + v];
            cv[4 * v] = c0.x; cv[4 * v + 1] = c0.y; cv[4 * v + 2] = c0.z; cv[4 * v + 3] = c0.w;
            const bw_f4 c1 = sc1[lane * NV + v];
            cv[4 * (NV + v)] = c1.x; cv[4 * (NV + v) + 1] = c1.y; cv[4 * (NV + v) + 2] = c1.z; cv[4 * (NV + v) + 3] = c1.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) token(zo + k * D, gzo + k * D, cv + k * 2 * D, gl[k], gz + k * D, gc + k * 2 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {              // the lane's own group: nobody else reads these words
            const bw_f4 r = {gz[4 * v], gz[4 * v + 1], gz[4 * v + 2], gz[4 * v + 3]};
            sz[lane * NV + v] = r;
            const bw_f4 c0 = {gc[4 * v], gc[4 * v + 1], gc[4 * v + 2], gc[4 * v + 3]};
            sc0[lane * NV + v] = c0;
            const bw_f4 c1 = {gc[4 * (NV + v)], gc[4 * (NV + v) + 1], gc[4 * (NV + v) + 2], gc[4 * (NV + v) + 3]};
            sc1[lane * NV + v] = c1;
        }
        wave_lds_order();
        bw_f4* dz = reinterpret_cast<bw_f4*>(a.g_z + tile * (kWave * TP * D));
        bw_f4* dc = reinterpret_cast<bw_f4*>(a.g_nn + tile * (kWave * TP * 2 * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) st_chunk<4, kNtOut>(reinterpret_cast<float*>(dz + v * kWave + lane), reinterpret_cast<const float*>(&sz[v * kWave + lane]));
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int i = v * kWave + lane, grp = i / NC, j = i - grp * NC;
            st_chunk<4, kNtOut>(reinterpret_cast<float*>(dc + i), reinterpret_cast<const float*>(&(j < NV ? sc0 : sc1)[grp * NV + (j < NV ? j : j - NV)]));
        }
        wave_lds_order();                            // the strips are refilled by the next tile
    }
    // tokens that do not fill a wave tile
    for (long tok = ntiles * kWave * TP + (long)blockIdx.x * kBlock + threadIdx.x; tok < a.ntok; tok += (long)gridDim.x * kBlock) {
        float zo[D], gzo[D], cv[2 * D], gz[D], gc[2 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            zo[d] = a.z_out[tok * D + d];
            gzo[d] = gz_src[tok * D + d];
            cv[2 * d] = a.nn[tok * 2 * D + 2 * d];
            cv[2 * d + 1] = a.nn[tok * 2 * D + 2 * d + 1];
        }
        token(zo, gzo, cv, a.g_ldj ? a.g_ldj[tok / a.N] : 0.f, gz, gc);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a.g_z[tok * D + d] = gz[d];
            a.g_nn[tok * 2 * D + 2 * d] = gc[2 * d];
            a.g_nn[tok * 2 * D + 2 * d + 1] = gc[2 * d + 1];
        }
    }
    if (HAS_SF) wave_register_reduce<D>(acc, comb[threadIdx.x >> 6], wave_partials_row(a.partials));
}

// ---- static-API split forms of the affine coupling (coupling_layer.py:76-98) ---------------------------------
struct AffParamsBwdArgs {
    const float* nn;
    const float* sf;
    const float* mask;
    const float* g_s;       // nullable
    const float* g_t;       // nullable
    float* g_nn;
    float* partials;
    int N, D, L, mr, mc, P;
    FastDiv div_p, div_d;
};
template <int VEC>
struct AffParamsChunk {
    float sr[VEC], gs[VEC], gt[VEC];
};
template <int VEC, int U, bool HAS_SF>
__global__ __launch_bounds__(kBlock) void affine_params_bwd_kernel(AffParamsBwdArgs a, FlatTiling tl) {
    extern __shared__ __attribute__((aligned(16))) float acc[];
    __shared__ AffTabA tabA_all[kWavesPerBlock][kBwdMaxTab];
    __shared__ AffTabB tabB_all[kWavesPerBlock][kBwdMaxTab];
    AffTabA* tabA = tabA_all[threadIdx.x >> 6];
    AffTabB* tabB = tabB_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    float* my_acc = acc + threadIdx.x;
    if (HAS_SF)
        for (int d = 0; d < a.D; ++d) my_acc[d * kBlock] = 0.f;
    const int ntab = a.P + VEC - 1;
    auto pre = [&]() {
        if (lane < ntab) {
            const int tp = lane % a.P;
            const int r = tp / a.D, d = tp - r * a.D;
            const float m = a.mask ? a.mask[r * a.mc + (a.mc == 1 ? 0 : d)] : 0.f;
            const float f = HAS_SF ? expf(a.sf[d]) : 1.f;
            const float fc = fmaxf(f, 1.f);
            AffTabA ta;
            ta.keep = 1.f - m;
            ta.keepf = (1.f - m) * f;
            ta.x3 = fc;
            ta.kscale = ta.keepf / fc;
            tabA[lane] = ta;
            AffTabB tb;
            tb.nfl = f >= 1.f ? -1.f : 0.f;
            tb.slot = d * kBlock;
            tabB[lane] = tb;
        }
        wave_lds_order();
    };
    const bool has_gs = a.g_s != nullptr, has_gt = a.g_t != nullptr;
    const float* gs_src = has_gs ? a.g_s : a.nn;
    const float* gt_src = has_gt ? a.g_t : a.nn;
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        AffParamsChunk<VEC> c;
        // only the raw scales are needed: the s halves of the interleaved (s, t) pairs
        float st[2 * VEC];
        ld_chunk<VEC, false>(a.nn + 2 * off, st);
        if (VEC > 1) ld_chunk<VEC, false>(a.nn + 2 * off + VEC, st + VEC);
        else st[1] = 0.f;
#pragma unroll
        for (int j = 0; j < VEC; ++j) c.sr[j] = st[2 * j];
        ld_chunk<VEC, false>(gs_src + off, c.gs);
        ld_chunk<VEC, false>(gt_src + off, c.gt);
        return c;
    };
    auto proc = [&](const AffParamsChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        const int ti0 = e0 - (int)fdiv((uint32_t)e0, a.div_p) * a.P;
        float gn[2 * VEC], contribs[VEC];
        int slots[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            const AffTabA ta = tabA[ti0 + j];
            const float gs = has_gs ? c.gs[j] : 0.f, gt = has_gt ? c.gt[j] : 0.f;
            float g_sr = gs * ta.keep;
            if (HAS_SF) {
                // s = tanh(s_raw / fc) f keep
                const float th = tanhf(c.sr[j] / ta.x3);
                const float sech2 = 1.f - th * th;
                g_sr = gs * sech2 * ta.kscale;
                const AffTabB tb = tabB[ti0 + j];
                const float contrib = fmaf(tb.nfl * c.sr[j], g_sr, gs * (th * ta.keepf));
                contribs[j] = contrib;
                slots[j] = tb.slot;
            }
            gn[2 * j] = g_sr;
            gn[2 * j + 1] = gt * ta.keep;
        }
        if (VEC > 1) {
            st_chunk<VEC, false>(a.g_nn + 2 * off, gn);
            st_chunk<VEC, false>(a.g_nn + 2 * off + VEC, gn + VEC);
        } else {
            st_chunk<2, false>(a.g_nn + 2 * off, gn);
        }
        if (HAS_SF) lane_private_add<VEC>(my_acc, slots, contribs, a.D >= VEC);
    };
    walk_flat_tiles<VEC, U, AffParamsChunk<VEC>>(tl, load, proc, pre);
    if (HAS_SF) wave_lane_private_reduce(acc, a.D, wave_partials_row(a.partials));
}

struct AffTransformBwdArgs {
    const float* z_out;
    const float* s;
    const float* t;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* g_s;
    float* g_t;
    int L, reverse;
};
template <int VEC>
struct AffTransformChunk {
    float zo[VEC], s[VEC], t[VEC], gzo[VEC];
    float gl;
};
template <int VEC, int U>
__global__ __launch_bounds__(kBlock) void affine_transform_bwd_kernel(AffTransformBwdArgs a, FlatTiling tl) {
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        AffTransformChunk<VEC> c;
        ld_chunk<VEC, false>(a.z_out + off, c.zo);
        ld_chunk<VEC, false>(a.s + off, c.s);
        if (a.reverse) ld_chunk<VEC, false>(a.t + off, c.t);
        if (a.g_zout) ld_chunk<VEC, false>(a.g_zout + off, c.gzo);
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (!a.reverse) c.t[j] = 0.f;
            if (!a.g_zout) c.gzo[j] = 0.f;
        }
        c.gl = a.g_ldj ? a.g_ldj[row] : 0.f;
        return c;
    };
    auto proc = [&](const AffTransformChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        float gz[VEC], gs[VEC], gt[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (!a.reverse) {
                gz[j] = c.gzo[j] * expf(c.s[j]);
                gt[j] = gz[j];
                gs[j] = c.gzo[j] * c.zo[j] + c.gl;
            } else {
                gz[j] = c.gzo[j] * expf(-c.s[j]);
                gt[j] = -c.gzo[j];
                gs[j] = -c.gzo[j] * (c.zo[j] + c.t[j]) - c.gl;
            }
        }
        st_chunk<VEC, false>(a.g_z + off, gz);
        st_chunk<VEC, false>(a.g_s + off, gs);
        st_chunk<VEC, false>(a.g_t + off, gt);
    };
    walk_flat_tiles<VEC, U, AffTransformChunk<VEC>>(tl, load, proc, NoPre());
}

// ---- ExtActNorm (activation_normalization.py:116-144) ------------------------------------------------
struct ExtBwdArgs {
    const float* z_out;
    const float* nn;        // [B,N,2D] = [bias | scales_raw]
    const float* pad;       // [B*N] nullable (weights the ldj only)
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* g_nn;
    long ntok;
    int N, D, reverse;
    FastDiv div_n;          // exact for token indices < 2^32 / N
    int fast_rows;
};
// one token's gradients: zo / gzo / cond [bias D | scales_raw D] in, gz / gcond out
template <int D, bool FAST>
__device__ __forceinline__ void ext_bwd_token(const float* zo, const float* gzo, const float* cond, float glp, bool reverse,
                                              float* gz, float* gcond) {
#pragma unroll
    for (int d = 0; d < D; ++d) {
        const float bias = cond[d];
        // th = tanh(scales_raw) = 1 - 2 r with r = 1 / (e^{2 x} + 1); sech^2 = 4 r (1 - r)
        float th, sech2;
        if (FAST) {
            const float r = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(cond[D + d] * 2.8853900817779268f) + 1.f);
            th = fmaf(-2.f, r, 1.f);
            sech2 = 4.f * r * (1.f - r);
        } else {
            th = tanhf(cond[D + d]);
            sech2 = 1.f - th * th;
        }
        float gb, gs;
        if (!reverse) {                         // z' = (z + bias) e^s ; ldj += s pad
            gz[d] = gzo[d] * bexp<FAST>(th);
            gb = gz[d];
            gs = fmaf(gzo[d], zo[d], glp);
        } else {                                // z' = z e^-s - bias ; ldj -= s pad
            gz[d] = gzo[d] * bexp<FAST>(-th);
            gb = -gzo[d];
            gs = -fmaf(gzo[d], zo[d] + bias, glp);
        }
        gcond[d] = gb;
        gcond[D + d] = gs * sech2;
    }
}
// Grouped-token form (the forward's ext_actnorm_group_kernel): one lane owns TP consecutive tokens so that its TP*D
// latents / gradients and its 2*TP*D conditioning values are whole 16-byte vectors.
template <int D, bool FAST>
__global__ __launch_bounds__(kBlock) void ext_actnorm_bwd_group_kernel(ExtBwdArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    const long ngroups = a.ntok / TP;
    for (long g = (long)blockIdx.x * kBlock + threadIdx.x; g < ngroups; g += (long)gridDim.x * kBlock) {
        const long tok0 = g * TP;
        float zo[TP * D], gzo[TP * D], cv[2 * TP * D], gz[TP * D], gc[2 * TP * D];
        const bw_f4* zs = reinterpret_cast<const bw_f4*>(a.z_out + tok0 * D);
        const bw_f4* cs = reinterpret_cast<const bw_f4*>(a.nn + tok0 * 2 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 q = __builtin_nontemporal_load(zs + v);
            zo[4 * v] = q.x; zo[4 * v + 1] = q.y; zo[4 * v + 2] = q.z; zo[4 * v + 3] = q.w;
        }
#pragma unroll
        for (int v = 0; v < 2 * NV; ++v) {
            // plain loads: a lane's consecutive 16-byte pieces share 128-byte lines (cnf_affine.hip, forward kernel)
            const bw_f4 q = cs[v];
            cv[4 * v] = q.x; cv[4 * v + 1] = q.y; cv[4 * v + 2] = q.z; cv[4 * v + 3] = q.w;
        }
        if (a.g_zout) {
            const bw_f4* gs = reinterpret_cast<const bw_f4*>(a.g_zout + tok0 * D);
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const bw_f4 q = gs[v];
                gzo[4 * v] = q.x; gzo[4 * v + 1] = q.y; gzo[4 * v + 2] = q.z; gzo[4 * v + 3] = q.w;
            }
        } else {
#pragma unroll
            for (int i = 0; i < TP * D; ++i) gzo[i] = 0.f;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            const long tok = tok0 + k;
            float glp = 0.f;
            if (a.g_ldj) {
                const long b = a.fast_rows ? (long)fdiv((uint32_t)tok, a.div_n) : tok / a.N;
                glp = a.g_ldj[b] * (a.pad ? a.pad[tok] : 1.f);
            }
            ext_bwd_token<D, FAST>(zo + k * D, gzo + k * D, cv + k * 2 * D, glp, a.reverse != 0, gz + k * D, gc + k * 2 * D);
        }
        bw_f4* dz = reinterpret_cast<bw_f4*>(a.g_z + tok0 * D);
        bw_f4* dc = reinterpret_cast<bw_f4*>(a.g_nn + tok0 * 2 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 q = {gz[4 * v], gz[4 * v + 1], gz[4 * v + 2], gz[4 * v + 3]};
            dz[v] = q;            // plain: nontemporal stores of a lane's 16-byte pieces at a TP*D*4-byte stride cost 40 -> 66 us
        }
#pragma unroll
        for (int v = 0; v < 2 * NV; ++v) {
            const bw_f4 q = {gc[4 * v], gc[4 * v + 1], gc[4 * v + 2], gc[4 * v + 3]};
            dc[v] = q;
        }
    }
    // tokens that do not fill a group
    for (long tok = ngroups * TP + (long)blockIdx.x * kBlock + threadIdx.x; tok < a.ntok; tok += (long)gridDim.x * kBlock) {
        float zo[D], gzo[D], cv[2 * D], gz[D], gc[2 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            zo[d] = a.z_out[tok * D + d];
            gzo[d] = a.g_zout ? a.g_zout[tok * D + d] : 0.f;
            cv[d] = a.nn[tok * 2 * D + d];
            cv[D + d] = a.nn[tok * 2 * D + D + d];
        }
        const float glp = a.g_ldj ? a.g_ldj[tok / a.N] * (a.pad ? a.pad[tok] : 1.f) : 0.f;
        ext_bwd_token<D, FAST>(zo, gzo, cv, glp, a.reverse != 0, gz, gc);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a.g_z[tok * D + d] = gz[d];
            a.g_nn[tok * 2 * D + d] = gc[d];
            a.g_nn[tok * 2 * D + D + d] = gc[D + d];
        }
    }
}
// Wave-tile form: a wave takes 64 consecutive token groups — one contiguous span of each tensor — with fully coalesced 16-byte
// loads (lane i takes vectors i, i + 64, ...), transposes them through its LDS strips so that lane i owns group i, and writes
// the gradients back the same way.  The group kernel above has every lane read its own 16-byte pieces at a 48 / 96-byte stride
// (D = 6): 39.7 us at the benchmark shape against this form's coalesced spans (profiles/r04_bwd_probe.txt).  The conditioning
// row of a group (2 NV vectors) lives in two strips of NV vectors per lane: a lane stride of NV vectors is free of bank conflicts
// for 128-bit reads where 2 NV is not.
template <int D, bool FAST>
__global__ __launch_bounds__(kBlock) void ext_actnorm_bwd_tile_kernel(ExtBwdArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4, NC = 2 * NV;
    __shared__ bw_f4 strip_all[kWavesPerBlock][4][kWave * NV];
    bw_f4* sz = strip_all[threadIdx.x >> 6][0];
    bw_f4* sg = strip_all[threadIdx.x >> 6][1];
    bw_f4* sc0 = strip_all[threadIdx.x >> 6][2];        // vectors [0, NV) of every group's conditioning row
    bw_f4* sc1 = strip_all[threadIdx.x >> 6][3];        // vectors [NV, 2 NV)
    const int lane = threadIdx.x & 63;
    const long ngroups = a.ntok / TP;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const bw_f4* srcz = reinterpret_cast<const bw_f4*>(a.z_out + tile * (kWave * TP * D));
        const bw_f4* srcg = reinterpret_cast<const bw_f4*>(a.g_zout + tile * (kWave * TP * D));
        const bw_f4* srcc = reinterpret_cast<const bw_f4*>(a.nn + tile * (kWave * TP * 2 * D));
        bw_f4 qz[NV], qg[NV], qc[NC];
#pragma unroll
        for (int v = 0; v < NV; ++v) qz[v] = __builtin_nontemporal_load(srcz + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NC; ++v) qc[v] = __builtin_nontemporal_load(srcc + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) qg[v] = srcg[v * kWave + lane];
        const long tok0 = (tile * kWave + lane) * TP;
        float glp[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) {
            glp[k] = 0.f;
            if (a.g_ldj) {
                const long tok = tok0 + k;
                const long b = a.fast_rows ? (long)fdiv((uint32_t)tok, a.div_n) : tok / a.N;
                glp[k] = a.g_ldj[b] * (a.pad ? a.pad[tok] : 1.f);
            }
        }
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sz[v * kWave + lane] = qz[v];
            sg[v * kWave + lane] = qg[v];
        }
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int i = v * kWave + lane, grp = i / NC, j = i - grp * NC;       // vector j of group grp's conditioning row
            (j < NV ? sc0 : sc1)[grp * NV + (j < NV ? j : j - NV)] = qc[v];
        }
        wave_lds_order();
        float zo[TP * D], gzo[TP * D], cv[2 * TP * D], gz[TP * D], gc[2 * TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = sz[lane * NV + v];
            zo[4 * v] = r.x; zo[4 * v + 1] = r.y; zo[4 * v + 2] = r.z; zo[4 * v + 3] = r.w;
            const bw_f4 q = sg[lane * NV + v];
            gzo[4 * v] = q.x; gzo[4 * v + 1] = q.y; gzo[4 * v + 2] = q.z; gzo[4 * v + 3] = q.w;
            const bw_f4 c0 = sc0[lane * NV + v];
            cv[4 * v] = c0.x; cv[4 * v + 1] = c0.y; cv[4 * v + 2] = c0.z; cv[4 * v + 3] = c0.w;
            const bw_f4 c1 = sc1[lane * NV + v];
            cv[4 * (NV + v)] = c1.x; cv[4 * (NV + v) + 1] = c1.y; cv[4 * (NV + v) + 2] = c1.z; cv[4 * (NV + v) + 3] = c1.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k)
            ext_bwd_token<D, FAST>(zo + k * D, gzo + k * D, cv + k * 2 * D, glp[k], a.reverse != 0, gz + k * D, gc + k * 2 * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {              // the lane's own group: nobody else reads these words
            const bw_f4 r = {gz[4 * v], gz[4 * v + 1], gz[4 * v + 2], gz[4 * v + 3]};
            sz[lane * NV + v] = r;
            const bw_f4 c0 = {gc[4 * v], gc[4 * v + 1], gc[4 * v + 2], gc[4 * v + 3]};
            sc0[lane * NV + v] = c0;
            const bw_f4 c1 = {gc[4 * (NV + v)], gc[4 * (NV + v) + 1], gc[4 * (NV + v) + 2], gc[4 * (NV + v) + 3]};
            sc1[lane * NV + v] = c1;
        }
        wave_lds_order();
        bw_f4* dz = reinterpret_cast<bw_f4*>(a.g_z + tile * (kWave * TP * D));
        bw_f4* dc = reinterpret_cast<bw_f4*>(a.g_nn + tile * (kWave * TP * 2 * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) st_chunk<4, kNtOut>(reinterpret_cast<float*>(dz + v * kWave + lane), reinterpret_cast<const float*>(&sz[v * kWave + lane]));
#pragma unroll
        for (int v = 0; v < NC; ++v) {
            const int i = v * kWave + lane, grp = i / NC, j = i - grp * NC;
            st_chunk<4, kNtOut>(reinterpret_cast<float*>(dc + i), reinterpret_cast<const float*>(&(j < NV ? sc0 : sc1)[grp * NV + (j < NV ? j : j - NV)]));
        }
        wave_lds_order();                            // the strips are refilled by the next tile
    }
    // tokens that do not fill a wave tile
    for (long tok = ntiles * kWave * TP + (long)blockIdx.x * kBlock + threadIdx.x; tok < a.ntok; tok += (long)gridDim.x * kBlock) {
        float zo[D], gzo[D], cv[2 * D], gz[D], gc[2 * D];
#pragma unroll
        for (int d = 0; d < D; ++d) {
            zo[d] = a.z_out[tok * D + d];
            gzo[d] = a.g_zout[tok * D + d];
            cv[d] = a.nn[tok * 2 * D + d];
            cv[D + d] = a.nn[tok * 2 * D + D + d];
        }
        const float glp = a.g_ldj ? a.g_ldj[tok / a.N] * (a.pad ? a.pad[tok] : 1.f) : 0.f;
        ext_bwd_token<D, FAST>(zo, gzo, cv, glp, a.reverse != 0, gz, gc);
#pragma unroll
        for (int d = 0; d < D; ++d) {
            a.g_z[tok * D + d] = gz[d];
            a.g_nn[tok * 2 * D + d] = gc[d];
            a.g_nn[tok * 2 * D + D + d] = gc[D + d];
        }
    }
}
// any D / unaligned tensors: one lane per element
template <bool FAST>
__global__ __launch_bounds__(kBlock) void ext_actnorm_bwd_elem_kernel(ExtBwdArgs a) {
    const long total = a.ntok * a.D;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const long tok = e / a.D;
        const int d = (int)(e - tok * a.D);
        const float cond[2] = {a.nn[tok * 2 * a.D + d], a.nn[tok * 2 * a.D + a.D + d]};
        const float zo = a.z_out[e], gzo = a.g_zout ? a.g_zout[e] : 0.f;
        const float glp = a.g_ldj ? a.g_ldj[tok / a.N] * (a.pad ? a.pad[tok] : 1.f) : 0.f;
        float gz, gc[2];
        ext_bwd_token<1, FAST>(&zo, &gzo, cond, glp, a.reverse != 0, &gz, gc);
        a.g_z[e] = gz;
        a.g_nn[tok * 2 * a.D + d] = gc[0];
        a.g_nn[tok * 2 * a.D + a.D + d] = gc[1];
    }
}

// ---- ActNorm (activation_normalization.py:24-48) ---------------------------------------------------
struct ActBwdArgs {
    const float* z_out;
    const float* bias;
    const float* scales;
    const float* pad;
    const float* length;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* partials;        // [gridDim.x, 2D]: d bias | d scales
    int B, N, D, L, reverse;
    FastDiv div_d;
    long ntok;
};
template <int VEC>
struct ActBwdChunk {
    float zo[VEC], gzo[VEC], pv[VEC];
};
template <int VEC, int U, bool FAST, bool HAS_PAD>
__global__ __launch_bounds__(kBlock) void actnorm_bwd_kernel(ActBwdArgs a, FlatTiling tl) {
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [2D][kBlock] lane-private words: d bias | d scales
    // period D plus its first VEC - 1 entries again, one private copy per wave (no workgroup barrier)
    __shared__ float sb_all[kWavesPerBlock][kBwdMaxD + 3], se_all[kWavesPerBlock][kBwdMaxD + 3];
    float* sb = sb_all[threadIdx.x >> 6];
    float* se = se_all[threadIdx.x >> 6];
    const int lane = threadIdx.x & 63;
    float* my_acc = acc + threadIdx.x;
    for (int p = 0; p < 2 * a.D; ++p) my_acc[p * kBlock] = 0.f;
    for (int i = lane; i < a.D + VEC - 1; i += kWave) {
        const int d = i % a.D;
        sb[i] = a.bias[d];
        se[i] = bexp<FAST>(a.reverse ? -a.scales[d] : a.scales[d]);
    }
    wave_lds_order();
    const bool has_gz = a.g_zout != nullptr;
    const float* gz_src = has_gz ? a.g_zout : a.z_out;
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        ActBwdChunk<VEC> c;
        ld_chunk<VEC, kNtSaved>(a.z_out + off, c.zo);
        ld_chunk<VEC, kNtUp>(gz_src + off, c.gzo);
        if (HAS_PAD) {
            int n = (int)fdiv((uint32_t)e0, a.div_d), d = e0 - n * a.D;
#pragma unroll
            for (int j = 0; j < VEC; ++j) {
                c.pv[j] = a.pad[(size_t)row * a.N + n];
                if (++d == a.D) {
                    d = 0;
                    ++n;
                }
            }
        }
        return c;
    };
    auto proc = [&](const ActBwdChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        const int d0 = e0 - (int)fdiv((uint32_t)e0, a.div_d) * a.D;
        float gz[VEC], vals[2 * VEC];
        int slots[2 * VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            float gzo = has_gz ? c.gzo[j] : 0.f, y = c.zo[j];
            if (HAS_PAD) {
                gzo *= c.pv[j];                                 // z' = (...) * pad
                // the pre-padding output: z'/pad where pad != 0 (padded positions have gzo = 0 anyway)
                y = c.pv[j] != 0.f ? c.zo[j] / c.pv[j] : 0.f;
            }
            float gb, gs;
            gz[j] = gzo * se[d0 + j];
            if (!a.reverse) {                       // y = (z + b) e^sc
                gb = gz[j];
                gs = gzo * y;
            } else {                                // y = z e^-sc - b
                gb = -gzo;
                gs = -gzo * (y + sb[d0 + j]);
            }
            int d = d0 + j;
            d -= (int)fdiv((uint32_t)d, a.div_d) * a.D;
            slots[j] = d * kBlock;
            slots[VEC + j] = (a.D + d) * kBlock;
            vals[j] = gb;
            vals[VEC + j] = gs;
        }
        st_chunk<VEC, kNtOut>(a.g_z + off, gz);
        lane_private_add<2 * VEC>(my_acc, slots, vals, a.D >= VEC);
    };
    walk_flat_tiles<VEC, U, ActBwdChunk<VEC>>(tl, load, proc, NoPre());
    // log-det term: ldj += (+-sum_d scales) * len_b  ->  d scales[d] += +-sum_b g_ldj[b] len_b: this lane's share of the samples
    float part = 0.f;
    if (a.g_ldj) {
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
            float len;
            if (a.length) len = a.length[b];
            else if (a.pad) {
                len = 0.f;
                for (int n = 0; n < a.N; ++n) len += a.pad[b * a.N + n];
            } else len = (float)a.N;
            part += a.g_ldj[b] * len;
        }
        if (a.reverse) part = -part;
    }
    const WaveRow out_row = wave_partials_row(a.partials);
    wave_lane_private_reduce(acc, 2 * a.D, out_row);
    // entry 2D of the row: the log-det term, added to every channel's d scales by the reduction launch
    part = rows_total_in_lane63(row16_sum(part));
    if (lane == kWave - 1) out_row[2 * a.D] = part;
}

// Token-owner form for D in {1..6, 8} (as the 1x1 convolution below): a wave takes spans of 64 token groups with coalesced
// 16-byte loads, transposes them through its LDS strips so that a lane owns whole tokens, and keeps the 2D parameter-gradient
// sums in REGISTERS (a lane's register d always belongs to channel d: no LDS read-add-write per element, which is what the
// flat-tile kernel above pays).  16.5 -> us at the benchmark shape: profiles/r04_bwd_probe.txt.
template <int D, bool FAST>
__global__ __launch_bounds__(kBlock) void actnorm_bwd_tile_kernel(ActBwdArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    constexpr int R = 2 * D + 1;
    __shared__ bw_f4 strip_all[kWavesPerBlock][2][kWave * NV];
    __shared__ float comb[kWavesPerBlock][4 * R];
    bw_f4* sx = strip_all[threadIdx.x >> 6][0];
    bw_f4* sg = strip_all[threadIdx.x >> 6][1];
    const int lane = threadIdx.x & 63;
    float bk[D], ek[D], acc[R];
#pragma unroll
    for (int i = 0; i < D; ++i) {
        bk[i] = a.bias[i];
        ek[i] = bexp<FAST>(a.reverse ? -a.scales[i] : a.scales[i]);
    }
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    const bool reverse = a.reverse != 0;
    auto token = [&](const float* zo, const float* gin, float p, float* gz) {
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float gzo = gin[i], y = zo[i];
            if (a.pad) {
                gzo *= p;                                    // z' = (...) * pad
                y = p != 0.f ? zo[i] / p : 0.f;              // the pre-padding output (padded positions have gzo = 0 anyway)
            }
            gz[i] = gzo * ek[i];
            if (!reverse) {                                  // y = (z + b) e^sc
                acc[i] += gz[i];
                acc[D + i] = fmaf(gzo, y, acc[D + i]);
            } else {                                         // y = z e^-sc - b
                acc[i] -= gzo;
                acc[D + i] = fmaf(-gzo, y + bk[i], acc[D + i]);
            }
        }
    };
    const long ngroups = a.ntok / TP;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const bw_f4* srcx = reinterpret_cast<const bw_f4*>(a.z_out + tile * (kWave * TP * D));
        const bw_f4* srcg = reinterpret_cast<const bw_f4*>(a.g_zout + tile * (kWave * TP * D));
        bw_f4 qx[NV], qg[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) qx[v] = __builtin_nontemporal_load(srcx + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) qg[v] = srcg[v * kWave + lane];
        const long g = tile * kWave + lane;
        float pv[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) pv[k] = a.pad ? a.pad[g * TP + k] : 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sx[v * kWave + lane] = qx[v];
            sg[v * kWave + lane] = qg[v];
        }
        wave_lds_order();
        float xin[TP * D], gin[TP * D], gx[TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = sx[lane * NV + v];
            xin[4 * v] = r.x; xin[4 * v + 1] = r.y; xin[4 * v + 2] = r.z; xin[4 * v + 3] = r.w;
            const bw_f4 q = sg[lane * NV + v];
            gin[4 * v] = q.x; gin[4 * v + 1] = q.y; gin[4 * v + 2] = q.z; gin[4 * v + 3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) token(xin + k * D, gin + k * D, pv[k], gx + k * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = {gx[4 * v], gx[4 * v + 1], gx[4 * v + 2], gx[4 * v + 3]};
            sx[lane * NV + v] = r;                  // the lane's own group: nobody else reads these words
        }
        wave_lds_order();
        bw_f4* dst = reinterpret_cast<bw_f4*>(a.g_z + tile * (kWave * TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) st_chunk<4, kNtOut>(reinterpret_cast<float*>(dst + v * kWave + lane), reinterpret_cast<const float*>(&sx[v * kWave + lane]));
        wave_lds_order();                            // the strips are refilled by the next tile
    }
    // tokens that do not fill a wave tile
    for (long t = ntiles * kWave * TP + (long)blockIdx.x * kBlock + threadIdx.x; t < a.ntok; t += (long)gridDim.x * kBlock) {
        float xin[D], gin[D], gx[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xin[i] = a.z_out[t * D + i];
            gin[i] = a.g_zout[t * D + i];
        }
        token(xin, gin, a.pad ? a.pad[t] : 1.f, gx);
#pragma unroll
        for (int i = 0; i < D; ++i) a.g_z[t * D + i] = gx[i];
    }
    // log-det term (see the flat-tile kernel): entry 2D of the row
    if (a.g_ldj) {
        float part = 0.f;
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
            float len;
            if (a.length) len = a.length[b];
            else if (a.pad) {
                len = 0.f;
                for (int n = 0; n < a.N; ++n) len += a.pad[b * a.N + n];
            } else len = (float)a.N;
            part = fmaf(a.g_ldj[b], len, part);
        }
        acc[2 * D] = reverse ? -part : part;
    }
    wave_register_reduce<R>(acc, comb[threadIdx.x >> 6], wave_partials_row(a.partials));
}

// ---- invertible 1x1 convolution (permutation_layers.py:106-136) -------------------------------------------
struct ConvBwdArgs {
    const float* x;
    const float* w;         // the matrix that was applied (W or W^-1), [D,D]
    const float* pad;
    const float* length;
    const float* g_zout;
    const float* g_ldj;
    float* g_x;
    float* partials;        // [gridDim.x, D*D + 1]: dW | d sldj
    long ntok;
    int B, N, D, reverse;
};
// d sldj = +- sum_b g_ldj[b] len_b: this thread's share of the samples
__device__ __forceinline__ float conv_sldj_part(const ConvBwdArgs& a) {
    float part = 0.f;
    if (a.g_ldj)
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock)
            part += a.g_ldj[b] * (a.length ? a.length[b] : (float)a.N);
    return a.reverse ? -part : part;
}
// A lane owns TP consecutive tokens (their TP*D floats are whole 16-byte vectors); wave tiles of 64 groups are loaded
// fully coalesced and transposed through the wave's LDS strip, as in the forward's actnorm_invconv_kernel.  The D x D
// weight-gradient partial sums stay in registers for all of a wave's tiles.
template <int D>
__global__ __launch_bounds__(kBlock) void invconv_bwd_kernel(ConvBwdArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    constexpr int R = D * D + 1;
    __shared__ bw_f4 strip_all[kWavesPerBlock][2][kWave * NV];
    __shared__ float comb[kWavesPerBlock][4 * R];
    bw_f4* sx = strip_all[threadIdx.x >> 6][0];
    bw_f4* sg = strip_all[threadIdx.x >> 6][1];
    const int lane = threadIdx.x & 63;
    float wk[D * D], dw[R];
#pragma unroll
    for (int i = 0; i < D * D; ++i) {
        wk[i] = a.w[i];
        dw[i] = 0.f;
    }
    auto token = [&](const float* xv, const float* gin, float p, float* gx) {
        float g[D];
#pragma unroll
        for (int j = 0; j < D; ++j) g[j] = a.pad ? gin[j] * p : gin[j];         // z' = (x @ W) pad
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float acc = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                acc = fmaf(g[j], wk[i * D + j], acc);                            // g_x[t,i] = sum_j g[t,j] W[i,j]
                dw[i * D + j] = fmaf(xv[i], g[j], dw[i * D + j]);                // dW[i,j] += x[t,i] g[t,j]
            }
            gx[i] = acc;
        }
    };
    const bool aligned = a.g_zout && ((reinterpret_cast<uintptr_t>(a.x) | reinterpret_cast<uintptr_t>(a.g_zout) |
                                       reinterpret_cast<uintptr_t>(a.g_x)) & 15) == 0;
    const long ngroups = aligned ? a.ntok / TP : 0;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const bw_f4* srcx = reinterpret_cast<const bw_f4*>(a.x + tile * (kWave * TP * D));
        const bw_f4* srcg = reinterpret_cast<const bw_f4*>(a.g_zout + tile * (kWave * TP * D));
        bw_f4 qx[NV], qg[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) qx[v] = __builtin_nontemporal_load(srcx + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) qg[v] = srcg[v * kWave + lane];
        const long g = tile * kWave + lane;
        float pv[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) pv[k] = a.pad ? a.pad[g * TP + k] : 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sx[v * kWave + lane] = qx[v];
            sg[v * kWave + lane] = qg[v];
        }
        wave_lds_order();
        float xin[TP * D], gin[TP * D], gx[TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = sx[lane * NV + v];
            xin[4 * v] = r.x; xin[4 * v + 1] = r.y; xin[4 * v + 2] = r.z; xin[4 * v + 3] = r.w;
            const bw_f4 s = sg[lane * NV + v];
            gin[4 * v] = s.x; gin[4 * v + 1] = s.y; gin[4 * v + 2] = s.z; gin[4 * v + 3] = s.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) token(xin + k * D, gin + k * D, pv[k], gx + k * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = {gx[4 * v], gx[4 * v + 1], gx[4 * v + 2], gx[4 * v + 3]};
            sx[lane * NV + v] = r;                  // the lane's own group: nobody else reads these words
        }
        wave_lds_order();
        bw_f4* dst = reinterpret_cast<bw_f4*>(a.g_x + tile * (kWave * TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) st_chunk<4, kNtOut>(reinterpret_cast<float*>(dst + v * kWave + lane), reinterpret_cast<const float*>(&sx[v * kWave + lane]));
        wave_lds_order();                            // the strips are refilled by the next tile
    }
    // tokens that do not fill a wave tile (or everything, for unaligned tensors / a missing upstream gradient)
    for (long t = ntiles * kWave * TP + (long)blockIdx.x * kBlock + threadIdx.x; t < a.ntok; t += (long)gridDim.x * kBlock) {
        float xin[D], gin[D], gx[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xin[i] = a.x[t * D + i];
            gin[i] = a.g_zout ? a.g_zout[t * D + i] : 0.f;
        }
        token(xin, gin, a.pad ? a.pad[t] : 1.f, gx);
#pragma unroll
        for (int i = 0; i < D; ++i) a.g_x[t * D + i] = gx[i];
    }
    dw[D * D] = conv_sldj_part(a);
    wave_register_reduce<R>(dw, comb[threadIdx.x >> 6], wave_partials_row(a.partials));
}
// any D <= 12: one lane per (token, input channel i), weight-gradient partial sums in lane-private LDS words
__global__ __launch_bounds__(kBlock) void invconv_bwd_generic_kernel(ConvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float acc[];     // [D*D + 1][kBlock]
    const int D = a.D, P = D * D + 1;
    for (int p = 0; p < P; ++p) acc[p * kBlock + threadIdx.x] = 0.f;
    float* my_acc = acc + threadIdx.x;
    const long total = a.ntok * D;
    for (long e = (long)blockIdx.x * kBlock + threadIdx.x; e < total; e += (long)gridDim.x * kBlock) {
        const long t = e / D;
        const int i = (int)(e - t * D);
        const float pv = a.pad ? a.pad[t] : 1.f;
        const float xi = a.x[e];
        float gx = 0.f;
        for (int j = 0; j < D; ++j) {
            const float g = (a.g_zout ? a.g_zout[t * D + j] : 0.f) * pv;
            gx = fmaf(g, a.w[i * D + j], gx);
            my_acc[(i * D + j) * kBlock] += xi * g;
        }
        a.g_x[e] = gx;
    }
    const float part = conv_sldj_part(a);
    if (part != 0.f) my_acc[D * D * kBlock] += part;
    wave_lane_private_reduce(acc, P, wave_partials_row(a.partials));
}

// ---- ActNorm + 1x1 convolution of one flow step, ONE backward kernel (forward direction; the forward is actnorm_invconv_kernel,
// cnf_linear.hip).  The layer pair's intermediate y = (z + b) e^s never went to HBM in the forward and does not here: it is
// recomputed per token — from the pair's INPUT z (FROM_OUT = false: the forward's own arithmetic, the same bits) or, where
// only the pair's OUTPUT exists (the pair was fused behind a coupling layer or the encoder, whose output stayed in
// registers), from out @ W^-1 (FROM_OUT = true).  12 bytes per element (saved tensor, upstream gradient, gradient out) against
// the 24 of the two layers' kernels run one after the other.  activation_normalization.py:24-48, permutation_layers.py:106-136.
struct ActConvBwdArgs {
    const float* saved;     // FROM_OUT ? the pair's output : the pair's input
    const float* bias;
    const float* scales;
    const float* w;         // [D,D]
    const float* w_inv;     // [D,D], FROM_OUT only
    const float* pad;
    const float* length;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    float* partials;        // rows of [dW (D*D) | d sldj | d bias (D) | d scales (D) | ActNorm's log-det term]
    long ntok;
    int B, N, D;
};
// W^-1 in fp64 by one wave (small_inverse_wg, cnf_common.h)
__global__ __launch_bounds__(kWave) void small_inverse_kernel(const float* w, float* w_inv, int D) {
    __shared__ double A[8][17];
    small_inverse_wg(w, w_inv, D, threadIdx.x, A);
}
template <int D, bool FROM_OUT>
__global__ __launch_bounds__(kBlock) void actconv_bwd_kernel(ActConvBwdArgs a) {
    constexpr int TP = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    constexpr int NV = TP * D / 4;
    constexpr int R = D * D + 2 * D + 2;
    constexpr int kSl = D * D, kB0 = D * D + 1, kS0 = D * D + 1 + D, kLa = D * D + 1 + 2 * D;
    __shared__ bw_f4 strip_all[kWavesPerBlock][2][kWave * NV];
    __shared__ float comb[kWavesPerBlock][4 * R];
    bw_f4* sx = strip_all[threadIdx.x >> 6][0];
    bw_f4* sg = strip_all[threadIdx.x >> 6][1];
    const int lane = threadIdx.x & 63;
    float wk[D * D], wi[FROM_OUT ? D * D : 1], bk[D], ek[D], acc[R];
#pragma unroll
    for (int i = 0; i < D * D; ++i) {
        wk[i] = a.w[i];
        if (FROM_OUT) wi[i] = a.w_inv[i];
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        bk[i] = a.bias[i];
        ek[i] = expf(a.scales[i]);              // the forward's expf (actnorm_invconv_kernel): the recomputed y has its bits
    }
#pragma unroll
    for (int i = 0; i < R; ++i) acc[i] = 0.f;
    auto token = [&](const float* sv, const float* gin, float p, float* gz) {
        float u[D], y[D], g[D];                  // u = (z + b) e^s, y = u pad: what the convolution was given
        if (!FROM_OUT) {
#pragma unroll
            for (int i = 0; i < D; ++i) {
                u[i] = (sv[i] + bk[i]) * ek[i];
                y[i] = a.pad ? u[i] * p : u[i];
            }
        } else {
            // out = (y @ W) pad: y = (out / pad) @ W^-1 where pad != 0 (elsewhere y = 0 and no gradient passes)
            const float ip = a.pad ? (p != 0.f ? 1.f / p : 0.f) : 1.f;
#pragma unroll
            for (int i = 0; i < D; ++i) {
                float t = 0.f;
#pragma unroll
                for (int j = 0; j < D; ++j) t = fmaf(a.pad ? sv[j] * ip : sv[j], wi[j * D + i], t);
                y[i] = t;
                u[i] = a.pad ? t * ip : t;
            }
        }
#pragma unroll
        for (int j = 0; j < D; ++j) g[j] = a.pad ? gin[j] * p : gin[j];         // out = (y @ W) pad
#pragma unroll
        for (int i = 0; i < D; ++i) {
            float gy = 0.f;
#pragma unroll
            for (int j = 0; j < D; ++j) {
                gy = fmaf(g[j], wk[i * D + j], gy);                              // g_y[i] = sum_j g[j] W[i,j]
                acc[i * D + j] = fmaf(y[i], g[j], acc[i * D + j]);               // dW[i,j] += y[i] g[j]
            }
            const float gu = a.pad ? gy * p : gy;                                // y = u pad
            const float gzv = gu * ek[i];
            gz[i] = gzv;
            acc[kB0 + i] += gzv;                                                 // d bias
            acc[kS0 + i] = fmaf(gu, u[i], acc[kS0 + i]);                         // d scales
        }
    };
    const bool aligned = a.g_zout && ((reinterpret_cast<uintptr_t>(a.saved) | reinterpret_cast<uintptr_t>(a.g_zout) |
                                       reinterpret_cast<uintptr_t>(a.g_z)) & 15) == 0;
    const long ngroups = aligned ? a.ntok / TP : 0;
    const long ntiles = ngroups / kWave;
    const long wave_id = (long)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6);
    const long nwaves = (long)gridDim.x * kWavesPerBlock;
    for (long tile = wave_id; tile < ntiles; tile += nwaves) {
        const bw_f4* srcx = reinterpret_cast<const bw_f4*>(a.saved + tile * (kWave * TP * D));
        const bw_f4* srcg = reinterpret_cast<const bw_f4*>(a.g_zout + tile * (kWave * TP * D));
        bw_f4 qx[NV], qg[NV];
#pragma unroll
        for (int v = 0; v < NV; ++v) qx[v] = __builtin_nontemporal_load(srcx + v * kWave + lane);
#pragma unroll
        for (int v = 0; v < NV; ++v) qg[v] = srcg[v * kWave + lane];
        const long g = tile * kWave + lane;
        float pv[TP];
#pragma unroll
        for (int k = 0; k < TP; ++k) pv[k] = a.pad ? a.pad[g * TP + k] : 1.f;
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            sx[v * kWave + lane] = qx[v];
            sg[v * kWave + lane] = qg[v];
        }
        wave_lds_order();
        float xin[TP * D], gin[TP * D], gx[TP * D];
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = sx[lane * NV + v];
            xin[4 * v] = r.x; xin[4 * v + 1] = r.y; xin[4 * v + 2] = r.z; xin[4 * v + 3] = r.w;
            const bw_f4 q = sg[lane * NV + v];
            gin[4 * v] = q.x; gin[4 * v + 1] = q.y; gin[4 * v + 2] = q.z; gin[4 * v + 3] = q.w;
        }
#pragma unroll
        for (int k = 0; k < TP; ++k) token(xin + k * D, gin + k * D, pv[k], gx + k * D);
#pragma unroll
        for (int v = 0; v < NV; ++v) {
            const bw_f4 r = {gx[4 * v], gx[4 * v + 1], gx[4 * v + 2], gx[4 * v + 3]};
            sx[lane * NV + v] = r;                  // the lane's own group: nobody else reads these words
        }
        wave_lds_order();
        bw_f4* dst = reinterpret_cast<bw_f4*>(a.g_z + tile * (kWave * TP * D));
#pragma unroll
        for (int v = 0; v < NV; ++v) st_chunk<4, kNtOut>(reinterpret_cast<float*>(dst + v * kWave + lane), reinterpret_cast<const float*>(&sx[v * kWave + lane]));
        wave_lds_order();                            // the strips are refilled by the next tile
    }
    // tokens that do not fill a wave tile (or everything, for unaligned tensors / a missing upstream gradient)
    for (long t = ntiles * kWave * TP + (long)blockIdx.x * kBlock + threadIdx.x; t < a.ntok; t += (long)gridDim.x * kBlock) {
        float xin[D], gin[D], gx[D];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            xin[i] = a.saved[t * D + i];
            gin[i] = a.g_zout ? a.g_zout[t * D + i] : 0.f;
        }
        token(xin, gin, a.pad ? a.pad[t] : 1.f, gx);
#pragma unroll
        for (int i = 0; i < D; ++i) a.g_z[t * D + i] = gx[i];
    }
    // log-det terms: ldj += (sum_d scales) len_a + sldj len_c with the forward's two lengths (ActNorm: length | sum(pad) | N,
    // convolution: length | N): this thread's share of the samples
    if (a.g_ldj) {
        float pa = 0.f, pc = 0.f;
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock) {
            float len_a, len_c;
            if (a.length) len_a = len_c = a.length[b];
            else {
                len_c = (float)a.N;
                if (a.pad) {
                    len_a = 0.f;
                    for (int n = 0; n < a.N; ++n) len_a += a.pad[b * a.N + n];
                } else len_a = (float)a.N;
            }
            pa = fmaf(a.g_ldj[b], len_a, pa);
            pc = fmaf(a.g_ldj[b], len_c, pc);
        }
        acc[kSl] = pc;
        acc[kLa] = pa;
    }
    wave_register_reduce<R>(acc, comb[threadIdx.x >> 6], wave_partials_row(a.partials));
}

// ---- logistic log-prob / NLL (distributions.py:129-163; set_modeling/task.py:96-118) ------------------------
// d/dx [-(softplus(v) + softplus(-v) + log sigma)] = -tanh(v/2) / sigma,  v = (x - mu)/sigma;  tanh(v/2) = 1 - 2/(e^v + 1)
template <bool FAST>
__device__ __forceinline__ float tanh_half(float v) {
    if (!FAST) return tanhf(0.5f * v);
    return fmaf(-2.f, rcp_exp_p1<true>(v), 1.f);
}
template <bool FAST>
__global__ __launch_bounds__(kBlock) void logistic_log_prob_bwd_kernel(const float* x, const float* g_out,
                                                                       float* g_x, long n, float mu,
                                                                       float sigma) {
    const float inv_sigma = 1.f / sigma;
    const bool vec = ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(g_out) | reinterpret_cast<uintptr_t>(g_x)) & 15) == 0;
    const long n4 = vec ? n >> 2 : 0;
    for (long i = (long)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (long)gridDim.x * kBlock) {
        float xv[4], gv[4], o[4];
        ld_chunk<4, kNtSaved>(x + 4 * i, xv);
        ld_chunk<4, kNtUp>(g_out + 4 * i, gv);
#pragma unroll
        for (int j = 0; j < 4; ++j) o[j] = gv[j] * (-tanh_half<FAST>((xv[j] - mu) * inv_sigma) * inv_sigma);
        st_chunk<4, kNtOut>(g_x + 4 * i, o);
    }
    for (long i = (n4 << 2) + (long)blockIdx.x * kBlock + threadIdx.x; i < n; i += (long)gridDim.x * kBlock)
        g_x[i] = g_out[i] * (-tanh_half<FAST>((x[i] - mu) * inv_sigma) * inv_sigma);
}

struct NllBwdArgs {
    const float* z;
    const float* pad;
    const float* length;
    const float* g_nll;     // [B]
    float* g_z;
    float* g_ldj;           // [B] nullable
    int B, N, D, L;
    float inv_sigma;
    FastDiv div_d;
};
template <int VEC>
struct NllBwdChunk {
    float z[VEC], pv[VEC];
    float gscale;           // g_nll[b] / len_b / sigma
};
// nll_b = (-ldj_b - sum logp(z) pad) / len_b
template <int VEC, int U, bool FAST>
__global__ __launch_bounds__(kBlock) void prior_nll_bwd_kernel(NllBwdArgs a, FlatTiling tl) {
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        NllBwdChunk<VEC> c;
        ld_chunk<VEC, kNtSaved>(a.z + off, c.z);
        int n = (int)fdiv((uint32_t)e0, a.div_d), d = e0 - n * a.D;
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            c.pv[j] = a.pad ? a.pad[(size_t)row * a.N + n] : 1.f;
            if (++d == a.D) {
                d = 0;
                ++n;
            }
        }
        c.gscale = a.g_nll[row] / (a.length ? a.length[row] : (float)a.N) * a.inv_sigma;
        return c;
    };
    auto proc = [&](const NllBwdChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) g[j] = c.gscale * c.pv[j] * tanh_half<FAST>(c.z[j] * a.inv_sigma);
        st_chunk<VEC, kNtOut>(a.g_z + off, g);
    };
    walk_flat_tiles<VEC, U, NllBwdChunk<VEC>>(tl, load, proc, NoPre());
    if (a.g_ldj)
        for (long b = (long)blockIdx.x * kBlock + threadIdx.x; b < a.B; b += (long)gridDim.x * kBlock)
            a.g_ldj[b] = -a.g_nll[b] / (a.length ? a.length[b] : (float)a.N);
}

// ---- sigmoid / logit flow (sigmoid_layer.py:24-47) -----------------------------------------------------------
struct SigBwdArgs {
    const float* z_in;
    const float* g_zout;
    const float* g_ldj;
    float* g_z;
    int L, reverse;
    float alpha;
};
template <int VEC>
struct SigBwdChunk {
    float x[VEC], gzo[VEC];
    float gl;
};
template <int VEC, int U, bool FAST>
__global__ __launch_bounds__(kBlock) void sigmoid_flow_bwd_kernel(SigBwdArgs a, FlatTiling tl) {
    auto load = [&](int row, int e0) {
        const size_t off = (size_t)row * a.L + e0;
        SigBwdChunk<VEC> c;
        ld_chunk<VEC, kNtSaved>(a.z_in + off, c.x);
        if (a.g_zout) ld_chunk<VEC, kNtUp>(a.g_zout + off, c.gzo);
#pragma unroll
        for (int j = 0; j < VEC; ++j)
            if (!a.g_zout) c.gzo[j] = 0.f;
        c.gl = a.g_ldj ? a.g_ldj[row] : 0.f;
        return c;
    };
    auto proc = [&](const SigBwdChunk<VEC>& c, int row, int e0, int) {
        const size_t off = (size_t)row * a.L + e0;
        float g[VEC];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
            if (!a.reverse) {                       // y = sigmoid(x), ldj += -x - 2 softplus(-x)
                // s = sigmoid(x) = 1 - 1/(e^x + 1);  s (1 - s) = r (1 - r),  1 - 2 s = 2 r - 1
                const float r = rcp_exp_p1<FAST>(c.x[j]);
                g[j] = c.gzo[j] * (r * (1.f - r)) + c.gl * fmaf(2.f, r, -1.f);
            } else {                                // u = x(1-a)+a/2 ; y = log u - log(1-u) ; ldj += -log u - log(1-u) + c
                const float u = c.x[j] * (1.f - a.alpha) + a.alpha * 0.5f;
                const float inv = FAST ? __builtin_amdgcn_rcpf(u * (1.f - u)) : 1.f / (u * (1.f - u));
                g[j] = (c.gzo[j] * inv + c.gl * (2.f * u - 1.f) * inv) * (1.f - a.alpha);
            }
        }
        st_chunk<VEC, kNtOut>(a.g_z + off, g);
    };
    walk_flat_tiles<VEC, U, SigBwdChunk<VEC>>(tl, load, proc, NoPre());
}

// rows = waves of the launch that wrote the partials
static int reduce_partials(const float* partials, int rows, int P, float* out, hipStream_t st) {
    enqueue_reduce(P, partials, rows, P, out, (float*)nullptr, P, -1, 0, st);
    return CNF_OK;
}
// chunks in flight per lane / chunk groups per wave: the knob (cnf_set_bwd_tile) when set, else the kernel's own default
static inline int bwd_u(int dflt) {
    const int u = g_bwd_u.load(std::memory_order_relaxed);
    return u >= 1 && u <= 3 ? u : dflt;
}
static inline int bwd_g(int dflt) {
    const int g = g_bwd_g.load(std::memory_order_relaxed);
    return g >= 1 ? g : dflt;
}
static inline int stream_grid(long n) {
    return (int)std::min<long>(std::max<long>((n + kBlock - 1) / kBlock, 1), 1 << 22);
}

}  // namespace cnf

using namespace cnf;

// chunk width x chunks in flight -> compile-time constants: f(integral_constant<VEC>, integral_constant<U>)
template <int X>
using ic = std::integral_constant<int, X>;
template <typename F>
static void dispatch_vec_u(int vec, int U, F&& f) {
    if (vec == 4) {
        if (U == 1) f(ic<4>{}, ic<1>{});
        else if (U == 3) f(ic<4>{}, ic<3>{});
        else f(ic<4>{}, ic<2>{});
    } else if (vec == 2) {
        f(ic<2>{}, ic<2>{});
    } else {
        f(ic<1>{}, ic<2>{});
    }
}
#define BWD_VU(V_, U_) constexpr int V = decltype(V_)::value, UU = decltype(U_)::value; (void)UU

template <int V, int UU>
static void launch_affine_bwd(const AffBwdArgs& a, const FlatTiling& tl, bool has_sf, bool reverse, hipStream_t st) {
    const dim3 grid = flat_grid(tl), block(kBlock);
    const size_t lds = has_sf ? (size_t)a.D * kBlock * sizeof(float) : 0;
    const bool fast = math_mode() == 1;
#define AFF_BWD(SF, REV)                                                                                            \
    do {                                                                                                            \
        if (fast) CNF_LAUNCH((affine_bwd_kernel<V, UU, SF, REV, true>), grid, block, lds, st, a, tl);               \
        else CNF_LAUNCH((affine_bwd_kernel<V, UU, SF, REV, false>), grid, block, lds, st, a, tl);                   \
    } while (0)
    if (has_sf) {
        if (reverse) AFF_BWD(true, true);
        else AFF_BWD(true, false);
    } else {
        if (reverse) AFF_BWD(false, true);
        else AFF_BWD(false, false);
    }
#undef AFF_BWD
}

extern "C" {

/* floats the caller must provide as `workspace` to the backward entry points that return parameter gradients: rows of
 * `param_count` (+ 1) partial sums — one per wave of the streaming kernels of this file (their rows are at most
 * kBwdMaxRowP wide), one per workgroup (at most 1024, twice for the mixture kernel and its fix-up launch) for the
 * mixture / encoder kernels — + one reduced row + the fix-up launch's flag words */
void cnf_bwd_defer_begin(void) {
    // jobs left over from a batch that was never flushed (a host that failed between begin and flush) are reduced now rather
    // than dropped: their outputs would stay uninitialised
    flush_reduce_jobs();
    g_defer.on = true;
}
int cnf_bwd_defer_flush(cnf_stream_t stream) {
    // the queued reductions run on the stream their kernels ran on (`stream` is checked against it, not trusted over it)
    const bool mismatch = g_defer.jobs.n > 0 && (hipStream_t)stream != g_defer.stream;
    flush_reduce_jobs();
    g_defer.on = false;
    if (mismatch) {
        set_error("cnf_bwd_defer_flush: the queued calls ran on another stream than the one given; their reductions were launched on their own stream");
        return CNF_ERR_ARG;
    }
    return launch_status("cnf_bwd_defer_flush");
}

int64_t cnf_bwd_workspace_floats(int param_count) {
    // wide rows: 1024 partial rows + 1024 rows of the mixture fix-up launch + its 5 x 1024 flag words
    return (int64_t)((param_count <= kBwdMaxRowP ? kBwdMaxRows : 2048) + 1) * (param_count + 1) + 8192;
}

void cnf_set_actnorm_bwd_tiles(int on) { g_act_bwd_tiles.store(on ? 1 : 0, std::memory_order_relaxed); }
void cnf_set_affine_bwd_tiles(int mode) { g_aff_bwd_tiles.store(mode < 0 || mode > 2 ? 1 : mode, std::memory_order_relaxed); }

void cnf_set_bwd_tile(int chunks_in_flight, int groups_per_tile) {
    if (chunks_in_flight >= 0 && chunks_in_flight <= 3) g_bwd_u.store(chunks_in_flight, std::memory_order_relaxed);
    if (groups_per_tile >= 0 && groups_per_tile <= 64) g_bwd_g.store(groups_per_tile, std::memory_order_relaxed);
}

int cnf_affine_coupling_bwd(const float* z_out, const float* nn_out, const float* scaling_factor,
                            const float* mask, int mask_rows, int mask_cols,
                            const float* g_zout, const float* g_ldj,
                            float* g_z, float* g_nn, float* g_scaling_factor, float* workspace,
                            int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && nn_out && g_z && g_nn, "cnf_affine_coupling_bwd: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0 && D <= kBwdMaxD, "cnf_affine_coupling_bwd: bad shape");
    CNF_REQUIRE(!scaling_factor || (g_scaling_factor && workspace), "cnf_affine_coupling_bwd: scaling_factor needs g_scaling_factor and workspace");
    if (B == 0) return CNF_OK;
    if (!mask) { mask_rows = 1; mask_cols = D; }
    CNF_REQUIRE(mask_rows >= 1 && (mask_cols == D || mask_cols == 1),
                "cnf_affine_coupling_bwd: mask must be [rows,%d] or [rows,1], got [%d,%d]", D, mask_rows, mask_cols);
    if (mask_rows > N) mask_rows = N;
    const int L = N * D;
    // the forward's limits (cnf_affine_coupling): rows below 65536 elements, mask period x D + 3 table entries
    CNF_REQUIRE((long)N * D < 65536, "cnf_affine_coupling_bwd: N*D=%ld exceeds 65535", (long)N * D);
    if (mask_rows * D + 3 > kBwdMaxTab) {
        set_error("cnf_affine_coupling_bwd: mask period %d x D %d too large", mask_rows, D);
        return CNF_ERR_UNSUPPORTED;
    }
    hipStream_t st = (hipStream_t)stream;
    if (mask_rows == 1 && mask_cols == D && (D == 2 || D == 3 || D == 4 || D == 6 || D == 8) &&
        aligned_to(16, {z_out, nn_out, g_zout, g_z, g_nn}) &&
        (g_aff_bwd_tiles.load(std::memory_order_relaxed) == 2 ||
         (g_aff_bwd_tiles.load(std::memory_order_relaxed) == 1 && (!scaling_factor || !reverse)))) {
        // channel mask: token-owner wave tiles, constants (and scaling-factor sums) in registers.  Default where it is the faster
        // kernel at the benchmark shape (profiles/r04_bwd_kernel_stats.csv, rocprofv3): without a scaling factor 29.8 vs 30.8 us,
        // with one in the forward direction 32.0 vs 32.6 us; in the inverse direction the flat tiles win (31.6 vs 32.0 us: the
        // kernel needs 152 VGPRs there).  Mode 2 / 0 force either kernel for A/B runs
        const long ntok = (long)B * N;
        AffTileArgs t{z_out, nn_out, scaling_factor, mask, g_zout, g_ldj, g_z, g_nn, workspace, ntok, N, D, make_fastdiv((uint32_t)N),
                      ntok < (1l << 32) / N ? 1 : 0};
        const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
        const long tiles = std::max<long>(ntok / ((long)kWave * tp), 1);
        const int G = bwd_g(2);
        const dim3 grid((unsigned)std::min<long>(std::max<long>((tiles + (long)kWavesPerBlock * G - 1) / ((long)kWavesPerBlock * G), 1), kBwdMaxBlocks));
        const dim3 block(kBlock);
        const bool fast = math_mode() == 1, has_sf = scaling_factor != nullptr, rev = reverse != 0;
#define AFT3(DD, SF, RV) \
        do { \
            if (fast) CNF_LAUNCH((affine_bwd_tile_kernel<DD, SF, RV, true>), grid, block, 0, st, t); \
            else CNF_LAUNCH((affine_bwd_tile_kernel<DD, SF, RV, false>), grid, block, 0, st, t); \
        } while (0)
#define AFT(DD) \
        case DD: \
            if (has_sf) { if (rev) AFT3(DD, true, true); else AFT3(DD, true, false); } \
            else { if (rev) AFT3(DD, false, true); else AFT3(DD, false, false); } \
            break;
        switch (D) { AFT(2) AFT(3) AFT(4) AFT(6) AFT(8) }
#undef AFT
#undef AFT3
        if (scaling_factor) reduce_partials(workspace, (int)grid.x * kWavesPerBlock, D, g_scaling_factor, st);
        return launch_status("cnf_affine_coupling_bwd");
    }
    AffBwdArgs a{z_out, nn_out, scaling_factor, mask, g_zout, g_ldj, g_z, g_nn, workspace,
                 N, D, L, mask_rows, mask_cols, mask_rows * D, make_fastdiv((uint32_t)(mask_rows * D)), make_fastdiv((uint32_t)D)};
    const int vec = vec_for(L, {z_out, nn_out, g_zout, g_z, g_nn});
    const int U = vec == 4 ? bwd_u(2) : 2;
    const FlatTiling tl = make_flat_tiling(B, L, U, vec, bwd_g(1));
    dispatch_vec_u(tl.vec, U, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
        launch_affine_bwd<V, UU>(a, tl, scaling_factor != nullptr, reverse != 0, st);
    });
    if (scaling_factor) reduce_partials(workspace, (int)flat_grid(tl).x * kWavesPerBlock, D, g_scaling_factor, st);
    return launch_status("cnf_affine_coupling_bwd");
}

int cnf_affine_params_bwd(const float* nn_out, const float* scaling_factor, const float* mask, int mask_rows, int mask_cols,
                          const float* g_s, const float* g_t, float* g_nn, float* g_scaling_factor, float* workspace,
                          int B, int N, int D, cnf_stream_t stream) {
    CNF_REQUIRE(nn_out && g_nn, "cnf_affine_params_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && D <= kBwdMaxD, "cnf_affine_params_bwd: bad shape");
    CNF_REQUIRE(!scaling_factor || (g_scaling_factor && workspace), "cnf_affine_params_bwd: scaling_factor needs g_scaling_factor and workspace");
    if (!mask) { mask_rows = 1; mask_cols = D; }
    if (mask_rows > N) mask_rows = N;
    CNF_REQUIRE((long)N * D < 65536, "cnf_affine_params_bwd: N*D=%ld exceeds 65535", (long)N * D);
    if (mask_rows * D + 3 > kBwdMaxTab) {
        set_error("cnf_affine_params_bwd: mask period %d x D %d too large", mask_rows, D);
        return CNF_ERR_UNSUPPORTED;
    }
    const int L = N * D;
    AffParamsBwdArgs a{nn_out, scaling_factor, mask, g_s, g_t, g_nn, workspace, N, D, L, mask_rows, mask_cols,
                       mask_rows * D, make_fastdiv((uint32_t)(mask_rows * D)), make_fastdiv((uint32_t)D)};
    const int vec = vec_for(L, {nn_out, g_s, g_t, g_nn});
    const FlatTiling tl = make_flat_tiling(B, L, 2, vec, bwd_g(1));
    const dim3 grid = flat_grid(tl), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const size_t lds = scaling_factor ? (size_t)D * kBlock * sizeof(float) : 0;
    dispatch_vec_u(tl.vec, 2, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
        if (scaling_factor) CNF_LAUNCH((affine_params_bwd_kernel<V, 2, true>), grid, block, lds, st, a, tl);
        else CNF_LAUNCH((affine_params_bwd_kernel<V, 2, false>), grid, block, lds, st, a, tl);
    });
    if (scaling_factor) reduce_partials(workspace, (int)grid.x * kWavesPerBlock, D, g_scaling_factor, st);
    return launch_status("cnf_affine_params_bwd");
}

int cnf_affine_transform_bwd(const float* z_out, const float* s, const float* t, const float* g_zout, const float* g_ldj,
                             float* g_z, float* g_s, float* g_t, int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && s && t && g_z && g_s && g_t, "cnf_affine_transform_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_affine_transform_bwd: bad shape");
    const int L = N * D;
    const int vec = vec_for(L, {z_out, s, t, g_zout, g_z, g_s, g_t});
    CNF_REQUIRE(flat_ok(L, vec), "cnf_affine_transform_bwd: rows of %d elements are too long", L);
    AffTransformBwdArgs a{z_out, s, t, g_zout, g_ldj, g_z, g_s, g_t, L, reverse};
    const FlatTiling tl = make_flat_tiling(B, L, 2, vec, bwd_g(1));
    const dim3 grid = flat_grid(tl), block(kBlock);
    dispatch_vec_u(tl.vec, 2, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
        CNF_LAUNCH((affine_transform_bwd_kernel<V, 2>), grid, block, 0, (hipStream_t)stream, a, tl);
    });
    return launch_status("cnf_affine_transform_bwd");
}

int cnf_ext_actnorm_bwd(const float* z_out, const float* nn_out, const float* pad,
                        const float* g_zout, const float* g_ldj, float* g_z, float* g_nn,
                        int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && nn_out && g_z && g_nn, "cnf_ext_actnorm_bwd: null tensor");
    CNF_REQUIRE(B >= 0 && N > 0 && D > 0, "cnf_ext_actnorm_bwd: bad shape");
    if (B == 0) return CNF_OK;
    const long ntok = (long)B * N;
    ExtBwdArgs a{z_out, nn_out, pad, g_zout, g_ldj, g_z, g_nn, ntok, N, D, reverse, make_fastdiv((uint32_t)N),
                 ntok < (1l << 32) / N ? 1 : 0};
    hipStream_t st = (hipStream_t)stream;
    const bool fast = math_mode() == 1;
    const bool aligned = aligned_to(16, {z_out, nn_out, g_zout, g_z, g_nn});
    const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    const dim3 block(kBlock);
    // wave tiles (coalesced spans, LDS transpose) where there is an upstream gradient and at least one tile per wave of a modest
    // grid; `G` tiles per wave as for the 1x1 convolution
    const long tiles = ntok / ((long)kWave * tp);
    const bool tiled = g_zout != nullptr && tiles >= 1;
    const int G = bwd_g(2);
    const dim3 grid = tiled ? dim3((unsigned)std::min<long>(std::max<long>((tiles + (long)kWavesPerBlock * G - 1) / ((long)kWavesPerBlock * G), 1), kBwdMaxBlocks))
                            : dim3(stream_grid((ntok + tp - 1) / tp));
#define EXT_BWD(DD)                                                                                                 \
    do {                                                                                                            \
        if (tiled) {                                                                                                \
            if (fast) CNF_LAUNCH((ext_actnorm_bwd_tile_kernel<DD, true>), grid, block, 0, st, a);                   \
            else CNF_LAUNCH((ext_actnorm_bwd_tile_kernel<DD, false>), grid, block, 0, st, a);                       \
        } else if (fast) CNF_LAUNCH((ext_actnorm_bwd_group_kernel<DD, true>), grid, block, 0, st, a);               \
        else CNF_LAUNCH((ext_actnorm_bwd_group_kernel<DD, false>), grid, block, 0, st, a);                          \
    } while (0)
    switch (aligned ? D : 0) {
        case 1: EXT_BWD(1); break;
        case 2: EXT_BWD(2); break;
        case 3: EXT_BWD(3); break;
        case 4: EXT_BWD(4); break;
        case 5: EXT_BWD(5); break;
        case 6: EXT_BWD(6); break;
        case 8: EXT_BWD(8); break;
        default:
            if (fast) CNF_LAUNCH((ext_actnorm_bwd_elem_kernel<true>), dim3(stream_grid(ntok * D)), block, 0, st, a);
            else CNF_LAUNCH((ext_actnorm_bwd_elem_kernel<false>), dim3(stream_grid(ntok * D)), block, 0, st, a);
    }
#undef EXT_BWD
    return launch_status("cnf_ext_actnorm_bwd");
}

int cnf_actnorm_bwd(const float* z_out, const float* bias, const float* scales,
                    const float* pad, const float* length, const float* g_zout, const float* g_ldj,
                    float* g_z, float* g_bias, float* g_scales, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(z_out && bias && scales && g_z && g_bias && g_scales && workspace, "cnf_actnorm_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0 && D <= kBwdMaxD, "cnf_actnorm_bwd: bad shape");
    const int L = N * D;
    const int vec = vec_for(L, {z_out, g_zout, g_z});
    CNF_REQUIRE(flat_ok(L, vec), "cnf_actnorm_bwd: rows of %d elements are too long", L);
    ActBwdArgs a{z_out, bias, scales, pad, length, g_zout, g_ldj, g_z, workspace, B, N, D, L, reverse, make_fastdiv((uint32_t)D), (long)B * N};
    const int U = vec == 4 ? bwd_u(2) : 2;
    hipStream_t st = (hipStream_t)stream;
    const bool fast = math_mode() == 1;
    const dim3 block(kBlock);
    if ((D <= 6 || D == 8) && g_zout && aligned_to(16, {z_out, g_zout, g_z}) && g_act_bwd_tiles.load(std::memory_order_relaxed)) {
        // token-owner wave tiles with register sums
        const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
        const long tiles = std::max<long>(a.ntok / ((long)kWave * tp), 1);
        const int G = bwd_g(4);
        const dim3 tgrid((unsigned)std::min<long>(std::max<long>((tiles + (long)kWavesPerBlock * G - 1) / ((long)kWavesPerBlock * G), 1), kBwdMaxBlocks));
#define ACT_T(DD) \
    case DD: \
        if (fast) CNF_LAUNCH((actnorm_bwd_tile_kernel<DD, true>), tgrid, block, 0, st, a); \
        else CNF_LAUNCH((actnorm_bwd_tile_kernel<DD, false>), tgrid, block, 0, st, a); \
        break;
        switch (D) { ACT_T(1) ACT_T(2) ACT_T(3) ACT_T(4) ACT_T(5) ACT_T(6) ACT_T(8) }
#undef ACT_T
        enqueue_reduce(2 * D, workspace, (int)tgrid.x * kWavesPerBlock, 2 * D + 1, g_bias, g_scales, D, g_ldj ? 2 * D : -1, D, st);
        return launch_status("cnf_actnorm_bwd");
    }
    // two groups per wave: the wave's closing reduction of its 2D + 1 sums weighs as much as one group's arithmetic
    const FlatTiling tl = make_flat_tiling(B, L, U, vec, bwd_g(2));
    const dim3 grid = flat_grid(tl);
    const size_t lds = (size_t)2 * D * kBlock * sizeof(float);
    dispatch_vec_u(tl.vec, U, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
#define ACT_BWD(FA, PD) CNF_LAUNCH((actnorm_bwd_kernel<V, UU, FA, PD>), grid, block, lds, st, a, tl)
        if (fast) {
            if (pad) ACT_BWD(true, true);
            else ACT_BWD(true, false);
        } else {
            if (pad) ACT_BWD(false, true);
            else ACT_BWD(false, false);
        }
#undef ACT_BWD
    });
    // partial rows are [d bias (D) | d scales (D) | log-det term]: summed straight into the two gradient tensors
    enqueue_reduce(2 * D, workspace, (int)grid.x * kWavesPerBlock, 2 * D + 1, g_bias, g_scales, D, g_ldj ? 2 * D : -1, D, st);
    return launch_status("cnf_actnorm_bwd");
}

int cnf_invconv_bwd(const float* x, const float* weight, const float* pad, const float* length,
                    const float* g_zout, const float* g_ldj,
                    float* g_x, float* g_weight, float* g_sldj, float* workspace,
                    int B, int N, int D, int reverse, cnf_stream_t stream) {
    CNF_REQUIRE(x && weight && g_x && g_weight && g_sldj && workspace, "cnf_invconv_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_invconv_bwd: bad shape");
    if (D > 12) {
        set_error("cnf_invconv_bwd: built for D <= 12 (got %d): a larger weight gradient is a GEMM (x^T g)", D);
        return CNF_ERR_UNSUPPORTED;
    }
    ConvBwdArgs a{x, weight, pad, length, g_zout, g_ldj, g_x, workspace, (long)B * N, B, N, D, reverse};
    const int P = D * D + 1;
    hipStream_t st = (hipStream_t)stream;
    const dim3 block(kBlock);
    // one wave tile = 64 token groups; every wave takes `G` tiles so that its register sums are combined once
    const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    const long tiles = std::max<long>(a.ntok / ((long)kWave * tp), 1);
    const int G = bwd_g(4);
    dim3 grid((unsigned)std::min<long>(std::max<long>((tiles + (long)kWavesPerBlock * G - 1) / ((long)kWavesPerBlock * G), 1), kBwdMaxBlocks));
    switch (D) {
        case 1: CNF_LAUNCH((invconv_bwd_kernel<1>), grid, block, 0, st, a); break;
        case 2: CNF_LAUNCH((invconv_bwd_kernel<2>), grid, block, 0, st, a); break;
        case 3: CNF_LAUNCH((invconv_bwd_kernel<3>), grid, block, 0, st, a); break;
        case 4: CNF_LAUNCH((invconv_bwd_kernel<4>), grid, block, 0, st, a); break;
        case 5: CNF_LAUNCH((invconv_bwd_kernel<5>), grid, block, 0, st, a); break;
        case 6: CNF_LAUNCH((invconv_bwd_kernel<6>), grid, block, 0, st, a); break;
        case 8: CNF_LAUNCH((invconv_bwd_kernel<8>), grid, block, 0, st, a); break;
        default:
            grid = dim3((unsigned)std::min<long>(std::max<long>((a.ntok * D + kBlock - 1) / kBlock, 1), kBwdMaxBlocks));
            CNF_LAUNCH(invconv_bwd_generic_kernel, grid, block, (size_t)P * kBlock * sizeof(float), st, a);
    }
    enqueue_reduce(P, workspace, (int)grid.x * kWavesPerBlock, P, g_weight, g_sldj, D * D, -1, 0, st);
    return launch_status("cnf_invconv_bwd");
}

int cnf_actnorm_invconv_bwd(const float* saved, int saved_is_output, const float* bias, const float* scales, const float* weight,
                            const float* weight_inv, const float* pad, const float* length,
                            const float* g_zout, const float* g_ldj, float* g_z, float* g_params, float* workspace,
                            int B, int N, int D, cnf_stream_t stream) {
    CNF_REQUIRE(saved && bias && scales && weight && g_z && g_params && workspace, "cnf_actnorm_invconv_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_actnorm_invconv_bwd: bad shape");
    if (!(D <= 6 || D == 8)) {
        set_error("cnf_actnorm_invconv_bwd: built for D in {1..6, 8} (got %d): run the two layers' backward kernels", D);
        return CNF_ERR_UNSUPPORTED;
    }
    const int P = D * D + 2 * D + 2;
    hipStream_t st = (hipStream_t)stream;
    if (saved_is_output && !weight_inv) {
        // behind the partial rows (cnf_bwd_workspace_floats(P) holds kBwdMaxRows + 1 rows of P + 1)
        float* inv = workspace + (size_t)kBwdMaxRows * P;
        CNF_LAUNCH(small_inverse_kernel, dim3(1), dim3(kWave), 0, st, weight, inv, D);
        weight_inv = inv;
    }
    ActConvBwdArgs a{saved, bias, scales, weight, weight_inv, pad, length, g_zout, g_ldj, g_z, workspace, (long)B * N, B, N, D};
    const dim3 block(kBlock);
    const int tp = (D % 4 == 0) ? 1 : (D % 2 == 0 ? 2 : 4);
    const long tiles = std::max<long>(a.ntok / ((long)kWave * tp), 1);
    const int G = bwd_g(4);
    const dim3 grid((unsigned)std::min<long>(std::max<long>((tiles + (long)kWavesPerBlock * G - 1) / ((long)kWavesPerBlock * G), 1), kBwdMaxBlocks));
#define ACB(DD) \
    case DD: \
        if (saved_is_output) CNF_LAUNCH((actconv_bwd_kernel<DD, true>), grid, block, 0, st, a); \
        else CNF_LAUNCH((actconv_bwd_kernel<DD, false>), grid, block, 0, st, a); \
        break;
    switch (D) {
        ACB(1) ACB(2) ACB(3) ACB(4) ACB(5) ACB(6) ACB(8)
    }
#undef ACB
    // rows [dW | d sldj | d bias | d scales | ActNorm's log-det term] -> g_params [dW | d sldj | d bias | d scales]
    enqueue_reduce(P - 1, workspace, (int)grid.x * kWavesPerBlock, P, g_params, (float*)nullptr, P - 1, g_ldj ? P - 1 : -1, D * D + 1 + D, st);
    return launch_status("cnf_actnorm_invconv_bwd");
}

int cnf_logistic_log_prob_bwd(const float* x, const float* g_logp, float* g_x, int64_t n, float mu, float sigma,
                              cnf_stream_t stream) {
    CNF_REQUIRE(x && g_logp && g_x && n >= 0, "cnf_logistic_log_prob_bwd: bad argument");
    if (n == 0) return CNF_OK;
    const dim3 grid(stream_grid((n + 3) / 4)), block(kBlock);
    if (math_mode() == 1)
        CNF_LAUNCH((logistic_log_prob_bwd_kernel<true>), grid, block, 0, (hipStream_t)stream, x, g_logp, g_x, (long)n, mu, sigma);
    else
        CNF_LAUNCH((logistic_log_prob_bwd_kernel<false>), grid, block, 0, (hipStream_t)stream, x, g_logp, g_x, (long)n, mu, sigma);
    return launch_status("cnf_logistic_log_prob_bwd");
}

int cnf_prior_nll_bwd(const float* z, const float* pad, const float* length, const float* g_nll,
                      float* g_z, float* g_ldj, int B, int N, int D, float sigma, cnf_stream_t stream) {
    CNF_REQUIRE(z && g_nll && g_z, "cnf_prior_nll_bwd: null tensor");
    CNF_REQUIRE(B > 0 && N > 0 && D > 0, "cnf_prior_nll_bwd: bad shape");
    const int L = N * D;
    const int vec = vec_for(L, {z, g_z});
    CNF_REQUIRE(flat_ok(L, vec), "cnf_prior_nll_bwd: rows of %d elements are too long", L);
    NllBwdArgs a{z, pad, length, g_nll, g_z, g_ldj, B, N, D, L, (float)(1.0 / (double)sigma), make_fastdiv((uint32_t)D)};
    const int U = vec == 4 ? bwd_u(2) : 2;
    const FlatTiling tl = make_flat_tiling(B, L, U, vec, bwd_g(1));
    const dim3 grid = flat_grid(tl), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const bool fast = math_mode() == 1;
    dispatch_vec_u(tl.vec, U, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
        if (fast) CNF_LAUNCH((prior_nll_bwd_kernel<V, UU, true>), grid, block, 0, st, a, tl);
        else CNF_LAUNCH((prior_nll_bwd_kernel<V, UU, false>), grid, block, 0, st, a, tl);
    });
    return launch_status("cnf_prior_nll_bwd");
}

int cnf_sigmoid_flow_bwd(const float* z_in, const float* g_zout, const float* g_ldj, float* g_z,
                         int B, int L, int reverse, float alpha, cnf_stream_t stream) {
    CNF_REQUIRE(z_in && g_z, "cnf_sigmoid_flow_bwd: null tensor");
    CNF_REQUIRE(B > 0 && L > 0, "cnf_sigmoid_flow_bwd: bad shape");
    const int vec = vec_for(L, {z_in, g_zout, g_z});
    CNF_REQUIRE(flat_ok(L, vec), "cnf_sigmoid_flow_bwd: rows of %d elements are too long", L);
    SigBwdArgs a{z_in, g_zout, g_ldj, g_z, L, reverse, alpha};
    const int U = vec == 4 ? bwd_u(2) : 2;
    const FlatTiling tl = make_flat_tiling(B, L, U, vec, bwd_g(1));
    const dim3 grid = flat_grid(tl), block(kBlock);
    hipStream_t st = (hipStream_t)stream;
    const bool fast = math_mode() == 1;
    dispatch_vec_u(tl.vec, U, [&](auto v_, auto u_) {
        BWD_VU(v_, u_);
        if (fast) CNF_LAUNCH((sigmoid_flow_bwd_kernel<V, UU, true>), grid, block, 0, st, a, tl);
        else CNF_LAUNCH((sigmoid_flow_bwd_kernel<V, UU, false>), grid, block, 0, st, a, tl);
    });
    return launch_status("cnf_sigmoid_flow_bwd");
}

}  // extern "C"
