/*
 * cnf_tuning.h — tuning, A/B, profiling and experimental entry points of libcnf_hip.so.
 *
 * NOT part of what a maintainer of the reference binds (that is include/cnf_hip.h: one entry point per eager op chain of the
 * reference plus the two semantic switches cnf_set_math_mode / cnf_set_inverse_mode).  Everything here exists for this
 * repository's own measurements: kernel-choice knobs whose two sides are compared in tests / tools, the dispatch-bound kernel
 * timer and the stream probes bench.py uses for its roofline, and the deferred-reduction API, which no product caller uses
 * (see its comment).  The reference has no counterpart for any of them.  All knobs are process-wide atomics: set them before
 * use; they are not per device or per thread.  Knobs whose losing side is on file were removed in round 6
 * (cnf_set_linear_tiles, cnf_set_mixture_bwd_prefetch).
 */
#ifndef CNF_TUNING_H
#define CNF_TUNING_H

#include "cnf_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Tuning knob for the row-streaming kernels: float4 chunks one wave owns per tile (default 128). */
void cnf_set_tile_chunks(int chunks);

/* Load scheduling of the affine coupling kernel: 0 = one chunk at a time, software-pipelined (the next
 * chunk's loads are issued before the current one is computed); 1..4 = that many chunks per lane
 * loaded back to back.  Default 2. */
void cnf_set_unroll(int u);

/* Items (transformed elements) one wave of the fp32 mixture forward kernel owns, 64..512 (default 128). */
void cnf_set_mixture_tile(int items);

/* Flat tiles of the streaming backward kernels (csrc/cnf_backward.hip): 16-byte chunks a lane keeps in flight (1..3) and
 * chunk groups one wave walks (1..64); 0 = every kernel's own default (2 chunks; 1 group for the flat-tile kernels, 2 for ExtActNorm, 4 for the token-owner ActNorm / 1x1 conv / fused-pair kernels).  A tuning knob like the ones above: the reference has no
 * counterpart (its backward is autograd, general/train.py:144-155).  All tuning knobs are process-wide atomics — set them
 * before use; they are not per device or per thread. */
void cnf_set_bwd_tile(int chunks_in_flight, int groups_per_tile);

/* cnf_actnorm_bwd: 1 (default) = the token-owner wave-tile kernel (register sums) for D in {1..6, 8}, 0 = always the flat-tile
 * kernel with lane-private LDS sums (A/B measurements and tests; same results up to the order of the additions). */
void cnf_set_actnorm_bwd_tiles(int on);

/* cnf_affine_coupling_bwd, channel masks at D in {2, 3, 4, 6, 8}: 1 (default) = the token-owner wave-tile kernel where it is the
 * faster one (no scaling factor, or the forward direction), 2 = always, 0 = always the flat-tile kernel (A/B measurements and
 * tests; same results up to the order of the additions). */
void cnf_set_affine_bwd_tiles(int mode);

/* fp32 mixture-coupling backward (cnf_mixture_coupling_bwd_f32), which streaming kernel: -1 (default) = the rolled run-time-K
 * kernel with 1 / 2 / 4 lanes per item by the amount of work; 2 / 3 / 4 force that kernel with 1 / 2 / 4 lanes (5-7: its build
 * held to 4 waves per SIMD); 0 / 1 = the unrolled register-slot kernels of K = 4 / 8 / 16 (natural registers / held to 4 waves
 * per SIMD), the defaults until round 4.  A/B knob; same gradients up to the order of the additions.  No reference counterpart. */
void cnf_set_mixture_bwd_waves(int mode);

/* fp32 mixture-coupling backward: megabytes of parameter-gradient rows above which a one-lane-per-item launch streams — nontemporal DMA
 * loads, nontemporal write-back, the reference layout's tokens written in address order with their zero blocks (default 128, the size
 * past which the memory-side cache no longer absorbs the rows; < 0 restores it; cnf_set_mixture_nt_mb(0) switches streaming off
 * altogether).  Same gradients bit for bit either way; the tests set 0 to run the streaming write-back on small shapes. */
void cnf_set_mixture_bwd_big_mb(int megabytes);

/* Kernel timing bound to the dispatch (bench.py's `roofline`; a timed launch costs ~4 us of queue time; the reference has no counterpart — its
 * only clock is the host-side time_per_step tracker, general/train.py:147-157).  cnf_prof_arm(n): the next n
 * kernel launches made by this host thread through this library carry their dispatch's own start / stop
 * timestamps (hipExtLaunchKernelGGL event pair; at most 8192 pairs between two collects).
 * cnf_prof_collect: waits for the timed launches, writes their durations in milliseconds in launch order
 * (HOST pointer), returns how many were written, and disarms. */
int cnf_prof_arm(int launches);

int cnf_prof_collect(float* ms_out_host, int capacity);

/* Diagnostic: a streaming kernel with the affine coupling's traffic mix and no real arithmetic — a [n] fp32 read,
 * b [2n] fp32 read, out [n] fp32 written (out = a + b_even * b_odd), 16-byte accesses, `chunks_per_lane` in {1,2,4}.
 * bench.py times it in the same run as the coupling kernel: the measured ceiling for 12 B read + 4 B written per
 * element on this device ("a measured stream-copy ceiling from the same run").  No reference
 * counterpart. */
int cnf_stream_probe(const float* a, const float* b, float* out, long n, int chunks_per_lane, cnf_stream_t stream);

/* The same for the affine coupling's BACKWARD mix (tools/bwd_probe.py, bench.py's extra.kernels): a [n], b [2n], c [n]
 * read, o1 [n], o2 [2n] written — 16 B read + 12 B written per element; chunks_per_lane in {1,2}; hint bit 0 / 1 / 2 =
 * nontemporal loads of (a, b) / of c / nontemporal stores.  No reference counterpart. */
int cnf_stream_probe_bwd(const float* a, const float* b, const float* c, float* o1, float* o2, long n,
                         int chunks_per_lane, int hint, cnf_stream_t stream);

/* Experiment kept for the record (tools/affine_fwd_tile_probe.py, profiles/r04_affine_fwd_tile_experiment.txt): the affine
 * coupling FORWARD (coupling_layer.py:53-63) in the token-owner wave-tile form that won in the backward kernels, D = 6,
 * channel mask, scaling factor, rows of 16..128 tokens, B*N a multiple of 128.  Same z bits as cnf_affine_coupling; SLOWER
 * (17.6-18.7 vs 16.7 us at the benchmark shape), so the flat row-tile kernel stays.  No reference counterpart. */
int cnf_probe_affine_fwd_tile(const float* z, const float* nn_out, const float* scaling_factor, const float* mask, const float* ldj_in,
                              float* z_out, float* ldj_out, int B, int N, int tiles_per_wave, int nontemporal_nn_loads,
                              cnf_stream_t stream);

/* Test / timing hook of the fp64 log / log1p / reciprocal the reference-precision mixture kernels use
 * (csrc/cnf_f64_math.h): out[i] = f(in[i]), which = 0 log of a positive normal, 1 reciprocal, 2 log1p on [0, 1]; 3 / 4 / 5 / 6 the
 * library's exp / log / division / log1p; reps > 1 applies f reps times per element (tools/f64_math_rates.py).  fp64 device
 * pointers.  No reference counterpart (numpy's / mpmath's log are what tests/test_gpu_f64_math.py compares with). */
int cnf_probe_f64_math(int which, const double* in, double* out, long n, int reps, cnf_stream_t stream);

/* Tuning / A-B knobs of the fp32 mixture kernels: which kernel serves math mode 1 (0 = token-pass kernel on
 * DMA-staged rows, default; 1 = the round-1 kernel); lanes per item for a run-time K (0 = automatic: K = 4 / 8 / 16
 * exactly, every other K <= 64 on predicated register slots, larger K on the rolled LDS loop; 1, 2, 4 = the rolled
 * loop with that many lanes per item, the A/B partner of the register slots); the number of waves a split-row launch
 * aims at (default 4096; never more workgroups than the device holds at once); and whether forward / inverse may stage whole tokens when skipping the untransformed
 * parameter blocks would skip no 128-byte lines anyway (default 1).  No counterpart in the reference (pure tuning). */
void cnf_set_mixture_kernel(int which);

void cnf_set_mixture_lanes(int lanes_per_item);

void cnf_set_mixture_split(int waves);

void cnf_set_mixture_whole_tokens(int on);

/* Staged parameter bytes of one forward / inverse launch (MB) above which its DMA loads carry the nontemporal hint: the rows are
 * read once.  Measured (profiles/r05_mixture_nt_sweep.txt): 2-8 % from 78 MB up (S*: 312 MB, fp32 forward 103.6 -> 95-100 us), -4 %
 * at configs[1]'s 52 MB.  Default 64; 0 = never; negative = default. */
void cnf_set_mixture_nt_mb(int megabytes);

/* A-B knob of the LDS-resident encoder kernels: 2 = two tokens per lane with 16-byte LDS constants wherever the shape
 * allows it (whole-row wave tiles of an even number of tokens, 16-byte aligned views, D in {1,2,3,4,6,8}), 1 = the
 * one-token-per-lane kernels (the fallback for every other shape), both on 256-token wave tiles; 0 (default) = by
 * measurement: since the forward sums class densities instead of streaming a log-sum-exp that is the one-token kernels
 * at every size, on 64- / 128-token tiles.  Same arithmetic per token: 1 and 2 give bit-identical latents, log-det and
 * decoded indices (linear_encoding.py:59-133,153-196); 0 differs from them only in the order of the per-row sums. */
void cnf_set_encoder_kernel(int which);

/* number of cnf_encoder_forward / cnf_encoder_decode calls this process served with the two-token kernels (tests) */
int64_t cnf_encoder_pair_launches(void);

/* A-B knob of the two entry points above: 0 (default) = by shape as described, 1 = always the two passes, 2 / 3 = the pair
 * kernel on its 256-lane / 512-lane workgroup wherever the pair lanes (192 / 448) hold the classes (with the pre-pass when no
 * class_prob_log is given). */
void cnf_set_encoder_bwd_kernel(int which);

/* EXPERIMENTAL — deferred reductions, for a host that owns a whole backward pass (general/train.py:144-155 `loss.backward()` as one
 * unit).  No product caller: PyTorch's autograd consumes a node's parameter gradients before the next node runs (the LU weight
 * assembly reads d loss / d W at once) and no autograd.Function of this repository issues two deferrable reductions, so the Python
 * host cannot use it; bench.py measures it and one test pins its bits.  Kept out of include/cnf_hip.h for that reason.
 * Between cnf_bwd_defer_begin() and cnf_bwd_defer_flush(stream) — host-thread local — cnf_affine_coupling_bwd,
 * cnf_affine_params_bwd, cnf_actnorm_bwd, cnf_invconv_bwd and cnf_actnorm_invconv_bwd only write their partial rows and queue
 * their closing reduction; the flush runs all of them as ONE launch on the stream of the calls (a call on another stream, or a
 * 25th call, flushes what is queued first; a begin flushes what an aborted batch left behind; a flush given another stream than the
 * calls' returns CNF_ERR_ARG after launching on the right one).  Every call in between needs its OWN workspace, alive until the
 * flush, and its parameter-gradient outputs hold nothing before it.  The gradients are the bits of the immediate reductions. */
void cnf_bwd_defer_begin(void);

int cnf_bwd_defer_flush(cnf_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CNF_TUNING_H */
