// How fast does a wave-pass reader stream HBM on gfx950, by load path and pass shape?  (round 6: the fp32 mixture forward with
// its arithmetic compiled out moves its 377 MB at 5.1 TB/s, the affine coupling kernel its 100 MB at 5.6, a plain float4 read of
// 512 MiB runs at 6.5 — which part of the gap belongs to the DMA path, the pass size, the serial wait per pass, the start-up?)
//
// A wave owns a contiguous chunk of PASSES x INSTR KiB and walks it pass by pass: INSTR loads of 16 bytes per lane (1 KiB per
// instruction), wait, touch the data, next pass.  MODE 0: global_load_dwordx4 into VGPRs; MODE 1: global_load_lds_dwordx4 into
// the wave's LDS stage (the token-pass kernels' path); MODE 2: the DMA of pass p + 1 issued before pass p is touched (two stages).
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/dma_stream tools/microbench/dma_stream.hip && /tmp/dma_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int MODE, int INSTR, int NT>
__global__ __launch_bounds__(256) void reader(const char* buf, long bytes, int passes, float* out, int pad_lds, char* dst = nullptr) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const long chunk = (long)passes * INSTR * 1024;
    const long base = ((long)blockIdx.x * 4 + wave) * chunk;
    if (base >= bytes) return;
    float acc = 0.f;
    char* stage = smem + (size_t)wave * (MODE == 2 ? 2 : 1) * INSTR * 1024;
    auto dma = [&](char* dst, long off) {
#pragma unroll
        for (int i = 0; i < INSTR; ++i)
            __builtin_amdgcn_global_load_lds((glb_void_t*)(buf + off + i * 1024 + lane * 16), (lds_void_t*)(dst + i * 1024), 16, 0, NT ? 2 : 0);
    };
    if (MODE == 2) dma(stage, base);
    for (int p = 0; p < passes; ++p) {
        const long off = base + (long)p * INSTR * 1024;
        if (MODE == 0) {
            f4 v[INSTR];
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                const f4* src = reinterpret_cast<const f4*>(buf + off + i * 1024 + lane * 16);
                v[i] = NT ? __builtin_nontemporal_load(src) : *src;
            }
#pragma unroll
            for (int i = 0; i < INSTR; ++i) acc += v[i].x + v[i].w;
        } else if (MODE == 3) {
            // write-only: INSTR float4 stores per lane and pass
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                f4* d = reinterpret_cast<f4*>(dst + off + i * 1024 + lane * 16);
                const f4 v = {acc, 1.f, 2.f, (float)p};
                if (NT) __builtin_nontemporal_store(v, d); else *d = v;
            }
        } else if (MODE == 4) {
            // copy: DMA into the stage, wait, LDS -> registers -> float4 stores (the backward's write-back), next pass
            dma(stage, off);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_wave_barrier();
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                const f4 v = *reinterpret_cast<const f4*>(stage + i * 1024 + lane * 16);
                f4* d = reinterpret_cast<f4*>(dst + off + i * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v, d); else *d = v;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        } else if (MODE == 5) {
            // copy with the stores issued BEHIND the next pass's DMA: the wait for that DMA (in-order counter) then leaves this
            // pass's INSTR stores outstanding instead of draining them — does a wave that both loads and stores lose time waiting for
            // its stores' acknowledgements?
            if (p == 0) dma(stage, off);
            constexpr int kLeave = INSTR;                                  // stores of the previous pass still in flight
            if (p == 0) __builtin_amdgcn_s_waitcnt(0x0F70);
            else __builtin_amdgcn_s_waitcnt((kLeave & 15) | ((kLeave >> 4) << 14) | (7 << 4) | (15 << 8));
            __builtin_amdgcn_wave_barrier();
            f4 v[INSTR];
#pragma unroll
            for (int i = 0; i < INSTR; ++i) v[i] = *reinterpret_cast<const f4*>(stage + i * 1024 + lane * 16);
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
            if (p + 1 < passes) dma(stage, off + INSTR * 1024);
#pragma unroll
            for (int i = 0; i < INSTR; ++i) {
                f4* d = reinterpret_cast<f4*>(dst + off + i * 1024 + lane * 16);
                if (NT) __builtin_nontemporal_store(v[i], d); else *d = v[i];
            }
        } else if (MODE == 1) {
            dma(stage, off);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_wave_barrier();
            acc += reinterpret_cast<const float*>(stage)[lane * 13];
            __builtin_amdgcn_s_waitcnt(0xC07F);          // lgkmcnt(0): the stage is read before it is refilled
            __builtin_amdgcn_wave_barrier();
        } else {
            char* cur = stage + (p & 1) * INSTR * 1024;
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_wave_barrier();
            if (p + 1 < passes) dma(stage + ((p + 1) & 1) * INSTR * 1024, off + INSTR * 1024);
            acc += reinterpret_cast<const float*>(cur)[lane * 13];
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (acc == 12345.678f) out[0] = acc;
}

template <int MODE, int INSTR, int NT>
static void run(const char* bufs[3], long bytes, int passes, float* out, int extra_lds, const char* what, char* dsts[3] = nullptr) {
    const long chunk = (long)passes * INSTR * 1024;
    const unsigned grid = (unsigned)((bytes / chunk + 3) / 4);
    const size_t lds = ((MODE == 0 || MODE == 3) ? 0 : (size_t)4 * (MODE == 2 ? 2 : 1) * INSTR * 1024) + extra_lds;      // (modes 4, 5: one stage)
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((reader<MODE, INSTR, NT>), dim3(grid), dim3(256), lds, 0, bufs[r], bytes, passes, out, 0, dsts ? dsts[r] : nullptr);
    CK(hipDeviceSynchronize());
    const int reps = 9;
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((reader<MODE, INSTR, NT>), dim3(grid), dim3(256), lds, 0, bufs[r % 3], bytes, passes, out, 0, dsts ? dsts[r % 3] : nullptr);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    ms /= reps;
    printf("%-28s %d KiB x %3d passes per wave, %6u workgroups, LDS %5zu B: %7.1f us  %5.2f TB/s\n", what, INSTR, passes, grid, lds, ms * 1e3,
           bytes / (ms * 1e-3) / 1e12);
}

int main() {
    const long bytes = 327155712l;          // S* compact: 16384 x 64 tokens x 312 bytes
    const char* bufs[3];
    float* out;
    for (int r = 0; r < 3; ++r) { char* p; CK(hipMalloc(&p, bytes + 65536)); CK(hipMemset(p, 0, bytes + 65536)); bufs[r] = p; }
    CK(hipMalloc(&out, 64));
    if (getenv("DMA_PROFILE")) {
        // the three reference streams only (counter runs: tools/pmc_stream_ref.sh)
        char* d3[3];
        for (int r = 0; r < 3; ++r) { CK(hipMalloc(&d3[r], bytes + 65536)); CK(hipMemset(d3[r], 0, bytes + 65536)); }
        run<1, 7, 1>(bufs, bytes, 16, out, 0, "LDS DMA, nt");
        run<3, 7, 0>(bufs, bytes, 16, out, 0, "write only", d3);
        run<4, 7, 1>(bufs, bytes, 16, out, 0, "copy, nt loads and stores", d3);
        return 0;
    }
    for (int passes : {1, 3, 4, 16}) {
        run<0, 7, 0>(bufs, bytes, passes, out, 0, "VGPR loads");
        run<0, 7, 1>(bufs, bytes, passes, out, 0, "VGPR loads, nt");
        run<1, 7, 0>(bufs, bytes, passes, out, 0, "LDS DMA");
        run<1, 7, 1>(bufs, bytes, passes, out, 0, "LDS DMA, nt");
        run<1, 7, 1>(bufs, bytes, passes, out, 4096, "LDS DMA, nt, 32 KiB LDS");
        run<2, 7, 1>(bufs, bytes, passes, out, 0, "LDS DMA, nt, 2 stages");
    }
    char* dsts[3];
    for (int r = 0; r < 3; ++r) { CK(hipMalloc(&dsts[r], bytes + 65536)); CK(hipMemset(dsts[r], 0, bytes + 65536)); }
    for (int passes : {1, 4, 16}) {
        run<3, 7, 0>(bufs, bytes, passes, out, 0, "write only", dsts);
        run<3, 7, 1>(bufs, bytes, passes, out, 0, "write only, nt", dsts);
        run<4, 7, 0>(bufs, bytes, passes, out, 0, "copy (DMA, LDS, stores)", dsts);
        run<4, 7, 1>(bufs, bytes, passes, out, 0, "copy, nt loads and stores", dsts);
        run<4, 3, 1>(bufs, bytes, passes * 2, out, 0, "copy, nt loads and stores", dsts);
        run<5, 7, 0>(bufs, bytes, passes, out, 0, "copy, stores behind next DMA", dsts);
        run<5, 7, 1>(bufs, bytes, passes, out, 0, "copy, stores behind next DMA, nt", dsts);
        run<5, 3, 1>(bufs, bytes, passes * 2, out, 0, "copy, stores behind next DMA, nt", dsts);
    }
    for (int passes : {2, 8}) {
        run<0, 14, 1>(bufs, bytes, passes, out, 0, "VGPR loads, nt");
        run<1, 14, 1>(bufs, bytes, passes, out, 0, "LDS DMA, nt");
    }
    for (int passes : {7, 28}) {
        run<0, 4, 1>(bufs, bytes, passes, out, 0, "VGPR loads, nt");
        run<1, 4, 1>(bufs, bytes, passes, out, 0, "LDS DMA, nt");
        run<2, 4, 1>(bufs, bytes, passes, out, 0, "LDS DMA, nt, 2 stages");
    }
    return 0;
}
