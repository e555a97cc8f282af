// What does the DATA MOVEMENT of the fp32 mixture backward (S* compact: 16384 rows x 64 tokens, 312-byte parameter rows per token,
// 6 latents per token) cost, stream by stream?  The kernel with its arithmetic compiled out takes ~206 us (profiles/
// r06_mixture_bwd_floor.txt) where a plain copy of the same 327 MB in 7 KiB wave-passes takes 107-130 us.  This program rebuilds the
// kernel's access pattern from the plain copy upwards, one ingredient per flag, with the kernel's own decomposition: a persistent grid
// of 1024 workgroups x 4 waves, one unit of 4 rows = 256 tokens = 13 passes of 21 tokens (6552 bytes) per wave.
//   flag 1   passes on the rows' own byte grid (6552-byte passes: DMA from the span's 16-byte grid down, 4-byte stores at either end
//            of the write-back) instead of 7 KiB aligned passes
//   flag 2   the item lanes' small loads: z and g_zout (one float per lane, 3 of a token's 6 channels) and g_ldj
//   flag 4   the item lanes' g_z stores (one float per lane, 12 of each token's 24 bytes)
//   flag 8   the copied-through channels: one g_zout load and one g_z store per lane (the other 12 bytes)
//   flag 16  nontemporal write-back
//   flag 32  no write-back at all (read side alone)
//   flag 64 / 128  s_sleep 4 / 1 behind every store instruction of the write-back (do paced stores help the mixed stream?)
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/bwd_stream tools/microbench/bwd_stream.hip && /tmp/bwd_stream
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
typedef float f4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void glb_void_t;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

constexpr int kTok = 312, kTPP = 21, kD = 6, kDA = 3, kN = 64;
static int g_grid = 1024, g_rows = 4, g_extra_lds = 2048;      // argv: grid, rows per unit, extra LDS bytes per workgroup (occupancy)

template <int FLAGS>
__global__ __launch_bounds__(256) void stream(const char* nn, char* gnn, const float* z, const float* gzo, const float* gl, float* gz,
                                              long units, float* sink, int kRows) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    char* stage = smem + (size_t)wave * 7 * 1024;
    const int tli = lane / kDA, d = lane - tli * kDA;
    float acc = 0.f;
    for (long unit = (long)blockIdx.x * 4 + wave; unit < units; unit += (long)gridDim.x * 4) {
        const long tok0 = unit * kRows * kN;
        const int ntok = kRows * kN;
        for (int tp = 0; tp < ntok; tp += kTPP) {
            const int npt = min(kTPP, ntok - tp);
            const bool valid = tli < npt;
            const long tok = tok0 + tp + (valid ? tli : 0);
            float x = 0.f, g = 0.f, l = 0.f, ct = 0.f;
            if (FLAGS & 2) {
                if (valid) { x = z[tok * kD + d]; g = gzo[tok * kD + d]; l = gl[tok / kN]; }
            }
            if ((FLAGS & 8) && lane < npt * (kD - kDA)) {
                const int tk = lane / (kD - kDA), c = kDA + lane - tk * (kD - kDA);
                ct = gzo[(tok0 + tp + tk) * kD + c];
            }
            // ---- DMA of the pass
            const int ppu = (ntok + kTPP - 1) / kTPP;
            const char* src = (FLAGS & 1) ? nn + (tok0 + tp) * kTok : nn + (unit * ppu + tp / kTPP) * 7168;
            const int bytes = (FLAGS & 1) ? npt * kTok : 7168;
            const int off0 = __builtin_amdgcn_readfirstlane((int)(reinterpret_cast<uintptr_t>(src) & 15));
            const char* abase = src - off0;
            const int ni = (bytes + off0 + 1023) >> 10;
            for (int i = 0; i < ni; ++i)
                __builtin_amdgcn_global_load_lds((glb_void_t*)(abase + ((size_t)(i * 64 + lane) << 4)), (lds_void_t*)(stage + (i << 10)), 16, 0, 2);
            __builtin_amdgcn_s_waitcnt(0x0F70);
            __builtin_amdgcn_wave_barrier();
            acc += x + g + l;
            if ((FLAGS & 4) && valid) gz[tok * kD + d] = g + x * 1e-30f;
            // ---- write-back on the span's 16-byte grid
            if (!(FLAGS & 32)) {
                char* gdst = (FLAGS & 1) ? gnn + (tok0 + tp) * kTok : gnn + (unit * ppu + tp / kTPP) * 7168;
                const int head = (16 - off0) & 15;
                const int hb = min(head, bytes);
                if (lane * 4 < hb) *reinterpret_cast<float*>(gdst + lane * 4) = *reinterpret_cast<const float*>(stage + off0 + lane * 4);
                const int body = (bytes - hb) & ~15;
                for (int b = hb + lane * 16; b < hb + body; b += 1024) {
                    const f4 v = *reinterpret_cast<const f4*>(stage + off0 + b);
                    if (FLAGS & 16) __builtin_nontemporal_store(v, reinterpret_cast<f4*>(gdst + b));
                    else *reinterpret_cast<f4*>(gdst + b) = v;
                    if (FLAGS & 64) __builtin_amdgcn_s_sleep(4);          // paced stores: ~256 cycles between the write-back's instructions
                    if (FLAGS & 128) __builtin_amdgcn_s_sleep(1);
                }
                const int tb = hb + body + lane * 4;
                if (tb < bytes) *reinterpret_cast<float*>(gdst + tb) = *reinterpret_cast<const float*>(stage + off0 + tb);
            } else {
                acc += reinterpret_cast<const float*>(stage)[lane * 13];
            }
            if ((FLAGS & 8) && lane < npt * (kD - kDA)) {
                const int tk = lane / (kD - kDA), c = kDA + lane - tk * (kD - kDA);
                gz[(tok0 + tp + tk) * kD + c] = ct;
            }
            __builtin_amdgcn_s_waitcnt(0xC07F);
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}

struct Bufs { char* nn[2]; char* gnn[2]; float* z[2]; float* gzo[2]; float* gz; float* gl; float* sink; };

template <int FLAGS>
static void run(const Bufs& b, long units, const char* what) {
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const size_t lds = 4 * 7 * 1024 + g_extra_lds;
    for (int r = 0; r < 2; ++r) hipLaunchKernelGGL((stream<FLAGS>), dim3(g_grid), dim3(256), lds, 0, b.nn[r], b.gnn[r], b.z[r], b.gzo[r], b.gl, b.gz, units, b.sink, g_rows);
    CK(hipDeviceSynchronize());
    const int reps = 8;
    float best = 1e9f, sum = 0.f;
    for (int t = 0; t < 5; ++t) {
        CK(hipEventRecord(e0));
        for (int r = 0; r < reps; ++r) hipLaunchKernelGGL((stream<FLAGS>), dim3(g_grid), dim3(256), lds, 0, b.nn[r & 1], b.gnn[r & 1], b.z[r & 1], b.gzo[r & 1], b.gl, b.gz, units, b.sink, g_rows);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        best = fminf(best, ms / reps);
        sum += ms / reps;
    }
    printf("flags %2d  %-72s %7.1f us (best %7.1f)\n", FLAGS, what, sum / 5 * 1e3, best * 1e3);
}

int main(int argc, char** argv) {
    if (argc > 1) g_grid = atoi(argv[1]);
    if (argc > 2) g_rows = atoi(argv[2]);
    if (argc > 3) g_extra_lds = atoi(argv[3]);
    const int kRows = g_rows;
    const long B = 16384, tokens = B * kN, units = B / kRows;
    const size_t nnb = (size_t)units * ((kRows * kN + kTPP - 1) / kTPP) * 7168 + 65536;            // >= tokens * 312
    printf("grid %d, %d rows per unit (%ld units), LDS %d B per workgroup\n", g_grid, kRows, units, 4 * 7 * 1024 + g_extra_lds);
    Bufs b;
    for (int r = 0; r < 2; ++r) {
        CK(hipMalloc(&b.nn[r], nnb)); CK(hipMemset(b.nn[r], 0, nnb));
        CK(hipMalloc(&b.gnn[r], nnb)); CK(hipMemset(b.gnn[r], 0, nnb));
        CK(hipMalloc(&b.z[r], tokens * kD * 4)); CK(hipMemset(b.z[r], 0, tokens * kD * 4));
        CK(hipMalloc(&b.gzo[r], tokens * kD * 4)); CK(hipMemset(b.gzo[r], 0, tokens * kD * 4));
    }
    CK(hipMalloc(&b.gz, tokens * kD * 4)); CK(hipMalloc(&b.gl, B * 4)); CK(hipMemset(b.gl, 0, B * 4));
    CK(hipMalloc(&b.sink, 64));
    if (argc > 1) {
        run<17>(b, units, "copy, rows' grid, nontemporal stores");
        run<31>(b, units, "  the kernel's movement, nontemporal write-back");
        run<47>(b, units, "  the kernel's movement without the write-back");
        run<31 + 64>(b, units, "  the kernel's movement, nontemporal, stores paced (s_sleep 4)");
        run<31 + 128>(b, units, "  the kernel's movement, nontemporal, stores paced (s_sleep 1)");
        run<15 + 64>(b, units, "  the kernel's movement, plain stores, paced (s_sleep 4)");
        return 0;
    }
    for (int rep = 0; rep < 2; ++rep) {
        run<32>(b, units, "read side alone, 7 KiB aligned passes");
        run<33>(b, units, "read side alone, 6552-byte passes on the rows' grid");
        run<0>(b, units, "copy, 7 KiB aligned passes");
        run<16>(b, units, "copy, 7 KiB aligned passes, nontemporal stores");
        run<1>(b, units, "copy, 6552-byte passes on the rows' grid");
        run<17>(b, units, "copy, rows' grid, nontemporal stores");
        run<3>(b, units, "  + small loads (z, g_zout, g_ldj)");
        run<7>(b, units, "  + g_z stores of the item lanes");
        run<15>(b, units, "  + copied-through channels (load + store): the kernel's movement");
        run<31>(b, units, "  the kernel's movement, nontemporal write-back");
        run<47>(b, units, "  the kernel's movement without the write-back");
    }
    return 0;
}
