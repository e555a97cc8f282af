// What does a partial cache line cost in HBM traffic on gfx950?  A kernel reads SPAN bytes out of every 128-byte line of a
// 512 MiB buffer (16 bytes per lane, the lanes of a line adjacent), SPAN = 16, 32, 64, 96, 128; run under
//   rocprofv3 --kernel-trace --pmc FETCH_SIZE -- /tmp/fetch_granularity
// and compare FETCH_SIZE x 2 (the guide's gfx950 correction) per launch with SPAN / 128 x 512 MiB: equal = the memory side
// fetches 32- or 64-byte sectors, 512 MiB whatever SPAN = it fetches whole 128-byte lines.  (The mixture forward's parameter
// spans are 208 of every 416 bytes: profiles/r05_mixture_fwd_traffic.txt.)  Also prints the time of each launch.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

template <int SPAN>
__global__ __launch_bounds__(256) void read_spans(const f4* buf, long lines, float* out) {
    constexpr int PER = SPAN / 16;                       // lanes per line
    float acc = 0.f;
    const long tid = (long)blockIdx.x * blockDim.x + threadIdx.x, nthreads = (long)gridDim.x * blockDim.x;
    for (long i = tid; i < lines * PER; i += nthreads) {
        const long line = i / PER;
        const int part = (int)(i - line * PER);
        const f4 v = __builtin_nontemporal_load(buf + line * 8 + part);
        acc += v.x + v.y + v.z + v.w;
    }
    if (acc == 12345.678f) out[0] = acc;                 // keep the loads
}

template <int SPAN>
static void run(const f4* buf, long lines, float* out) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    read_spans<SPAN><<<4096, 256>>>(buf, lines, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    read_spans<SPAN><<<4096, 256>>>(buf, lines, out);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    printf("span %3d of 128 bytes: %7.1f us, %6.1f MiB asked for, %6.1f GB/s of asked-for bytes\n", SPAN, ms * 1e3, lines * (double)SPAN / 1048576.0,
           lines * (double)SPAN / (ms * 1e-3) / 1e9);
}

int main() {
    const long bytes = 512l << 20, lines = bytes / 128;
    f4* buf; float* out;
    CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&out, 64));
    CK(hipMemset(buf, 0, bytes));
    run<16>(buf, lines, out); run<32>(buf, lines, out); run<64>(buf, lines, out); run<96>(buf, lines, out); run<128>(buf, lines, out);
    return 0;
}
