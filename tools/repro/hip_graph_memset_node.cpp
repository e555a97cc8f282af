// Stand-alone reproducer (HIP runtime only, no PyTorch, none of this library) of the fault behind round 2's wrong-answer
// training graph (profiles/r03_graph_train_root_cause.txt): a MEMSET NODE captured into a hipGraph
// writes garbage from the second launch of the graph on, for sizes between 16 bytes and 4 KiB.
//
//   hipcc --offload-arch=gfx950 -O2 tools/repro/hip_graph_memset_node.cpp -o /tmp/memset_node && /tmp/memset_node
//
// Per size: capture { hipMemsetAsync(buf, 0, n); add_one<<<>>>(buf) } on a stream, instantiate, launch the graph four
// times; after every launch all n / 4 floats must read 1.0.  Also the same sequence without a graph (always right) and a
// graph whose zero fill is a kernel instead of a memset node (always right: the workaround this library uses).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#define CHECK(x)                                                                           \
    do {                                                                                   \
        hipError_t e_ = (x);                                                               \
        if (e_ != hipSuccess) {                                                            \
            std::printf("%s failed: %s\n", #x, hipGetErrorString(e_));                     \
            return 2;                                                                      \
        }                                                                                  \
    } while (0)

__global__ void add_one(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] += 1.f;
}
__global__ void zero_fill(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = 0.f;
}

static bool all_ones(const float* dev, size_t n, float* lo, float* hi) {
    std::vector<float> h(n);
    if (hipMemcpy(h.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost) != hipSuccess) return false;
    *lo = INFINITY; *hi = -INFINITY;
    bool ok = true;
    for (float v : h) {
        if (!(v == 1.f)) ok = false;
        if (!(v >= *lo)) *lo = v;                // NaN-aware: a NaN ends up in lo / hi
        if (!(v <= *hi)) *hi = v;
    }
    return ok;
}

int main() {
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    int faults = 0;
    float* other = nullptr;
    CHECK(hipMalloc(&other, 1 << 16));
    CHECK(hipMemset(other, 0, 1 << 16));
    const size_t sizes[] = {4, 8, 16, 64, 256, 1024, 4096, 8192, 65536, 1 << 20};
    for (int mode = 0; mode < 3; ++mode) {
        const char* what = mode == 0 ? "memset node in a graph" : (mode == 1 ? "zero-fill KERNEL in a graph" : "hipMemsetAsync without a graph");
        for (size_t bytes : sizes) {
            const size_t n = bytes / 4;
            float* buf = nullptr;
            CHECK(hipMalloc(&buf, bytes));
            std::vector<float> seven(n, 7.f);
            CHECK(hipMemcpy(buf, seven.data(), bytes, hipMemcpyHostToDevice));
            const unsigned grid = (unsigned)((n + 255) / 256 > 1024 ? 1024 : (n + 255) / 256);
            hipGraph_t graph = nullptr;
            hipGraphExec_t exec = nullptr;
            if (mode < 2) {
                CHECK(hipStreamBeginCapture(st, hipStreamCaptureModeGlobal));
                if (mode == 0) CHECK(hipMemsetAsync(buf, 0, bytes, st));
                else hipLaunchKernelGGL(zero_fill, dim3(grid), dim3(256), 0, st, buf, n);
                hipLaunchKernelGGL(add_one, dim3(grid), dim3(256), 0, st, buf, n);
                CHECK(hipStreamEndCapture(st, &graph));
                CHECK(hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0));
            }
            std::printf("%-32s %8zu bytes:", what, bytes);
            for (int launch = 0; launch < 4; ++launch) {
                if (mode < 2) {
                    CHECK(hipGraphLaunch(exec, st));
                } else {
                    CHECK(hipMemsetAsync(buf, 0, bytes, st));
                    hipLaunchKernelGGL(add_one, dim3(grid), dim3(256), 0, st, buf, n);
                }
                CHECK(hipStreamSynchronize(st));
                // unrelated work between the launches, as any real program has it: other kernels with other arguments,
                // on the same and on the null stream, an allocation and its release
                for (int r = 0; r < 8; ++r) {
                    hipLaunchKernelGGL(add_one, dim3(4), dim3(256), 0, st, other, (size_t)1024);
                    hipLaunchKernelGGL(zero_fill, dim3(2), dim3(256), 0, 0, other + 1024, (size_t)(100 + r));
                }
                float* tmp = nullptr;
                CHECK(hipMalloc(&tmp, 1 << 16));
                CHECK(hipMemsetAsync(tmp, 0x7f, 1 << 16, st));
                CHECK(hipDeviceSynchronize());
                CHECK(hipFree(tmp));
                float lo, hi;
                const bool ok = all_ones(buf, n, &lo, &hi);
                std::printf("  launch %d %s", launch + 1, ok ? "ok" : "WRONG");
                if (!ok) {
                    std::printf(" (min %g max %g)", lo, hi);
                    ++faults;
                }
            }
            std::printf("\n");
            if (exec) CHECK(hipGraphExecDestroy(exec));
            if (graph) CHECK(hipGraphDestroy(graph));
            CHECK(hipFree(buf));
        }
    }
    std::printf("%s: %d wrong read-backs\n", faults ? "FAULT REPRODUCED" : "no fault on this stack", faults);
    return 0;
}
