// Torch-free user of the C ABI (include/cnf_hip.h): plain HIP allocations, the affine coupling forward + NLL epilogue,
// the inverse, and a scalar CPU loop of the same arithmetic (coupling_layer.py:53-63, task.py:96-118) as the check.
// Built and run by tests/test_gpu_parity.py::test_c_abi_without_torch:
//   hipcc --offload-arch=gfx950 abi_roundtrip.cpp -I include -L categoricalnf_amd/lib -lcnf_hip -Wl,-rpath,<lib dir>
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

#include "cnf_hip.h"

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); return 2; } } while (0)

static float frand(unsigned* s) {            // LCG -> roughly N(0,1) by summing uniforms
    float acc = 0.f;
    for (int i = 0; i < 4; ++i) { *s = *s * 1664525u + 1013904223u; acc += (float)((*s >> 8) & 0xffffff) / 16777216.f; }
    return (acc - 2.f) * 1.7320508f;
}

int main() {
    const int B = 257, N = 19, D = 6;
    const size_t E = (size_t)B * N * D;
    std::vector<float> z(E), nn(2 * E), sf(D), mask(D), zo(E), ldj(B), nll(B), zr(E), ldjr(B);
    unsigned seed = 1234u;
    for (auto& v : z) v = frand(&seed);
    for (auto& v : nn) v = 0.5f * frand(&seed);
    for (int d = 0; d < D; ++d) { sf[d] = 0.2f * frand(&seed); mask[d] = d < D / 2 ? 1.f : 0.f; }
    float *dz, *dnn, *dsf, *dmask, *dzo, *dldj, *dnll, *dzr, *dldjr;
    int* dflags;
    double* dsums;
    CHECK(hipMalloc(&dz, E * 4)); CHECK(hipMalloc(&dnn, 2 * E * 4)); CHECK(hipMalloc(&dsf, D * 4)); CHECK(hipMalloc(&dmask, D * 4));
    CHECK(hipMalloc(&dzo, E * 4)); CHECK(hipMalloc(&dldj, B * 4)); CHECK(hipMalloc(&dnll, B * 4)); CHECK(hipMalloc(&dzr, E * 4));
    CHECK(hipMalloc(&dldjr, B * 4)); CHECK(hipMalloc(&dflags, 4)); CHECK(hipMalloc(&dsums, 16));
    CHECK(hipMemcpy(dz, z.data(), E * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dnn, nn.data(), 2 * E * 4, hipMemcpyHostToDevice));
    CHECK(hipMemcpy(dsf, sf.data(), D * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dmask, mask.data(), D * 4, hipMemcpyHostToDevice));
    CHECK(hipMemset(dflags, 0, 4));
    hipStream_t st;
    CHECK(hipStreamCreate(&st));
    const float sigma = 1.f / 1.81f, log_sigma = logf(sigma);
    if (cnf_abi_version() != 1) { printf("unexpected ABI version %d\n", cnf_abi_version()); return 1; }
    int rc = cnf_affine_coupling_nll(dz, dnn, dsf, dmask, 1, D, nullptr, dzo, dldj, nullptr, nullptr, nullptr, dnll, dsums,
                                     B, N, D, sigma, log_sigma, dflags, st);
    if (rc != CNF_OK) { printf("forward failed: %s\n", cnf_last_error()); return 1; }
    rc = cnf_affine_coupling(dzo, dnn, dsf, dmask, 1, D, dldj, dzr, dldjr, B, N, D, /*reverse=*/1, dflags, st);
    if (rc != CNF_OK) { printf("inverse failed: %s\n", cnf_last_error()); return 1; }
    CHECK(hipStreamSynchronize(st));
    double sums[2];
    int flags;
    CHECK(hipMemcpy(zo.data(), dzo, E * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ldj.data(), dldj, B * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(nll.data(), dnll, B * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(zr.data(), dzr, E * 4, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(ldjr.data(), dldjr, B * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(sums, dsums, 16, hipMemcpyDeviceToHost));
    CHECK(hipMemcpy(&flags, dflags, 4, hipMemcpyDeviceToHost));
    // scalar restatement
    double worst_z = 0, worst_l = 0, worst_n = 0, worst_rt = 0, worst_lr = 0, total = 0;
    for (int b = 0; b < B; ++b) {
        double l = 0, lp = 0;
        for (int n = 0; n < N; ++n)
            for (int d = 0; d < D; ++d) {
                const size_t i = ((size_t)b * N + n) * D + d;
                const double f = exp((double)sf[d]), keep = 1.0 - mask[d];
                const double s = tanh(nn[2 * i] / fmax(f, 1.0)) * f * keep, t = nn[2 * i + 1] * keep;
                const double o = (z[i] + t) * exp(s);
                l += s;
                const double v = fabs(o / sigma);
                lp += -(v + 2.0 * log1p(exp(-v))) - log_sigma;
                worst_z = fmax(worst_z, fabs(o - zo[i]));
                worst_rt = fmax(worst_rt, fabs((double)zr[i] - z[i]));
            }
        const double want = (-l - lp) / N;
        worst_l = fmax(worst_l, fabs(l - ldj[b]));
        worst_n = fmax(worst_n, fabs(want - nll[b]));
        worst_lr = fmax(worst_lr, fabs((double)ldjr[b]));
        total += nll[b];
    }
    printf("ABI_C max errors: z %.2e ldj %.2e nll %.2e round-trip z %.2e ldj %.2e | sum %.6f vs %.6f count %.0f flags %d\n",
           worst_z, worst_l, worst_n, worst_rt, worst_lr, sums[0], total, sums[1], flags);
    const bool ok = worst_z < 2e-5 && worst_l < 2e-4 && worst_n < 2e-5 && worst_rt < 1e-4 && worst_lr < 1e-4 &&
                    fabs(sums[0] - total) < 1e-6 * fabs(total) + 1e-9 && sums[1] == B && flags == 0;
    printf(ok ? "ABI_C OK\n" : "ABI_C FAIL\n");
    return ok ? 0 : 1;
}
