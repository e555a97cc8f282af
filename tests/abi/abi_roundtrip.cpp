// Torch-free user of the C ABI (include/cnf_hip.h): plain HIP allocations and scalar fp64 CPU loops of the same arithmetic as the
// check — (1) the affine coupling forward + NLL epilogue and the inverse (coupling_layer.py:53-63, task.py:96-118); (2) the
// mixture-CDF coupling forward on the reference layout and on the compact layout (same bits), and its inverse
// (mixture_cdf_layer.py:95-180); (3) the mixture-model encoder forward and the arg-max decode (linear_encoding.py:59-196).
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
    printf("ABI_C affine max errors: z %.2e ldj %.2e nll %.2e round-trip z %.2e ldj %.2e | sum %.6f vs %.6f count %.0f flags %d\n",
           worst_z, worst_l, worst_n, worst_rt, worst_lr, sums[0], total, sums[1], flags);
    bool ok = worst_z < 2e-5 && worst_l < 2e-4 && worst_n < 2e-5 && worst_rt < 1e-4 && worst_lr < 1e-4 &&
              fabs(sums[0] - total) < 1e-6 * fabs(total) + 1e-9 && sums[1] == B && flags == 0;

    // ---- (2) mixture-CDF coupling: K = 4 mixtures, channel mask [1,1,1,0,0,0] --------------------------------------------------
    {
        // (the compact layout's tensor must hold a multiple of 4 floats — the DMA's last 16-byte chunk — or the entry point declines with
        // CNF_ERR_UNSUPPORTED and the caller expands to the reference layout: 256 of the 257 rows here)
        const int K = 4, P = 2 + 3 * K, DA = D - D / 2, d0 = D / 2, B2 = 256;
        std::vector<float> mnn(E * P), mcomp((size_t)B * N * DA * P), msf(D), mmsf((size_t)D * K), mzo(E), mzc(E), mzr(E), ml(B), mlc(B), mlr(B);
        for (auto& v : mnn) v = 0.6f * frand(&seed);
        for (auto& v : msf) v = 0.2f * frand(&seed);
        for (auto& v : mmsf) v = 0.2f * frand(&seed);
        for (size_t t = 0; t < (size_t)B * N; ++t)
            for (int j = 0; j < DA * P; ++j) mcomp[t * DA * P + j] = mnn[(t * D + d0) * P + j];
        const int act[3] = {3, 4, 5};
        float *dmnn, *dmcomp, *dmsf, *dmmsf, *dmzo, *dmzc, *dmzr, *dml, *dmlc, *dmlr;
        CHECK(hipMalloc(&dmnn, E * P * 4)); CHECK(hipMalloc(&dmcomp, mcomp.size() * 4)); CHECK(hipMalloc(&dmsf, D * 4));
        CHECK(hipMalloc(&dmmsf, D * K * 4)); CHECK(hipMalloc(&dmzo, E * 4)); CHECK(hipMalloc(&dmzc, E * 4)); CHECK(hipMalloc(&dmzr, E * 4));
        CHECK(hipMalloc(&dml, B * 4)); CHECK(hipMalloc(&dmlc, B * 4)); CHECK(hipMalloc(&dmlr, B * 4));
        CHECK(hipMemcpy(dmnn, mnn.data(), E * P * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dmcomp, mcomp.data(), mcomp.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dmsf, msf.data(), D * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dmmsf, mmsf.data(), D * K * 4, hipMemcpyHostToDevice));
        rc = cnf_mixture_coupling(dz, dmnn, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, nullptr, dmzo, dml, nullptr, B2, N, D, K, 0, -1.0, 1.0, 0, dflags, st);
        if (rc != CNF_OK) { printf("mixture forward failed: %s\n", cnf_last_error()); return 1; }
        rc = cnf_mixture_coupling_compact(dz, dmcomp, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, nullptr, dmzc, dmlc, nullptr, B2, N, D, K, 0,
                                          -1.0, 1.0, 0, nullptr, 0, dflags, st);
        if (rc != CNF_OK) { printf("mixture forward (compact layout) failed: %s\n", cnf_last_error()); return 1; }
        rc = cnf_mixture_coupling(dmzo, dmnn, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, nullptr, dmzr, dmlr, nullptr, B2, N, D, K, 1, -1.0, 1.0, 0, dflags, st);
        if (rc != CNF_OK) { printf("mixture inverse failed: %s\n", cnf_last_error()); return 1; }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(mzo.data(), dmzo, E * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(mzc.data(), dmzc, E * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(mzr.data(), dmzr, E * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ml.data(), dml, B * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(mlc.data(), dmlc, B * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(mlr.data(), dmlr, B * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(&flags, dflags, 4, hipMemcpyDeviceToHost));
        double wz = 0, wl = 0, wrt = 0, wlr = 0;
        bool same = true;
        auto bound = [](double raw, double f) { return tanh(raw / fmax(f, 1.0)) * f; };
        for (int b = 0; b < B2; ++b) {
            double l = 0;
            for (int n = 0; n < N; ++n)
                for (int d = 0; d < D; ++d) {
                    const size_t i = ((size_t)b * N + n) * D + d;
                    same = same && mzo[i] == mzc[i];
                    wrt = fmax(wrt, fabs((double)mzr[i] - z[i]));
                    if (d < d0) { wz = fmax(wz, fabs((double)mzo[i] - z[i])); continue; }
                    const float* r = &mnn[i * P];
                    const double t = r[0], log_s = bound(r[1], exp((double)msf[d]));
                    double mx = -1e300, se = 0, cdf = 0, pdf = 0;
                    for (int k = 0; k < K; ++k) mx = fmax(mx, (double)r[2 + k]);
                    for (int k = 0; k < K; ++k) {
                        const double w = exp(r[2 + k] - mx), ls = bound(r[2 + 2 * K + k], exp((double)mmsf[d * K + k]));
                        const double zk = (z[i] - r[2 + K + k]) * exp(-ls), sg = 1.0 / (1.0 + exp(-zk));
                        se += w; cdf += w * sg; pdf += w * sg * (1.0 - sg) * exp(-ls);
                    }
                    const double u = cdf / se, y = log(u) - log1p(-u);
                    wz = fmax(wz, fabs((y + t) * exp(log_s) - mzo[i]));
                    l += log_s - log(u) - log1p(-u) + log(pdf / se);
                }
            wl = fmax(wl, fabs(l - ml[b]) / fmax(1.0, fabs(l)));
            wlr = fmax(wlr, fabs((double)ml[b] + mlr[b]) / fmax(1.0, fabs(l)));
            same = same && ml[b] == mlc[b];
        }
        printf("ABI_C mixture max errors: z %.2e ldj (rel) %.2e round-trip z %.2e ldj %.2e | compact layout bit-identical: %d flags %d\n",
               wz, wl, wrt, wlr, (int)same, flags);
        ok = ok && wz < 5e-5 && wl < 1e-4 && wrt < 2e-4 && wlr < 1e-4 && same && flags == 0;

        // backward (fp32 streaming kernel) on both layouts with g_zout = 1, g_ldj = 1: the compact rows are the reference layout's
        // transformed blocks bit for bit, its untransformed blocks are exact zeros, the copied-through channels pass g_zout, and
        // g_z of one element agrees with a central difference of (sum z' + ldj) through the forward entry point
        {
            const size_t nref = (size_t)B2 * N * D * P, ncmp = (size_t)B2 * N * DA * P, ne = (size_t)B2 * N * D;
            std::vector<float> ones(ne, 1.f), gz(ne), gzc(ne), gnn(nref), gnc(ncmp), gsf(D), gmsf((size_t)D * K), gsfc(D), gmsfc((size_t)D * K);
            float *done, *dgz, *dgzc, *dgnn, *dgnc, *dgsf, *dgmsf, *dgsfc, *dgmsfc, *dws;
            const int64_t wsf = cnf_bwd_workspace_floats(D + D * K);
            CHECK(hipMalloc(&done, ne * 4)); CHECK(hipMalloc(&dgz, ne * 4)); CHECK(hipMalloc(&dgzc, ne * 4)); CHECK(hipMalloc(&dgnn, nref * 4));
            CHECK(hipMalloc(&dgnc, ncmp * 4)); CHECK(hipMalloc(&dgsf, D * 4)); CHECK(hipMalloc(&dgmsf, D * K * 4)); CHECK(hipMalloc(&dgsfc, D * 4));
            CHECK(hipMalloc(&dgmsfc, D * K * 4)); CHECK(hipMalloc(&dws, (size_t)wsf * 4));
            CHECK(hipMemcpy(done, ones.data(), ne * 4, hipMemcpyHostToDevice));
            CHECK(hipMemset(dgnn, 0xff, nref * 4)); CHECK(hipMemset(dgnc, 0xff, ncmp * 4));
            rc = cnf_mixture_coupling_bwd_f32(dz, dmnn, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, done, done, dgz, dgnn, dgsf, dgmsf, dws,
                                              B2, N, D, K, -1.0, 1.0, 0, st);
            if (rc != CNF_OK) { printf("mixture backward failed: %s\n", cnf_last_error()); return 1; }
            rc = cnf_mixture_coupling_compact_bwd_f32(dz, dmcomp, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, done, done, dgzc, dgnc, dgsfc, dgmsfc, dws,
                                                      B2, N, D, K, -1.0, 1.0, 0, st);
            if (rc != CNF_OK) { printf("mixture backward (compact layout) failed: %s\n", cnf_last_error()); return 1; }
            CHECK(hipStreamSynchronize(st));
            CHECK(hipMemcpy(gz.data(), dgz, ne * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gzc.data(), dgzc, ne * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gnn.data(), dgnn, nref * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gnc.data(), dgnc, ncmp * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gsf.data(), dgsf, D * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gsfc.data(), dgsfc, D * 4, hipMemcpyDeviceToHost));
            CHECK(hipMemcpy(gmsf.data(), dgmsf, D * K * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(gmsfc.data(), dgmsfc, D * K * 4, hipMemcpyDeviceToHost));
            bool bsame = true, zeros = true, pass = true;
            for (size_t t = 0; t < (size_t)B2 * N; ++t) {
                for (int j = 0; j < d0 * P; ++j) zeros = zeros && gnn[t * D * P + j] == 0.f;
                for (int j = 0; j < DA * P; ++j) bsame = bsame && gnn[(t * D + d0) * P + j] == gnc[t * DA * P + j];
                for (int d = 0; d < D; ++d) {
                    bsame = bsame && gz[t * D + d] == gzc[t * D + d];
                    if (d < d0) pass = pass && gz[t * D + d] == 1.f;
                }
            }
            for (int d = 0; d < D; ++d) bsame = bsame && gsf[d] == gsfc[d];
            for (int i = 0; i < D * K; ++i) bsame = bsame && gmsf[i] == gmsfc[i];
            // central difference in z[0, 0, d0]: only row 0's outputs move
            const size_t ie = (size_t)d0;
            double fd[2];
            std::vector<float> zrow((size_t)N * D), lrow(1);
            for (int sgn = 0; sgn < 2; ++sgn) {
                const float h = 1e-2f, zv = z[ie] + (sgn ? h : -h);
                CHECK(hipMemcpy(dz + ie, &zv, 4, hipMemcpyHostToDevice));
                rc = cnf_mixture_coupling(dz, dmnn, dmsf, dmmsf, dmask, 1, D, act, 3, nullptr, 0, 0, nullptr, dmzo, dml, nullptr, 1, N, D, K, 0, -1.0, 1.0, 0, dflags, st);
                if (rc != CNF_OK) { printf("mixture forward (difference) failed: %s\n", cnf_last_error()); return 1; }
                CHECK(hipStreamSynchronize(st));
                CHECK(hipMemcpy(zrow.data(), dmzo, (size_t)N * D * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(lrow.data(), dml, 4, hipMemcpyDeviceToHost));
                double tot = lrow[0];
                for (float v : zrow) tot += v;
                fd[sgn] = tot;
            }
            CHECK(hipMemcpy(dz + ie, &z[ie], 4, hipMemcpyHostToDevice));
            const double num = (fd[1] - fd[0]) / 2e-2, ana = gz[ie];
            printf("ABI_C mixture backward: compact rows bit-identical %d, untransformed blocks zero %d, copied-through channels pass g_zout %d, "
                   "g_z[0,0,%d] %.5f vs central difference %.5f\n", (int)bsame, (int)zeros, (int)pass, d0, ana, num);
            ok = ok && bsame && zeros && pass && fabs(num - ana) < 2e-2 * fmax(1.0, fabs(ana));
        }
    }

    // ---- (3) mixture-model encoder: C = 7 classes, forward from given logistic noise, then the arg-max decode --------------------
    {
        const int C = 7, T = B * N;
        std::vector<long long> cat(T), dec(T);
        std::vector<float> eps((size_t)T * D), table((size_t)C * 2 * D), prior(C), ez((size_t)T * D), el(B);
        for (int t = 0; t < T; ++t) { seed = seed * 1664525u + 1013904223u; cat[t] = (seed >> 10) % C; }
        for (auto& v : eps) v = 0.3f * frand(&seed);
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d) { table[c * 2 * D + d] = 6.f * (c - 3) + 0.3f * frand(&seed); table[c * 2 * D + D + d] = 0.2f * frand(&seed); }
        for (int c = 0; c < C; ++c) prior[c] = -logf((float)C);
        long long *dcat, *ddec;
        float *deps, *dtable, *dprior, *dez, *del;
        CHECK(hipMalloc(&dcat, T * 8)); CHECK(hipMalloc(&ddec, T * 8)); CHECK(hipMalloc(&deps, eps.size() * 4)); CHECK(hipMalloc(&dtable, table.size() * 4));
        CHECK(hipMalloc(&dprior, C * 4)); CHECK(hipMalloc(&dez, ez.size() * 4)); CHECK(hipMalloc(&del, B * 4));
        CHECK(hipMemcpy(dcat, cat.data(), T * 8, hipMemcpyHostToDevice)); CHECK(hipMemcpy(deps, eps.data(), eps.size() * 4, hipMemcpyHostToDevice));
        CHECK(hipMemcpy(dtable, table.data(), table.size() * 4, hipMemcpyHostToDevice)); CHECK(hipMemcpy(dprior, prior.data(), C * 4, hipMemcpyHostToDevice));
        rc = cnf_encoder_forward((const int64_t*)dcat, deps, dtable, dprior, nullptr, 1.f, nullptr, dez, del, nullptr, B, N, D, C, sigma, log_sigma, dflags, st);
        if (rc != CNF_OK) { printf("encoder forward failed: %s\n", cnf_last_error()); return 1; }
        rc = cnf_encoder_decode(dez, dtable, dprior, (int64_t*)ddec, B, N, D, C, sigma, log_sigma, st);
        if (rc != CNF_OK) { printf("encoder decode failed: %s\n", cnf_last_error()); return 1; }
        CHECK(hipStreamSynchronize(st));
        CHECK(hipMemcpy(dec.data(), ddec, T * 8, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(ez.data(), dez, ez.size() * 4, hipMemcpyDeviceToHost));
        CHECK(hipMemcpy(el.data(), del, B * 4, hipMemcpyDeviceToHost)); CHECK(hipMemcpy(&flags, dflags, 4, hipMemcpyDeviceToHost));
        auto logp = [&](double x) { const double v = fabs(x / sigma); return -(v + 2.0 * log1p(exp(-v))) - log_sigma; };
        double wz = 0, wl = 0;
        long wrong = 0;
        for (int b = 0; b < B; ++b) {
            double l = 0;
            for (int n = 0; n < N; ++n) {
                const int t = b * N + n;
                const int c = (int)cat[t];
                double init = 0, ldjf = 0, zt[16];
                for (int d = 0; d < D; ++d) {
                    const double sc = tanh((double)table[c * 2 * D + D + d]);
                    init += logp(eps[(size_t)t * D + d]);
                    ldjf += sc;
                    zt[d] = (eps[(size_t)t * D + d] + table[c * 2 * D + d]) * exp(sc);
                    wz = fmax(wz, fabs(zt[d] - ez[(size_t)t * D + d]));
                }
                const double point = init - ldjf + prior[c];
                double mxs = -1e300, terms[16];
                for (int k = 0; k < C; ++k) {
                    double back = 0;
                    for (int d = 0; d < D; ++d) {
                        const double sc = tanh((double)table[k * 2 * D + D + d]);
                        back += logp(zt[d] * exp(-sc) - table[k * 2 * D + d]) - sc;
                    }
                    terms[k] = k == c ? point : back + prior[k];
                    mxs = fmax(mxs, terms[k]);
                }
                double se = 0;
                for (int k = 0; k < C; ++k) se += exp(terms[k] - mxs);
                l += (point - (mxs + log(se))) - (init - ldjf);
                wrong += dec[t] != cat[t];
            }
            wl = fmax(wl, fabs(l - el[b]) / fmax(1.0, fabs(l)));
        }
        printf("ABI_C encoder max errors: z %.2e ldj (rel) %.2e | decoded categories wrong: %ld of %d flags %d\n", wz, wl, wrong, T, flags);
        ok = ok && wz < 2e-5 && wl < 1e-4 && wrong == 0 && flags == 0;
    }
    printf(ok ? "ABI_C OK\n" : "ABI_C FAIL\n");
    return ok ? 0 : 1;
}
