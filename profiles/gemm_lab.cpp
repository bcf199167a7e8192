// In-process A/B harness for the GEMM kernels: links libadenet_hip.so and drives adn::gemm() directly, so that
// epilogue variants (bf16-only output, act'(Y) product, fused column sums, accumulate) of one shape are timed
// back to back on one GPU at one clock.  Build + run (MI355X):
//   hipcc -O2 -std=c++17 --offload-arch=gfx950 profiles/gemm_lab.cpp -Iip_avsr_amd/csrc -Iinclude \
//         -Lip_avsr_amd/csrc -ladenet_hip -Wl,-rpath,$PWD/ip_avsr_amd/csrc -o /tmp/gemm_lab && /tmp/gemm_lab
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
#include "adn_common.h"

using namespace adn;
#ifdef ADN_GEMM_STAMPS
extern "C" int adn_debug_gemm_stamps(unsigned long long*, int);   // libadenet_hip.so built with -DADN_GEMM_STAMPS
#endif

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Case { const char* name; int layout, M, N, K; int lean, acc, ygrad, colsum, biasrelu; };

static float* dalloc(size_t floats, bool fill) {
    float* p; CK(hipMalloc((void**)&p, floats * 4));
    if (fill) {
        std::vector<float> h(floats);
        unsigned s = 12345u + (unsigned)floats;
        for (size_t i = 0; i < floats; ++i) { s = s * 1664525u + 1013904223u; h[i] = ((s >> 8) & 0xffff) / 65536.f - 0.5f; }
        CK(hipMemcpy(p, h.data(), floats * 4, hipMemcpyHostToDevice));
    } else CK(hipMemset(p, 0, floats * 4));
    return p;
}

int main(int argc, char** argv) {
    const int R = 20800;
    std::vector<Case> cases = {
        {"fwd fc1 bias+relu lean", GEMM_NN, R, 2000, 1200, 1, 0, 0, 0, 1},
        {"fwd fc1 plain C+C16", GEMM_NN, R, 2000, 1200, 0, 0, 0, 0, 0},
        {"dX fc2 lean y colsum", GEMM_NN, R, 2000, 1000, 1, 0, 1, 1, 0},
        {"dX fc2 lean y", GEMM_NN, R, 2000, 1000, 1, 0, 1, 0, 0},
        {"dX fc2 lean", GEMM_NN, R, 2000, 1000, 1, 0, 0, 0, 0},
        {"dX fc3 lean y colsum", GEMM_NN, R, 1000, 500, 1, 0, 1, 1, 0},
        {"dX fc3 lean y", GEMM_NN, R, 1000, 500, 1, 0, 1, 0, 0},
        {"dX fc3 lean", GEMM_NN, R, 1000, 500, 1, 0, 0, 0, 0},
        {"dX bn lean y colsum", GEMM_NN, R, 500, 50, 1, 0, 1, 1, 0},
        {"dX bn lean", GEMM_NN, R, 500, 50, 1, 0, 0, 0, 0},
        {"xproj K=150 fp32 out", GEMM_NN, R, 1000, 150, 0, 0, 0, 0, 0},
        {"xproj K=250 fp32 out", GEMM_NN, R, 1000, 250, 0, 0, 0, 0, 0},
        {"xproj K=250 fp32 acc", GEMM_NN, R, 1000, 250, 0, 1, 0, 0, 0},
        {"xproj K=250 lean", GEMM_NN, R, 1000, 250, 1, 0, 0, 0, 0},
        {"fwd fc2 lean", GEMM_NN, R, 1000, 2000, 1, 0, 0, 0, 1},
        {"fwd fc3 lean", GEMM_NN, R, 500, 1000, 1, 0, 0, 0, 1},
        {"x3 fwd fc1 bias+relu lean", GEMM_NN, 3 * R, 2000, 1200, 1, 0, 0, 0, 1},
        {"x3 dX fc2 lean y colsum", GEMM_NN, 3 * R, 2000, 1000, 1, 0, 1, 1, 0},
        {"x3 fwd fc2 lean", GEMM_NN, 3 * R, 1000, 2000, 1, 0, 0, 0, 1},
        {"x3 fwd fc3 lean", GEMM_NN, 3 * R, 500, 1000, 1, 0, 0, 0, 1},
        {"x3 dX fc3 lean y colsum", GEMM_NN, 3 * R, 1000, 500, 1, 0, 1, 1, 0},
        {"dW fc1 TN acc", GEMM_TN, 1200, 2000, R, 0, 1, 0, 0, 0},
        {"dW fc2 TN acc", GEMM_TN, 2000, 1000, R, 0, 1, 0, 0, 0},
        {"dW fc3 TN acc", GEMM_TN, 1000, 500, R, 0, 1, 0, 0, 0},
        {"dW lstm TN acc", GEMM_TN, 250, 1000, R, 0, 1, 0, 0, 0},
        {"dW lstm-in TN acc", GEMM_TN, 150, 1000, R, 0, 1, 0, 0, 0},
        {"dW agg-cat TN", GEMM_TN, 768, 1000, R, 0, 0, 0, 0, 0},
        {"dW bn TN acc", GEMM_TN, 500, 50, R, 0, 1, 0, 0, 0},
    };
    const char* only = argc > 1 ? argv[1] : nullptr;
    if (only && !strcmp(only, "sweep")) {           // K sweep of one output shape: fixed cost vs per-stage cost
        cases.clear(); only = nullptr;
        static char names[32][32];
        int i = 0;
        for (int N : {1000, 500})
            for (int K : {64, 128, 256, 512, 1024, 2048}) {
                snprintf(names[i], 32, "sweep lean N=%d K=%d", N, K);
                cases.push_back({names[i], GEMM_NN, R, N, K, 1, 0, 0, 0, 0}); ++i;
            }
        for (int K : {64, 256, 1024}) {
            snprintf(names[i], 32, "sweep f32out N=1000 K=%d", K);
            cases.push_back({names[i], GEMM_NN, R, 1000, K, 0, 0, 0, 0, 0}); ++i;
        }
    }
    if (only && !strcmp(only, "msweep")) {          // M sweep around whole rounds of 256 workgroups: is the tail round real?
        cases.clear(); only = nullptr;
        static char names[64][40];
        int i = 0;
        for (int N : {1000, 2000})
            for (int mt : {128, 144, 152, 160, 161, 163, 168, 176, 184, 192}) {
                snprintf(names[i], 40, "msweep fwd N=%d mtiles=%d", N, mt);
                cases.push_back({names[i], GEMM_NN, mt * 128, N, N == 1000 ? 2000 : 1200, 1, 0, 0, 0, 1}); ++i;
            }
    }
    const int padto = getenv("LAB_PAD") ? atoi(getenv("LAB_PAD")) : 8;      // leading-dimension rounding (elements)
    auto pad = [padto](int n) { return (n + padto - 1) / padto * padto; };
    hipStream_t st; CK(hipStreamCreate(&st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("%-26s %3s %6s %6s %6s | %9s %9s %9s\n", "case", "lay", "M", "N", "K", "us", "TFLOP/s", "GB/s(out)");
    for (const auto& c : cases) {
        if (only && !strstr(c.name, only)) continue;
        const int ar = c.layout == GEMM_TN ? c.K : c.M, ac = pad(c.layout == GEMM_TN ? c.M : c.K);
        const int br = c.layout == GEMM_NT ? c.N : c.K, bc = pad(c.layout == GEMM_NT ? c.K : c.N);
        const int ldc = pad(c.N);
        float* A = dalloc((size_t)ar * ac, true); float* B = dalloc((size_t)br * bc, true);
        float* C = dalloc((size_t)c.M * ldc, false); float* Y = dalloc((size_t)c.M * ldc, true);
        float* bias = dalloc(ldc, true); float* cs = dalloc(ldc, false);
        const size_t wsf = (size_t)((c.M + 63) / 64) * ldc;
        float* ws = dalloc(wsf, false);
        void *A16, *B16, *C16, *Y16;
        CK(hipMalloc(&A16, (size_t)ar * ac * 2)); CK(hipMalloc(&B16, (size_t)br * bc * 2));
        CK(hipMalloc(&C16, (size_t)c.M * ldc * 2)); CK(hipMalloc(&Y16, (size_t)c.M * ldc * 2));
        to_bf16(A, A16, (size_t)ar * ac, st); to_bf16(B, B16, (size_t)br * bc, st); to_bf16(Y, Y16, (size_t)c.M * ldc, st);
        GemmArgs g;
        g.layout = c.layout; g.M = c.M; g.N = c.N; g.K = c.K; g.A = A; g.lda = ac; g.B = B; g.ldb = bc;
        g.C = c.lean ? nullptr : C; g.ldc = ldc; g.accumulate = c.acc; g.precision = ADN_PRECISION_BF16;
        g.A16 = A16; g.B16 = B16; g.C16 = (c.layout == GEMM_TN) ? nullptr : C16;
        if (c.ygrad) { g.Y = Y; g.ldy = ldc; g.Y16 = Y16; g.act_grad = ADN_ACT_RECTIFY; }
        int done = 0;
        if (c.colsum) { g.colsum = cs; g.colsum_done = &done; g.colsum_ws = ws; g.colsum_ws_floats = wsf; }
        if (c.biasrelu) { g.bias = bias; g.act = ADN_ACT_RECTIFY; }
        for (int i = 0; i < 3; ++i) if (gemm(g, st) != 0) { fprintf(stderr, "gemm failed: %s\n", c.name); return 1; }
#ifdef ADN_GEMM_STAMPS
        adn_debug_gemm_stamps(nullptr, 1);
#endif
        const int iters = 20;
        CK(hipEventRecord(e0, st));
        for (int i = 0; i < iters; ++i) gemm(g, st);
        CK(hipEventRecord(e1, st));
        CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        const double us = ms * 1e3 / iters;
#ifdef ADN_GEMM_STAMPS
        { unsigned long long st8[8];
          adn_debug_gemm_stamps(st8, 0);
          const double nst = (double)iters * 2 * ((c.K + 31) / 32) * 3;      // 2 stamping waves, ~3 tiles per workgroup (rough)
          printf("   stamps (us per stage, rough): wait %.3f  barrier %.3f  dma issue %.3f  reads+mfma %.3f\n", st8[0] / 100.0 / nst,
                 st8[1] / 100.0 / nst, st8[2] / 100.0 / nst, st8[3] / 100.0 / nst); }
#endif
        const double outb = (double)c.M * c.N * ((c.lean ? 0 : 4) + (g.C16 ? 2 : 0) + (c.ygrad ? 2 : 0) + (c.acc ? 4 : 0));
        printf("%-26s %3s %6d %6d %6d | %9.1f %9.1f %9.1f%s\n", c.name, c.layout == 0 ? "NN" : (c.layout == 1 ? "NT" : "TN"),
               c.M, c.N, c.K, us, 2.0 * c.M * c.N * c.K / us / 1e6, outb / us / 1e3, c.colsum && !done ? "  (colsum NOT fused)" : "");
        if (getenv("LAB_VERIFY")) {                  // sampled check against a double-precision dot product of the bf16 operands
            CK(hipMemsetAsync(C, 0, (size_t)c.M * ldc * 4, st));
            gemm(g, st);
            CK(hipStreamSynchronize(st));
            std::vector<unsigned short> hA((size_t)ar * ac), hB((size_t)br * bc), hC16((size_t)c.M * ldc), hY16((size_t)c.M * ldc);
            std::vector<float> hC((size_t)c.M * ldc), hb(ldc);
            CK(hipMemcpy(hA.data(), A16, hA.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hB.data(), B16, hB.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hC16.data(), C16, hC16.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hY16.data(), Y16, hY16.size() * 2, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost));
            CK(hipMemcpy(hb.data(), bias, ldc * 4, hipMemcpyDeviceToHost));
            auto f = [](unsigned short h) { unsigned u = (unsigned)h << 16; float x; memcpy(&x, &u, 4); return x; };
            double worst = 0; int bad = 0; unsigned rs = 777;
            const int nsamp = 4000;
            for (int sidx = 0; sidx < nsamp; ++sidx) {
                rs = rs * 1664525u + 1013904223u; int i = (rs >> 8) % c.M;
                rs = rs * 1664525u + 1013904223u; int j = (rs >> 8) % c.N;
                if (sidx < 64) { i = (sidx & 1) ? c.M - 1 - (sidx >> 1) : (sidx >> 1); }          // edges
                if (sidx >= 64 && sidx < 128) { j = (sidx & 1) ? c.N - 1 - ((sidx - 64) >> 1) : ((sidx - 64) >> 1); }
                double acc = 0;
                for (int k = 0; k < c.K; ++k) {
                    const float a = c.layout == GEMM_TN ? f(hA[(size_t)k * ac + i]) : f(hA[(size_t)i * ac + k]);
                    const float b = c.layout == GEMM_NT ? f(hB[(size_t)j * bc + k]) : f(hB[(size_t)k * bc + j]);
                    acc += (double)a * b;
                }
                if (c.biasrelu) { acc += hb[j]; if (acc < 0) acc = 0; }
                if (c.ygrad && !(f(hY16[(size_t)i * ldc + j]) > 0.f)) acc = 0;
                const double got = c.lean ? f(hC16[(size_t)i * ldc + j]) : hC[(size_t)i * ldc + j];
                const double tol = (c.lean ? 1.0e-2 : 2e-3) * (fabs(acc) + 1.0);
                const double err = fabs(got - acc);
                if (err > tol) { if (bad < 5) printf("   MISMATCH (%d,%d): got %g want %g\n", i, j, got, acc); ++bad; }
                if (err > worst) worst = err;
            }
            printf("   verify: %d/%d outside tolerance, worst abs err %.3g\n", bad, nsamp, worst);
        }
        (void)hipFree(A); (void)hipFree(B); (void)hipFree(C); (void)hipFree(Y); (void)hipFree(bias); (void)hipFree(cs); (void)hipFree(ws);
        (void)hipFree(A16); (void)hipFree(B16); (void)hipFree(C16); (void)hipFree(Y16);
    }
    return 0;
}
