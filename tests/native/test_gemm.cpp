// Native (no Python) check of the MFMA GEMM against a double-precision host loop, plus a first
// throughput reading on the three GEMM shapes of the ml-20m training step.  Runs in seconds.
#include "../../rectorch_amd/csrc/rtx_gemm.h"

#include <math.h>
#include <stdlib.h>
#include <vector>

void rtx_set_error(const char* fmt, ...);
const char* rtx_last_error_str();

static uint32_t rng_state = 12345;
static float frand()
{
    rng_state = rng_state * 1664525u + 1013904223u;
    return ((rng_state >> 8) * (1.0f / 16777216.0f)) * 2.f - 1.f;
}

#define CK(x)                                                              \
    do {                                                                   \
        hipError_t e = (x);                                                \
        if (e != hipSuccess) {                                             \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e), __FILE__, __LINE__); \
            exit(2);                                                       \
        }                                                                  \
    } while (0)

template <typename T>
static int run_case(const char* name, int M, int N, int K, int splits, int epi, int M_real, int N_real, int shape)
{
    // padded operand buffers
    std::vector<T> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<double> dA((size_t)M * K), dB((size_t)N * K);
    for (size_t i = 0; i < hA.size(); ++i) { hA[i] = Elem<T>::from(frand()); dA[i] = Elem<T>::to(hA[i]); }
    for (size_t i = 0; i < hB.size(); ++i) { hB[i] = Elem<T>::from(frand() * 0.5f + 0.1f); dB[i] = Elem<T>::to(hB[i]); }
    std::vector<float> hbias(N);
    for (int i = 0; i < N; ++i) hbias[i] = frand();
    T *A, *B;
    float *C, *bias, *gb;
    const long ldc = (epi == RTX_EPI_STORE) ? N : N_real;
    const size_t csz = (epi == RTX_EPI_STORE) ? (size_t)splits * M * N : (size_t)M_real * N_real;
    CK(hipMalloc(&A, hA.size() * sizeof(T)));
    CK(hipMalloc(&B, hB.size() * sizeof(T)));
    CK(hipMalloc(&C, csz * sizeof(float)));
    CK(hipMalloc(&bias, N * sizeof(float)));
    CK(hipMalloc(&gb, M * sizeof(float)));
    CK(hipMemcpy(A, hA.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), N * sizeof(float), hipMemcpyHostToDevice));
    CK(hipMemset(C, 0xff, csz * sizeof(float)));
    CK(hipMemset(gb, 0xff, M * sizeof(float)));
    RtxGemm g = {};
    g.A = A; g.B = B; g.lda = K; g.ldb = K;
    int bm, bn;
    rtx_gemm_tile_dims(shape, &bm, &bn);
    g.tile_shape = shape; g.m_tiles = M / bm; g.n_tiles = N / bn;
    g.k_slices = (int)((size_t)K * sizeof(T) / 128);
    g.splits = splits; g.C = C; g.ldc = ldc; g.slab_stride = (long)M * N;
    g.bias = bias; g.gbias = gb; g.M_real = M_real; g.N_real = N_real;
    int rc = rtx_gemm_launch(g, sizeof(T) == 2, epi, 0);
    if (rc) { printf("[%s] launch failed rc=%d\n", name, rc); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hC(csz), hgb(M);
    CK(hipMemcpy(hC.data(), C, csz * sizeof(float), hipMemcpyDeviceToHost));
    CK(hipMemcpy(hgb.data(), gb, M * sizeof(float), hipMemcpyDeviceToHost));
    double max_err = 0, max_ref = 0;
    long bad = 0;
    int printed = 0;
    const int Mc = (epi == RTX_EPI_STORE) ? M : M_real;
    const int Nc = (epi == RTX_EPI_STORE) ? N : (epi == RTX_EPI_GRAD ? N_real + 1 : N_real);
    for (int m = 0; m < Mc; ++m)
        for (int n = 0; n < Nc; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += dA[(size_t)m * K + k] * dB[(size_t)n * K + k];
            double got;
            if (epi == RTX_EPI_STORE) {
                got = 0;
                for (int s = 0; s < splits; ++s) got += hC[(size_t)s * M * N + (size_t)m * N + n];
            } else if (epi == RTX_EPI_BIAS_ROWS) {
                ref += hbias[n];
                got = hC[(size_t)m * ldc + n];
            } else {
                got = (n < N_real) ? hC[(size_t)m * N_real + n] : hgb[m];
            }
            double err = fabs(got - ref);
            if (!(err <= 1e-4 * sqrt((double)K))) {
                ++bad;
                if (printed++ < 5) printf("   mismatch (%d,%d): got %.6f ref %.6f\n", m, n, got, ref);
            }
            if (err > max_err) max_err = err;
            if (fabs(ref) > max_ref) max_ref = fabs(ref);
        }
    printf("[%s] %s tile%d M=%d N=%d K=%d splits=%d epi=%d  max_err=%.3e (max|ref|=%.2f) bad=%ld -> %s\n", name,
           sizeof(T) == 2 ? "bf16" : "f32 ", shape, M, N, K, splits, epi, max_err, max_ref, bad, bad ? "FAIL" : "ok");
    hipFree(A); hipFree(B); hipFree(C); hipFree(bias); hipFree(gb);
    return bad != 0;
}

template <typename T>
static void perf_case(const char* name, int M, int N, int K, int splits, int epi, int shape)
{
    T *A, *B;
    float* C;
    std::vector<T> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = Elem<T>::from(frand());
    for (auto& v : hB) v = Elem<T>::from(frand());
    CK(hipMalloc(&A, hA.size() * sizeof(T)));
    CK(hipMalloc(&B, hB.size() * sizeof(T)));
    CK(hipMalloc(&C, (size_t)splits * M * N * sizeof(float)));
    CK(hipMemcpy(A, hA.data(), hA.size() * sizeof(T), hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * sizeof(T), hipMemcpyHostToDevice));
    RtxGemm g = {};
    g.A = A; g.B = B; g.lda = K; g.ldb = K;
    int bm, bn;
    rtx_gemm_tile_dims(shape, &bm, &bn);
    g.tile_shape = shape; g.m_tiles = M / bm; g.n_tiles = N / bn;
    g.k_slices = (int)((size_t)K * sizeof(T) / 128);
    g.splits = splits; g.C = C; g.ldc = N; g.slab_stride = (long)M * N;
    g.M_real = M; g.N_real = (epi == RTX_EPI_GRAD) ? N - 1 : N;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rtx_gemm_launch(g, sizeof(T) == 2, epi, 0);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) rtx_gemm_launch(g, sizeof(T) == 2, epi, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1000.0 / it;
    printf("[perf %s] %s M=%d N=%d K=%d splits=%d tile%d: %.1f us  %.1f TFLOP/s\n", name, sizeof(T) == 2 ? "bf16" : "f32 ", M, N, K,
           splits, shape, us, 2.0 * M * N * K / us * 1e-6);
    hipFree(A); hipFree(B); hipFree(C);
}

int main(int argc, char** argv)
{
    int fails = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  gcnArch=%s\n", prop.name, prop.multiProcessorCount, prop.gcnArchName);
    for (int shape = 0; shape < 3; ++shape) {   // 128x128, 256x128, 128x256
        fails += run_case<bf16_t>("store", 512, 768, 704, 1, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<bf16_t>("splitk3", 512, 768, 704, 3, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<bf16_t>("splitk5", 256, 256, 704, 5, RTX_EPI_STORE, 256, 256, shape);
        fails += run_case<bf16_t>("splitk11", 256, 512, 1408, 11, RTX_EPI_STORE, 256, 512, shape);
        fails += run_case<bf16_t>("bias", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, shape);
        fails += run_case<bf16_t>("grad", 768, 512, 512, 1, RTX_EPI_GRAD, 700, 300, shape);
        fails += run_case<bf16_t>("grad-tall", 2304, 256, 128, 1, RTX_EPI_GRAD, 2300, 200, shape);
        fails += run_case<bf16_t>("bias-wide", 256, 2304, 64, 1, RTX_EPI_BIAS_ROWS, 250, 2300, shape);
        fails += run_case<float>("store", 512, 768, 352, 1, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<float>("splitk3", 512, 768, 352, 3, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<float>("bias", 512, 768, 320, 1, RTX_EPI_BIAS_ROWS, 410, 701, shape);
        fails += run_case<float>("grad", 768, 512, 256, 1, RTX_EPI_GRAD, 700, 300, shape);
    }
    if (argc > 1) {
        // ml-20m step shapes: fwd-1 / dH3 (skinny, split-K), logits, dW4 / dW1
        perf_case<bf16_t>("fwd1", 512, 640, 20224, 24, RTX_EPI_STORE, 0);
        for (int s : {8, 16, 24, 32}) perf_case<bf16_t>("fwd1", 512, 640, 20224, s, RTX_EPI_STORE, 1);
        for (int shape : {0, 1, 2}) perf_case<bf16_t>("logits", 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS, shape);
        for (int shape : {0, 1}) perf_case<bf16_t>("dW4", 20224, 640, 512, 1, RTX_EPI_GRAD, shape);
        for (int shape : {0, 2}) perf_case<bf16_t>("dW1", 640, 20224, 512, 1, RTX_EPI_GRAD, shape);
        for (int K : {64, 128, 320, 640, 1280, 2560}) perf_case<bf16_t>("logitsK", 512, 20224, K, 1, RTX_EPI_BIAS_ROWS, 0);
        for (int K : {64, 640, 2560}) perf_case<bf16_t>("logitsK-store", 512, 20224, K, 1, RTX_EPI_STORE, 0);
        for (int K : {64, 512, 2048}) perf_case<bf16_t>("dW4K", 20224, 640, K, 1, RTX_EPI_GRAD, 0);
        perf_case<bf16_t>("small", 512, 512, 640, 1, RTX_EPI_STORE, 0);
        perf_case<bf16_t>("small-s5", 512, 512, 640, 5, RTX_EPI_STORE, 0);
        for (int shape : {0, 1, 2}) perf_case<bf16_t>("sq4k", 4096, 4096, 4096, 1, RTX_EPI_STORE, shape);
        perf_case<float>("fwd1", 512, 640, 20224, 24, RTX_EPI_STORE, 1);
        for (int shape : {0, 1}) perf_case<float>("logits", 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS, shape);
        perf_case<float>("dW4", 20224, 640, 512, 1, RTX_EPI_GRAD, 1);
        for (int shape : {0, 1}) perf_case<float>("sq4k", 4096, 4096, 4096, 1, RTX_EPI_STORE, shape);
    }
    printf("%s (%d failing cases)\n", fails ? "GEMM TESTS FAILED" : "GEMM TESTS PASSED", fails);
    return fails ? 1 : 0;
}
