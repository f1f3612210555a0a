// Native (no Python) check of the MFMA GEMM against a double-precision host loop, plus a first
// throughput reading on the three GEMM shapes of the ml-20m training step.  Runs in seconds.
#include "../../rectorch_amd/csrc/rtx_gemm.h"
#ifdef RTX_GEMM_ABLATE
int rtx_gemm_ablate_launch(const RtxGemm& g, int abl, hipStream_t st);
#endif

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
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


// ---- ds_read_b64_tr_b16 semantics probe: with linear per-lane addresses (lane * 8 bytes) lane l, element j must receive
//      lds[(l & 15) + j * 16 + (l >> 4) * 64] (16-bit elements) -- what gemm_dma.hip / dw_adam.hip are built on ------------------
__global__ void k_tr_probe(unsigned short* out)
{
    __shared__ __attribute__((aligned(16))) unsigned short lds[256];
    for (int i = threadIdx.x; i < 256; i += 64) lds[i] = (unsigned short)i;
    __syncthreads();
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2;
    u32x2 v;
    const unsigned addr = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned short*)lds + threadIdx.x * 8;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=&v"(v) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = (unsigned short)(v[0] & 0xffff);
    out[threadIdx.x * 4 + 1] = (unsigned short)(v[0] >> 16);
    out[threadIdx.x * 4 + 2] = (unsigned short)(v[1] & 0xffff);
    out[threadIdx.x * 4 + 3] = (unsigned short)(v[1] >> 16);
}

static int tr_probe()
{
    unsigned short* d;
    CK(hipMalloc(&d, 256 * 2));
    hipLaunchKernelGGL(k_tr_probe, dim3(1), dim3(64), 0, 0, d);
    CK(hipDeviceSynchronize());
    unsigned short h[256];
    CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    int bad = 0;
    for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 4; ++j) bad += h[l * 4 + j] != (unsigned short)((l & 15) + j * 16 + (l >> 4) * 64);
    printf("[tr-probe] ds_read_b64_tr_b16: %s\n", bad ? "UNEXPECTED MAPPING" : "ok");
    if (bad) {
        for (int l = 0; l < 64; ++l) printf("   lane %2d: %3d %3d %3d %3d\n", l, h[l * 4], h[l * 4 + 1], h[l * 4 + 2], h[l * 4 + 3]);
    }
    hipFree(d);
    return bad != 0;
}

static int g_xcd_block = 0;   // RtxGemm::xcd_block of the LDS-DMA cases below (second argument "xcd")

// ---- LDS-DMA GEMM (gemm_dma.hip): NT / NN, store (split-K) and bias (+ log-sum-exp partials) epilogues -------------------------
static int run_dma_case(const char* name, int form, int cfg, int M, int N, int K, int splits, int epi, int M_real, int N_real, int odd_ld)
{
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<double> dA((size_t)M * K), dB((size_t)N * K);   // dB[n][k] whatever the storage form
    for (size_t i = 0; i < hA.size(); ++i) { hA[i] = f32_to_bf16(frand()); dA[i] = bf16_to_f32(hA[i]); }
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const bf16_t v = f32_to_bf16(frand() * 0.5f + 0.1f);
            dB[(size_t)n * K + k] = bf16_to_f32(v);
            if (form == RTX_FORM_NT) hB[(size_t)n * K + k] = v; else hB[(size_t)k * N + n] = v;
        }
    std::vector<float> hbias(N);
    for (int i = 0; i < N; ++i) hbias[i] = frand();
    const long ldc = (epi == RTX_EPI_STORE) ? N : (odd_ld ? N_real : N);
    const size_t csz = (epi == RTX_EPI_STORE) ? (size_t)splits * M * N : (size_t)M_real * ldc;
    bf16_t *A, *B;
    float *C, *bias;
    float2* part;
    const int strips = N / 64;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2));
    CK(hipMalloc(&C, csz * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&part, (size_t)M * strips * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0xff, csz * 4));
    CK(hipMemset(part, 0xff, (size_t)M * strips * 8));
    RtxGemm g = {};
    g.form = form; g.A = A; g.B = B; g.lda = K; g.ldb = (form == RTX_FORM_NT) ? K : N;
    int bm, bn;
    rtx_gemm_dma_tile_dims(cfg, &bm, &bn);
    g.tile_shape = cfg; g.m_tiles = M / bm; g.n_tiles = N / bn; g.k_slices = K / 64; g.splits = splits;
    g.xcd_block = (splits == 1) ? g_xcd_block : 0;
    g.C = C; g.ldc = ldc; g.slab_stride = (long)M * N; g.bias = bias; g.M_real = M_real; g.N_real = N_real;
    if (epi == RTX_EPI_BIAS_ROWS) { g.lse_part = part; g.lse_ld = strips; }
    int rc = rtx_gemm_dma_launch(g, epi, 0);
    if (rc) { printf("[dma %s] launch failed rc=%d: %s\n", name, rc, rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hC(csz);
    std::vector<float2> hp((size_t)M * strips);
    CK(hipMemcpy(hC.data(), C, csz * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hp.data(), part, hp.size() * 8, hipMemcpyDeviceToHost));
    double max_err = 0, max_lse = 0;
    long bad = 0;
    int printed = 0;
    const int Mc = (epi == RTX_EPI_STORE) ? M : M_real, Nc = (epi == RTX_EPI_STORE) ? N : N_real;
    std::vector<double> refrow(N);
    for (int m = 0; m < Mc; ++m) {
        for (int n = 0; n < Nc; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += dA[(size_t)m * K + k] * dB[(size_t)n * K + k];
            double got;
            if (epi == RTX_EPI_STORE) {
                got = 0;
                for (int s2 = 0; s2 < splits; ++s2) got += hC[(size_t)s2 * M * N + (size_t)m * N + n];
            } else {
                ref += hbias[n];
                got = hC[(size_t)m * ldc + n];
            }
            refrow[n] = ref;
            const double err = fabs(got - ref);
            if (!(err <= 1e-4 * sqrt((double)K))) {
                ++bad;
                if (printed++ < 5) printf("   mismatch (%d,%d): got %.6f ref %.6f\n", m, n, got, ref);
            }
            max_err = std::max(max_err, err);
        }
        if (epi == RTX_EPI_BIAS_ROWS) {   // row log-sum-exp from the strip partials
            double mx = -1e300;
            for (int n = 0; n < Nc; ++n) mx = std::max(mx, refrow[n]);
            double se = 0;
            for (int n = 0; n < Nc; ++n) se += exp(refrow[n] - mx);
            const double ref_lse = mx + log(se);
            double gm = -1e300;
            for (int q = 0; q < strips; ++q) if (hp[(size_t)m * strips + q].y > 0.f) gm = std::max(gm, (double)hp[(size_t)m * strips + q].x);
            double gs = 0;
            for (int q = 0; q < strips; ++q) if (hp[(size_t)m * strips + q].y > 0.f) gs += hp[(size_t)m * strips + q].y * exp(hp[(size_t)m * strips + q].x - gm);
            const double e2 = fabs(gm + log(gs) - ref_lse);
            max_lse = std::max(max_lse, e2);
            if (!(e2 <= 1e-4 * sqrt((double)K))) { ++bad; if (printed++ < 5) printf("   lse mismatch row %d: got %.6f ref %.6f\n", m, gm + log(gs), ref_lse); }
        }
    }
    // BIAS: nothing may be written outside the valid block
    if (epi == RTX_EPI_BIAS_ROWS && !odd_ld) {
        for (int m = 0; m < M_real; ++m)
            for (int n = N_real; n < N; ++n) {
                uint32_t u; memcpy(&u, &hC[(size_t)m * ldc + n], 4);
                if (u != 0xffffffffu) { ++bad; if (printed++ < 5) printf("   pad written at (%d,%d)\n", m, n); }
            }
    }
    printf("[dma %s] %s cfg%d M=%d N=%d K=%d splits=%d epi=%d  max_err=%.3e lse_err=%.3e bad=%ld -> %s\n", name, form == RTX_FORM_NT ? "NT" : "NN", cfg,
           M, N, K, splits, epi, max_err, max_lse, bad, bad ? "FAIL" : "ok");
    hipFree(A); hipFree(B); hipFree(C); hipFree(bias); hipFree(part);
    return bad != 0;
}

static void perf_dma(const char* name, int form, int cfg, int M, int N, int K, int splits, int epi)
{
    bf16_t *A, *B;
    float *C, *bias;
    float2* part;
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = f32_to_bf16(frand());
    for (auto& v : hB) v = f32_to_bf16(frand());
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2));
    CK(hipMalloc(&C, (size_t)splits * M * N * 4)); CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&part, (size_t)M * (N / 64) * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4));
    RtxGemm g = {};
    g.form = form; g.A = A; g.B = B; g.lda = K; g.ldb = (form == RTX_FORM_NT) ? K : N;
    int bm, bn;
    rtx_gemm_dma_tile_dims(cfg, &bm, &bn);
    g.tile_shape = cfg; g.m_tiles = M / bm; g.n_tiles = N / bn; g.k_slices = K / 64; g.splits = splits;
    g.C = C; g.ldc = N; g.slab_stride = (long)M * N; g.bias = bias; g.M_real = M - 12; g.N_real = N - 116;
    g.xcd_block = g_xcd_block;
    if (epi == RTX_EPI_BIAS_ROWS) { g.lse_part = part; g.lse_ld = N / 64; }
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rtx_gemm_dma_launch(g, epi, 0);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) rtx_gemm_dma_launch(g, epi, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / it;
    printf("[perf dma %s] %s cfg%d M=%d N=%d K=%d splits=%d epi=%d: %.1f us  %.1f TFLOP/s\n", name, form == RTX_FORM_NT ? "NT" : "NN", cfg, M, N, K, splits, epi,
           us, 2.0 * M * N * K / us * 1e-6);
    hipFree(A); hipFree(B); hipFree(C); hipFree(bias); hipFree(part);
}

// The training step's logits as IEEE half (RtxGemm::C16, round 4): the same launch with the float32 and with the half output must
// agree bit for bit after rounding -- half(clamp(C)) -- and leave identical log-sum-exp partials.
static int run_logits16_case(int M, int N, int K, int M_real, int N_real, int shape, float scale)
{
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& x : hA) x = f32_to_bf16(frand() * scale);
    for (auto& x : hB) x = f32_to_bf16(frand());
    std::vector<float> hbias(N);
    for (auto& x : hbias) x = frand();
    bf16_t *A, *B;
    float *C, *bias;
    _Float16* C16;
    float2 *p0, *p1;
    const int strips = N / 64;
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)M * N * 4)); CK(hipMalloc(&C16, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&p0, (size_t)M * strips * 8)); CK(hipMalloc(&p1, (size_t)M * strips * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(bias, hbias.data(), N * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0, (size_t)M * N * 4)); CK(hipMemset(C16, 0xff, (size_t)M * N * 2));
    CK(hipMemset(p0, 0, (size_t)M * strips * 8)); CK(hipMemset(p1, 0xff, (size_t)M * strips * 8));
    RtxGemm g = {};
    g.A = A; g.B = B; g.lda = K; g.ldb = K;
    int bm, bn;
    rtx_gemm_tile_dims(shape, &bm, &bn);
    g.tile_shape = shape; g.m_tiles = M / bm; g.n_tiles = N / bn; g.k_slices = K * 2 / 128; g.splits = 1;
    g.C = C; g.ldc = N; g.bias = bias; g.M_real = M_real; g.N_real = N_real; g.lse_part = p0; g.lse_ld = strips;
    int rc = rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_BIAS_ROWS, 0);
    g.C16 = C16; g.ldc16 = N; g.lse_part = p1;
    rc |= rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_BIAS_ROWS, 0);
    if (rc) { printf("[logits16] launch failed: %s\n", rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hC((size_t)M * N);
    std::vector<_Float16> h16((size_t)M * N);
    std::vector<float2> q0((size_t)M * strips), q1((size_t)M * strips);
    CK(hipMemcpy(hC.data(), C, hC.size() * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(h16.data(), C16, h16.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(q0.data(), p0, q0.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(q1.data(), p1, q1.size() * 8, hipMemcpyDeviceToHost));
    long bad = 0, clamped = 0;
    for (int m = 0; m < M_real; ++m) {
        for (int n = 0; n < N_real; ++n) {
            float c = hC[(size_t)m * N + n];
            if (fabsf(c) > 65504.f) { c = c > 0 ? 65504.f : -65504.f; ++clamped; }
            const _Float16 want = (_Float16)c, got = h16[(size_t)m * N + n];
            if (memcmp(&want, &got, 2)) { if (bad++ < 5) printf("   logits16 (%d,%d): f32 %.6f half %.6f\n", m, n, c, (float)got); }
        }
        for (int k = 0; k < (N_real + 63) / 64; ++k)
            if (memcmp(&q0[(size_t)m * strips + k], &q1[(size_t)m * strips + k], 8)) { if (bad++ < 5) printf("   lse partial (%d,%d) differs\n", m, k); }
    }
    printf("[logits16] tile%d M=%d N=%d K=%d (%d x %d real) scale %.0f: %ld clamped, bad=%ld -> %s\n", shape, M, N, K, M_real, N_real, scale, clamped, bad, bad ? "FAIL" : "ok");
    hipFree(A); hipFree(B); hipFree(C); hipFree(C16); hipFree(bias); hipFree(p0); hipFree(p1);
    return bad != 0;
}

// the logits product as the training step launches it: half-precision logits + log-sum-exp strip partials
static void perf_logits16(int M, int N, int K, int shape)
{
    bf16_t *A, *B;
    float* bias;
    _Float16* C16;
    float2* part;
    std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
    for (auto& v : hA) v = f32_to_bf16(frand());
    for (auto& v : hB) v = f32_to_bf16(frand() * 0.1f);
    CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C16, (size_t)M * N * 2));
    CK(hipMalloc(&bias, N * 4)); CK(hipMalloc(&part, (size_t)M * (N / 64) * 8));
    CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemset(bias, 0, N * 4));
    RtxGemm g = {};
    g.A = A; g.B = B; g.lda = K; g.ldb = K;
    int bm, bn;
    rtx_gemm_tile_dims(shape, &bm, &bn);
    g.tile_shape = shape; g.m_tiles = M / bm; g.n_tiles = N / bn; g.k_slices = K * 2 / 128; g.splits = 1;
    g.ldc = N; g.bias = bias; g.M_real = M - 12; g.N_real = N - 116; g.lse_part = part; g.lse_ld = N / 64; g.C16 = C16; g.ldc16 = N;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_BIAS_ROWS, 0);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_BIAS_ROWS, 0);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / it;
    printf("[perf logits16+lse] bf16 M=%d N=%d K=%d tile%d: %.1f us  %.1f TFLOP/s\n", M, N, K, shape, us, 2.0 * M * N * K / us * 1e-6);
    hipFree(A); hipFree(B); hipFree(C16); hipFree(bias); hipFree(part);
}

// ---- weight gradient in TN form, fused with Adam (dw_adam.hip) -------------------------------------------------------------------
static int run_dw_case(const char* name, int cfg, int epi, int M_real, int N_real, int K_real, float lam, float wd, int keep)
{
    const int Mp = rtx_pad(M_real), Np = rtx_pad(N_real), Kp = rtx_pad_batch(K_real);
    std::vector<bf16_t> hD((size_t)Kp * Mp, 0), hX((size_t)Kp * Np, 0);
    for (int k = 0; k < K_real; ++k) {
        for (int m = 0; m < Mp; ++m) hD[(size_t)k * Mp + m] = f32_to_bf16(m < M_real ? frand() * 0.05f : frand());   // pad columns: garbage (must not leak)
        for (int n = 0; n < N_real; ++n) hX[(size_t)k * Np + n] = f32_to_bf16(frand());
        hX[(size_t)k * Np + N_real] = f32_to_bf16(1.f);
    }
    const size_t P = (size_t)M_real * N_real;
    std::vector<float> hp(P), hm(P), hv(P);
    for (size_t i = 0; i < P; ++i) { hp[i] = frand(); hm[i] = frand() * 0.01f; hv[i] = fabsf(frand()) * 1e-4f; }
    bf16_t *D, *X, *sh, *g16;
    float *p, *m, *v, *gk, *gb, *sumsq;
    CK(hipMalloc(&D, hD.size() * 2)); CK(hipMalloc(&X, hX.size() * 2));
    CK(hipMalloc(&p, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4)); CK(hipMalloc(&gk, P * 4)); CK(hipMalloc(&g16, P * 2));
    CK(hipMalloc(&gb, Mp * 4)); CK(hipMalloc(&sh, (size_t)Mp * Np * 2)); CK(hipMalloc(&sumsq, 4));
    CK(hipMemcpy(D, hD.data(), hD.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(p, hp.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(m, hm.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(v, hv.data(), P * 4, hipMemcpyHostToDevice));
    CK(hipMemset(gk, 0xff, P * 4)); CK(hipMemset(g16, 0xff, P * 2)); CK(hipMemset(gb, 0xff, Mp * 4)); CK(hipMemset(sh, 0, (size_t)Mp * Np * 2));
    double ss = 0;
    for (float w : hp) ss += (double)w * w;
    const float hss = (float)ss;
    CK(hipMemcpy(sumsq, &hss, 4, hipMemcpyHostToDevice));
    RtxDw d = {};
    d.A = D; d.lda = Mp; d.B = X; d.ldb = Np;
    d.m_tiles = Mp / rtx_dw_tile_rows(cfg); d.n_tiles = (Np + rtx_dw_tile_cols(cfg) - 1) / rtx_dw_tile_cols(cfg); d.k_slices = Kp / 64;
    d.M_real = M_real; d.N_real = N_real; d.gbias = gb;
    const float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    const int step = 3;
    const float step_size = (float)(lr / (1.0 - pow((double)b1, step))), bc2 = (float)sqrt(1.0 - pow((double)b2, step));
    std::vector<float> hbp(Mp), hbm(Mp), hbv(Mp);
    for (int i = 0; i < Mp; ++i) { hbp[i] = frand(); hbm[i] = frand() * 0.01f; hbv[i] = fabsf(frand()) * 1e-4f; }
    float *bp, *bm, *bv, *bss;
    CK(hipMalloc(&bp, Mp * 4)); CK(hipMalloc(&bm, Mp * 4)); CK(hipMalloc(&bv, Mp * 4)); CK(hipMalloc(&bss, 4));
    CK(hipMemcpy(bp, hbp.data(), Mp * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(bm, hbm.data(), Mp * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(bv, hbv.data(), Mp * 4, hipMemcpyHostToDevice));
    double bssd = 0;
    for (int i = 0; i < M_real; ++i) bssd += (double)hbp[i] * hbp[i];
    const float hbss = (float)bssd;
    CK(hipMemcpy(bss, &hbss, 4, hipMemcpyHostToDevice));
    if (epi == RTX_DW_ADAM) {
        d.bias_p = bp; d.bias_m = bm; d.bias_v = bv; d.bias_sumsq = lam != 0.f ? bss : nullptr;
        d.adam.p = p; d.adam.m = m; d.adam.v = v; d.adam.gkeep = keep ? gk : nullptr; d.adam.sh = sh; d.adam.ld_sh = Np;
        d.adam.step_size = step_size; d.adam.bc2_sqrt = bc2; d.adam.beta1 = b1; d.adam.beta2 = b2; d.adam.eps = eps; d.adam.weight_decay = wd;
        d.adam.lam = lam; d.adam.sumsq = lam != 0.f ? sumsq : nullptr;
    } else {
        d.gW = gk; d.g16 = keep ? g16 : nullptr;
    }
    int rc = rtx_dw_launch(d, epi, cfg, 0);
    if (rc) { printf("[dw %s] launch failed rc=%d: %s\n", name, rc, rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> gp(P), gm(P), gv(P), gg(P), ggb(Mp);
    std::vector<bf16_t> gsh((size_t)Mp * Np), gg16(P);
    CK(hipMemcpy(gp.data(), p, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gm.data(), m, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), v, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gg.data(), gk, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ggb.data(), gb, Mp * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gsh.data(), sh, gsh.size() * 2, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gg16.data(), g16, P * 2, hipMemcpyDeviceToHost));
    std::vector<float> gbp(Mp), gbm(Mp), gbv(Mp);
    CK(hipMemcpy(gbp.data(), bp, Mp * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gbm.data(), bm, Mp * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gbv.data(), bv, Mp * 4, hipMemcpyDeviceToHost));
    long bad = 0;
    int printed = 0;
    double e_g = 0, e_p = 0, e_m = 0, e_v = 0;
    const float reg = lam != 0.f ? lam / sqrtf(hss) : 0.f;
    for (int r = 0; r < M_real; ++r) {
        for (int c = 0; c <= N_real; ++c) {
            double ref = 0;
            for (int k = 0; k < K_real; ++k) ref += (double)bf16_to_f32(hD[(size_t)k * Mp + r]) * bf16_to_f32(hX[(size_t)k * Np + c]);
            if (c == N_real) {
                const double e = fabs(ggb[r] - ref);
                if (!(e <= 1e-4 * sqrt((double)K_real))) { ++bad; if (printed++ < 5) printf("   bias grad row %d: got %.6f ref %.6f\n", r, ggb[r], ref); }
                if (epi == RTX_DW_ADAM) {   // the bias took its Adam step in the same launch
                    const float breg = lam != 0.f ? lam / sqrtf(hbss) : 0.f;
                    float g1 = ggb[r] + breg * hbp[r];
                    if (wd != 0.f) g1 += wd * hbp[r];
                    const float m1 = hbm[r] + (g1 - hbm[r]) * (1.f - b1);
                    const float v1 = hbv[r] * b2 + (1.f - b2) * g1 * g1;
                    const float p1 = hbp[r] - step_size * (m1 / (sqrtf(v1) / bc2 + eps));
                    if (!(fabs(gbp[r] - p1) <= 2e-6 && fabs(gbm[r] - m1) <= 2e-6 && fabs(gbv[r] - v1) <= 2e-6)) {
                        ++bad;
                        if (printed++ < 5) printf("   bias adam row %d: p %.7f/%.7f m %.7f/%.7f\n", r, gbp[r], p1, gbm[r], m1);
                    }
                }
                continue;
            }
            const size_t o = (size_t)r * N_real + c;
            const bool have_g = (epi == RTX_DW_GRAD) || keep;
            if (have_g) {
                const double e = fabs(gg[o] - ref);
                e_g = std::max(e_g, e);
                if (!(e <= 1e-4 * sqrt((double)K_real))) { ++bad; if (printed++ < 5) printf("   grad (%d,%d): got %.6f ref %.6f\n", r, c, gg[o], ref); }
            }
            if (epi == RTX_DW_GRAD) {
                if (keep && gg16[o] != f32_to_bf16(gg[o])) { ++bad; if (printed++ < 5) printf("   bf16 grad image (%d,%d) differs\n", r, c); }
                continue;
            }
            // Adam with the DEVICE's f32 gradient when it is available (isolates the optimizer arithmetic), else the reference
            float g0 = have_g ? gg[o] : (float)ref;
            float g1 = g0 + reg * hp[o];
            if (wd != 0.f) g1 += wd * hp[o];
            const float m1 = hm[o] + (g1 - hm[o]) * (1.f - b1);
            const float v1 = hv[o] * b2 + (1.f - b2) * g1 * g1;
            const float denom = sqrtf(v1) / bc2 + eps;
            const float p1 = hp[o] - step_size * (m1 / denom);
            const double tol = have_g ? 2e-6 : 2e-3;
            e_p = std::max(e_p, (double)fabs(gp[o] - p1)); e_m = std::max(e_m, (double)fabs(gm[o] - m1)); e_v = std::max(e_v, (double)fabs(gv[o] - v1));
            if (!(fabs(gp[o] - p1) <= tol && fabs(gm[o] - m1) <= tol && fabs(gv[o] - v1) <= tol)) {
                ++bad;
                if (printed++ < 5) printf("   adam (%d,%d): p %.7f/%.7f m %.7f/%.7f v %.3e/%.3e\n", r, c, gp[o], p1, gm[o], m1, gv[o], v1);
            }
            if (gsh[(size_t)r * Np + c] != f32_to_bf16(gp[o])) { ++bad; if (printed++ < 5) printf("   compute copy (%d,%d) is not bf16(p)\n", r, c); }
        }
    }
    if (epi == RTX_DW_ADAM)   // the padding of the compute copy stays zero
        for (int r = 0; r < Mp; ++r)
            for (int c = 0; c < Np; ++c)
                if ((r >= M_real || c >= N_real) && gsh[(size_t)r * Np + c] != 0) { ++bad; if (printed++ < 5) printf("   compute-copy pad (%d,%d) written\n", r, c); }
    printf("[dw %s] cfg%d epi=%d %dx%d K=%d lam=%.2f wd=%.3f keep=%d  err g=%.2e p=%.2e m=%.2e v=%.2e bad=%ld -> %s\n", name, cfg, epi, M_real, N_real, K_real,
           lam, wd, keep, e_g, e_p, e_m, e_v, bad, bad ? "FAIL" : "ok");
    hipFree(D); hipFree(X); hipFree(p); hipFree(m); hipFree(v); hipFree(gk); hipFree(g16); hipFree(gb); hipFree(sh); hipFree(sumsq);
    hipFree(bp); hipFree(bm); hipFree(bv); hipFree(bss);
    return bad != 0;
}

// `sets` > 1: that many separate p / m / v / compute-copy buffer sets used round-robin, so that a launch never finds its
// optimizer state in the 256-MiB Infinity Cache left there by the launch before (what happens inside a training step)
static void perf_dw(const char* name, int cfg, int epi, int M_real, int N_real, int K_real, int sets = 1)
{
    const int Mp = rtx_pad(M_real), Np = rtx_pad(N_real), Kp = rtx_pad_batch(K_real);
    std::vector<bf16_t> hD((size_t)Kp * Mp), hX((size_t)Kp * Np);
    for (auto& x : hD) x = f32_to_bf16(frand() * 0.01f);
    for (auto& x : hX) x = f32_to_bf16(frand());
    const size_t P = (size_t)M_real * N_real;
    bf16_t *D, *X;
    float *gb, *gW;
    std::vector<float*> p(sets), m(sets), v(sets);
    std::vector<bf16_t*> sh(sets);
    CK(hipMalloc(&D, hD.size() * 2)); CK(hipMalloc(&X, hX.size() * 2));
    for (int i = 0; i < sets; ++i) {
        CK(hipMalloc(&p[i], P * 4)); CK(hipMalloc(&m[i], P * 4)); CK(hipMalloc(&v[i], P * 4)); CK(hipMalloc(&sh[i], (size_t)Mp * Np * 2));
        CK(hipMemset(p[i], 0, P * 4)); CK(hipMemset(m[i], 0, P * 4)); CK(hipMemset(v[i], 0, P * 4));
    }
    CK(hipMalloc(&gW, P * 4));
    CK(hipMalloc(&gb, Mp * 4));
    CK(hipMemcpy(D, hD.data(), hD.size() * 2, hipMemcpyHostToDevice));
    CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
    RtxDw d = {};
    d.A = D; d.lda = Mp; d.B = X; d.ldb = Np;
    d.m_tiles = Mp / rtx_dw_tile_rows(cfg); d.n_tiles = (Np + rtx_dw_tile_cols(cfg) - 1) / rtx_dw_tile_cols(cfg); d.k_slices = Kp / 64;
    d.M_real = M_real; d.N_real = N_real; d.gbias = gb;
    d.adam.ld_sh = Np;
    d.adam.step_size = 1e-3f; d.adam.bc2_sqrt = 0.05f; d.adam.beta1 = 0.9f; d.adam.beta2 = 0.999f; d.adam.eps = 1e-8f;
    d.gW = gW;
    auto go = [&](int i) {
        d.adam.p = p[i % sets]; d.adam.m = m[i % sets]; d.adam.v = v[i % sets]; d.adam.sh = sh[i % sets];
        rtx_dw_launch(d, epi, cfg, 0);
    };
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < 3; ++i) go(i);
    CK(hipEventRecord(e0, 0));
    const int it = 20;
    for (int i = 0; i < it; ++i) go(i);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / it;
    const double bytes = (epi == RTX_DW_ADAM ? 26.0 : 4.0) * P + 2.0 * Kp * (Mp + Np);
    printf("[perf dw %s%s] cfg%d epi=%d %dx%d K=%d: %.1f us  %.2f TB/s (p,m,v r+w + compute copy + operands)  %.1f TFLOP/s\n", name,
           sets > 1 ? " cold" : "", cfg, epi, M_real, N_real, K_real, us, bytes / us * 1e-6, 2.0 * P * Kp / us * 1e-6);
    hipFree(D); hipFree(X); hipFree(gW); hipFree(gb);
    for (int i = 0; i < sets; ++i) { hipFree(p[i]); hipFree(m[i]); hipFree(v[i]); hipFree(sh[i]); }
}

// Several matrices in one launch (rtx_dw_launch_group) must produce, bit for bit, what one launch per matrix produces.
static int run_dw_group_case(int cfg, int odd = 0)   // odd: one matrix with rows of N % 4 != 0 floats (the whole launch takes the strided epilogue)
{
    const int K_real = 250, Kp = rtx_pad_batch(K_real);
    const int shapes[3][2] = {{70, 132}, {300, odd ? 201 : 200}, {130, 600}};
    struct Buf { bf16_t *D, *X, *sh[2]; float *p[2], *m[2], *v[2], *bp[2], *bm[2], *bv[2]; int Mp, Np; size_t P; };
    Buf b[3];
    RtxDw d[2][3];
    for (int k = 0; k < 3; ++k) {
        const int M = shapes[k][0], N = shapes[k][1];
        b[k].Mp = rtx_pad(M); b[k].Np = rtx_pad(N); b[k].P = (size_t)M * N;
        std::vector<bf16_t> hD((size_t)Kp * b[k].Mp, 0), hX((size_t)Kp * b[k].Np, 0);
        for (int r = 0; r < K_real; ++r) {
            for (int c = 0; c < b[k].Mp; ++c) hD[(size_t)r * b[k].Mp + c] = f32_to_bf16(frand() * 0.05f);
            for (int c = 0; c < N; ++c) hX[(size_t)r * b[k].Np + c] = f32_to_bf16(frand());
            hX[(size_t)r * b[k].Np + N] = f32_to_bf16(1.f);
        }
        std::vector<float> hp(b[k].P), hm(b[k].P), hv(b[k].P), hb(b[k].Mp);
        for (size_t i = 0; i < b[k].P; ++i) { hp[i] = frand(); hm[i] = frand() * 0.01f; hv[i] = fabsf(frand()) * 1e-4f; }
        for (auto& x : hb) x = frand() * 0.1f;
        CK(hipMalloc(&b[k].D, hD.size() * 2)); CK(hipMalloc(&b[k].X, hX.size() * 2));
        CK(hipMemcpy(b[k].D, hD.data(), hD.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(b[k].X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
        for (int s = 0; s < 2; ++s) {
            CK(hipMalloc(&b[k].p[s], b[k].P * 4)); CK(hipMalloc(&b[k].m[s], b[k].P * 4)); CK(hipMalloc(&b[k].v[s], b[k].P * 4));
            CK(hipMalloc(&b[k].bp[s], b[k].Mp * 4)); CK(hipMalloc(&b[k].bm[s], b[k].Mp * 4)); CK(hipMalloc(&b[k].bv[s], b[k].Mp * 4));
            CK(hipMalloc(&b[k].sh[s], (size_t)b[k].Mp * b[k].Np * 2)); CK(hipMemset(b[k].sh[s], 0, (size_t)b[k].Mp * b[k].Np * 2));
            CK(hipMemcpy(b[k].p[s], hp.data(), b[k].P * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(b[k].m[s], hm.data(), b[k].P * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(b[k].v[s], hv.data(), b[k].P * 4, hipMemcpyHostToDevice));
            CK(hipMemcpy(b[k].bp[s], hb.data(), b[k].Mp * 4, hipMemcpyHostToDevice)); CK(hipMemset(b[k].bm[s], 0, b[k].Mp * 4)); CK(hipMemset(b[k].bv[s], 0, b[k].Mp * 4));
            RtxDw& q = d[s][k];
            q = RtxDw{};
            q.A = b[k].D; q.lda = b[k].Mp; q.B = b[k].X; q.ldb = b[k].Np;
            q.m_tiles = b[k].Mp / rtx_dw_tile_rows(cfg); q.n_tiles = (b[k].Np + rtx_dw_tile_cols(cfg) - 1) / rtx_dw_tile_cols(cfg); q.k_slices = Kp / 64;
            q.M_real = M; q.N_real = N;
            q.adam.p = b[k].p[s]; q.adam.m = b[k].m[s]; q.adam.v = b[k].v[s]; q.adam.sh = b[k].sh[s]; q.adam.ld_sh = b[k].Np;
            q.adam.step_size = 1e-3f * (k + 1); q.adam.bc2_sqrt = 0.05f; q.adam.beta1 = 0.9f; q.adam.beta2 = 0.999f; q.adam.eps = 1e-8f;
            q.bias_p = b[k].bp[s]; q.bias_m = b[k].bm[s]; q.bias_v = b[k].bv[s];
        }
    }
    int rc = rtx_dw_launch_group(d[0], 3, RTX_DW_ADAM, cfg, 0);
    for (int k = 0; k < 3 && !rc; ++k) rc = rtx_dw_launch(d[1][k], RTX_DW_ADAM, cfg, 0);
    if (rc) { printf("[dw group cfg%d] launch failed rc=%d: %s\n", cfg, rc, rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    long diff = 0;
    for (int k = 0; k < 3; ++k) {
        auto cmp = [&](const void* x, const void* y, size_t bytes) {
            std::vector<unsigned char> hx(bytes), hy(bytes);
            CK(hipMemcpy(hx.data(), x, bytes, hipMemcpyDeviceToHost)); CK(hipMemcpy(hy.data(), y, bytes, hipMemcpyDeviceToHost));
            diff += memcmp(hx.data(), hy.data(), bytes) != 0;
        };
        cmp(b[k].p[0], b[k].p[1], b[k].P * 4); cmp(b[k].m[0], b[k].m[1], b[k].P * 4); cmp(b[k].v[0], b[k].v[1], b[k].P * 4);
        cmp(b[k].bp[0], b[k].bp[1], b[k].Mp * 4); cmp(b[k].bm[0], b[k].bm[1], b[k].Mp * 4); cmp(b[k].bv[0], b[k].bv[1], b[k].Mp * 4);
        cmp(b[k].sh[0], b[k].sh[1], (size_t)b[k].Mp * b[k].Np * 2);
        hipFree(b[k].D); hipFree(b[k].X);
        for (int s = 0; s < 2; ++s) { hipFree(b[k].p[s]); hipFree(b[k].m[s]); hipFree(b[k].v[s]); hipFree(b[k].bp[s]); hipFree(b[k].bm[s]); hipFree(b[k].bv[s]); hipFree(b[k].sh[s]); }
    }
    printf("[dw group%s] cfg%d 3 matrices in one launch vs one launch each: %ld differing buffers -> %s\n", odd ? " odd" : "", cfg, diff, diff ? "FAIL" : "ok");
    return diff ? 1 : 0;
}

// ---- float32 GEMM with a K-major operand (gemm_f32.hip) ------------------------------------------------------------------------------
static int run_f32_case(const char* name, int form, int M, int N, int K, int splits, int epi, int M_real, int N_real)
{
    std::vector<float> hA((size_t)M * K), hB((size_t)N * K);
    std::vector<double> dA((size_t)M * K), dB((size_t)N * K);   // d*[row][k]
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < K; ++k) {
            const float v = frand();
            dA[(size_t)m * K + k] = v;
            if (form == RTX_FORM_TN) hA[(size_t)k * M + m] = v; else hA[(size_t)m * K + k] = v;
        }
    for (int n = 0; n < N; ++n)
        for (int k = 0; k < K; ++k) {
            const float v = frand() * 0.5f + 0.1f;
            dB[(size_t)n * K + k] = v;
            hB[(size_t)k * N + n] = v;
        }
    const size_t csz = (epi == RTX_EPI_STORE) ? (size_t)splits * M * N : (size_t)M_real * N_real;
    float *A, *B, *C, *gb;
    CK(hipMalloc(&A, hA.size() * 4)); CK(hipMalloc(&B, hB.size() * 4)); CK(hipMalloc(&C, csz * 4)); CK(hipMalloc(&gb, M * 4));
    CK(hipMemcpy(A, hA.data(), hA.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(B, hB.data(), hB.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemset(C, 0xff, csz * 4)); CK(hipMemset(gb, 0xff, M * 4));
    RtxGemm g = {};
    g.form = form; g.A = A; g.B = B; g.lda = (form == RTX_FORM_TN) ? M : K; g.ldb = N;
    g.m_tiles = M / 128; g.n_tiles = N / 128; g.k_slices = K / 32; g.splits = splits;
    g.C = C; g.ldc = N; g.slab_stride = (long)M * N; g.gbias = gb; g.M_real = M_real; g.N_real = N_real;
    int rc = rtx_gemm_f32_km_launch(g, epi, 0);
    if (rc) { printf("[f32 %s] launch failed rc=%d: %s\n", name, rc, rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    std::vector<float> hC(csz), hgb(M);
    CK(hipMemcpy(hC.data(), C, csz * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(hgb.data(), gb, M * 4, hipMemcpyDeviceToHost));
    double max_err = 0;
    long bad = 0;
    int printed = 0;
    const int Mc = (epi == RTX_EPI_STORE) ? M : M_real, Nc = (epi == RTX_EPI_STORE) ? N : N_real + 1;
    for (int m = 0; m < Mc; ++m)
        for (int n = 0; n < Nc; ++n) {
            double ref = 0;
            for (int k = 0; k < K; ++k) ref += dA[(size_t)m * K + k] * dB[(size_t)n * K + k];
            double got;
            if (epi == RTX_EPI_STORE) {
                got = 0;
                for (int s2 = 0; s2 < splits; ++s2) got += hC[(size_t)s2 * M * N + (size_t)m * N + n];
            } else {
                got = (n < N_real) ? hC[(size_t)m * N_real + n] : hgb[m];
            }
            const double err = fabs(got - ref);
            if (!(err <= 2e-6 * sqrt((double)K) * 4)) { ++bad; if (printed++ < 5) printf("   mismatch (%d,%d): got %.7f ref %.7f\n", m, n, got, ref); }
            max_err = std::max(max_err, err);
        }
    printf("[f32 %s] %s M=%d N=%d K=%d splits=%d epi=%d  max_err=%.3e bad=%ld -> %s\n", name, form == RTX_FORM_TN ? "TN" : "NN", M, N, K, splits, epi, max_err, bad,
           bad ? "FAIL" : "ok");
    hipFree(A); hipFree(B); hipFree(C); hipFree(gb);
    return bad != 0;
}

static int run_dw_cases()
{
    int fails = 0;
    for (int cfg = 0; cfg < 9; ++cfg) {   // (5, 6, 7: 32x256 on two / three stages, on four stages of 32-row slices; 8: 128x128 on four stages of 32-row slices)   64x128 / 32x128 (3 stages) / 32x128 (2 stages) / 128x128 (2 stages, 32x64 per wave) / 128x128 on four waves
        fails += run_dw_case("adam", cfg, RTX_DW_ADAM, 300, 200, 250, 0.f, 0.f, 1);
        fails += run_dw_case("adam-nokeep", cfg, RTX_DW_ADAM, 130, 600, 500, 0.f, 0.f, 0);
        fails += run_dw_case("adam-dae", cfg, RTX_DW_ADAM, 70, 132, 100, 0.2f, 0.001f, 1);
        fails += run_dw_case("adam-tall", cfg, RTX_DW_ADAM, 1000, 24, 128, 0.f, 0.f, 1);
        fails += run_dw_case("adam-oddcols", cfg, RTX_DW_ADAM, 130, 301, 190, 0.f, 0.f, 1);        // rows of N % 4 != 0 floats: the strided epilogue
        fails += run_dw_case("adam-odd-dae", cfg, RTX_DW_ADAM, 70, 133, 100, 0.2f, 0.001f, 1);
        fails += run_dw_case("adam-odd-tall", cfg, RTX_DW_ADAM, 1000, 27, 128, 0.f, 0.f, 0);
        fails += run_dw_case("adam-many-tiles", cfg, RTX_DW_ADAM, 5000, 1100, 100, 0.f, 0.f, 1);       // several workgroups per slot
        fails += run_dw_case("adam-many-tiles-odd", cfg, RTX_DW_ADAM, 1101, 5003, 70, 0.1f, 0.001f, 0);
        fails += run_dw_case("grad", cfg, RTX_DW_GRAD, 300, 200, 250, 0.f, 0.f, 1);
        fails += run_dw_case("grad-oddcols", cfg, RTX_DW_GRAD, 77, 301, 190, 0.f, 0.f, 1);
        fails += run_dw_case("grad-tiny", cfg, RTX_DW_GRAD, 2, 1, 3, 0.f, 0.f, 0);
        fails += run_dw_group_case(cfg);
        fails += run_dw_group_case(cfg, 1);
    }
    return fails;
}

// ---- what the optimizer state's ACCESS PATTERN costs, without any of the kernel around it ---------------------
// The fused weight-gradient + Adam kernel reads and rewrites p / exp_avg / exp_avg_sq (3 float32 arrays in place) and writes the bf16
// compute copy: six address streams + one, as 64 x 128 tiles of a row-major [M][N] tensor (a tile row = 512 contiguous bytes; rows
// 2400 B or 80 432 B apart), 512 threads per workgroup, one float4 per thread and 16-row pass, every load of a thread issued before
// its first store -- dw_adam.hip's epilogue exactly.  This kernel does ONLY that, for NA = 1, 2, 3 arrays (2, 4, 6 streams), with and
// without the bf16 copy, non-temporal or default cache policy, and -- the comparison -- the same bytes walked as FLAT contiguous
// 32-KB chunks per workgroup (what a plain multi-tensor Adam does).  The float4 copy figure of the hardware guide (6.29 TB/s) is the
// NA = 1 flat line.
__device__ int g_tile_group = 0;
template <int NA, bool SH, bool NT, bool FLAT, int TN = 128, bool XM = true>
__global__ __launch_bounds__(512, 2) void k_state_stream(float* a0, float* a1, float* a2, bf16_t* sh, int M, int N, int ld_sh, int m_tiles, int n_tiles)
{
    typedef __attribute__((ext_vector_type(4))) float f4;
    typedef f4 f4u __attribute__((aligned(4)));
    float* arr[3] = {a0, a1, a2};
    const int tid = threadIdx.x;
    f4 v[NA][4];
    size_t off[4];
    bool on[4];
    if (FLAT) {
        const size_t total = (size_t)M * N;
        const unsigned nb = gridDim.x, per = (nb + 7) / 8;
        // XM (flat): every XCD walks ONE contiguous eighth of the array (the tile kernels' order) instead of chunks dealt round-robin
        const size_t chunk = XM ? (size_t)(blockIdx.x & 7) * per + (blockIdx.x >> 3) : (size_t)blockIdx.x;
        const size_t base = chunk * 8192;     // 8192 floats = 32 KB per workgroup and array
#pragma unroll
        for (int q = 0; q < 4; ++q) { off[q] = base + (size_t)(q * 512 + tid) * 4; on[q] = off[q] + 4 <= total; }
    } else {
        const int n_cover = (g_tile_group > 1 && n_tiles > m_tiles) ? (n_tiles + g_tile_group - 1) / g_tile_group * g_tile_group : n_tiles;
        const int total = m_tiles * n_cover, per_xcd = (total + 7) / 8;
        const int id = XM ? (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3) : (int)blockIdx.x;
        if (id >= total) return;
        int tm, tn;
        if (n_tiles <= m_tiles) { tm = id / n_tiles; tn = id % n_tiles; } else { tn = id / m_tiles; tm = id % m_tiles; }
        if (g_tile_group > 1 && n_tiles > m_tiles) {
            // STATE_STREAM_GROUP=G: G neighbouring column tiles of one row tile are dispatched back to back (they touch G x 512 contiguous bytes of
            // every row at about the same time), then the next row tile of the same G strips ...
            const int G = g_tile_group, per_grp = G * m_tiles, grp = id / per_grp, in = id % per_grp;
            tn = grp * G + in % G; tm = in / G;
            if (tn >= n_tiles) return;
        }
        constexpr int TPR = TN / 4, RPP = 512 / TPR, TM = 8192 / TN;      // threads per tile row, rows per pass, tile rows
        const int rowl = tid / TPR, col = tn * TN + (tid % TPR) * 4;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int row = tm * TM + q * RPP + rowl;
            on[q] = row < M && col + 4 <= N;
            off[q] = (size_t)min(row, M - 1) * N + min(col, N - 4);
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int k = 0; k < NA; ++k)
            v[k][q] = NT ? __builtin_nontemporal_load((const f4u*)(arr[k] + off[q])) : *(const f4u*)(arr[k] + off[q]);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        if (!on[q]) continue;
        f4 acc = v[0][q];
#pragma unroll
        for (int k = 0; k < NA; ++k) {
            const f4 w = v[k][q] * 1.0001f + 1e-9f;
            acc += w;
            if (NT) __builtin_nontemporal_store(w, (f4u*)(arr[k] + off[q])); else *(f4u*)(arr[k] + off[q]) = w;
        }
        if (SH) {
            const size_t r = off[q] / N, c = off[q] % N;      // (tile mapping: the element's own row / column; flat: the same arithmetic the flat Adam does)
            store4<bf16_t>(sh + r * ld_sh + (c & ~(size_t)3), acc[0], acc[1], acc[2], acc[3]);
        }
    }
}

template <int NA, bool SH, bool NT, bool FLAT, int TN = 128, bool XM = true>
static double state_stream_once(float** arr, bf16_t* sh, int M, int N, int set, size_t set_stride)
{
    const int Np = rtx_pad(N), m_tiles = (M + 8192 / TN - 1) / (8192 / TN), n_tiles = (N + TN - 1) / TN;
    static const int grp = getenv("STATE_STREAM_GROUP") ? atoi(getenv("STATE_STREAM_GROUP")) : 0;
    { static bool once = false; if (!once) { once = true; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tile_group), &grp, sizeof(int)); } }
    const int n_cover = grp > 1 ? (n_tiles + grp - 1) / grp * grp : n_tiles;
    const unsigned grid = FLAT ? (unsigned)(((size_t)M * N + 8191) / 8192) : (unsigned)(8 * ((m_tiles * n_cover + 7) / 8));
    // STATE_STREAM_LDS=bytes: unused dynamic LDS per workgroup, to pin how many workgroups share a CU (73728: two, as the fused kernel; 0: what registers allow)
    static const int lds_pin = getenv("STATE_STREAM_LDS") ? atoi(getenv("STATE_STREAM_LDS")) : 0;
    if (lds_pin > 65536) { static bool once = false; if (!once) { once = true; (void)hipFuncSetAttribute((const void*)k_state_stream<NA, SH, NT, FLAT, TN, XM>, hipFuncAttributeMaxDynamicSharedMemorySize, lds_pin); } }
    hipLaunchKernelGGL((k_state_stream<NA, SH, NT, FLAT, TN, XM>), dim3(grid), dim3(512), lds_pin, 0, arr[0] + set * set_stride, arr[1] + set * set_stride,
                       arr[2] + set * set_stride, sh, M, N, Np, m_tiles, n_tiles);
    return 0;
}

template <int NA, bool SH, bool NT, bool FLAT, int TN = 128, bool XM = true>
static void state_stream_case(float** arr, bf16_t* sh, int M, int N, size_t set_stride, int sets)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int i = 0; i < sets; ++i) state_stream_once<NA, SH, NT, FLAT, TN, XM>(arr, sh, M, N, i, set_stride);
    CK(hipDeviceSynchronize());
    const int it = 12;
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < it; ++i) state_stream_once<NA, SH, NT, FLAT, TN, XM>(arr, sh, M, N, i % sets, set_stride);   // rotating sets: cold in the 256-MB cache
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / it, bytes = (double)M * N * (8.0 * NA + (SH ? 2.0 : 0.0));
    char shape[32];
    snprintf(shape, sizeof shape, "%dx%d tiles", 8192 / TN, TN);
    if (!XM) strcat(shape, FLAT ? "" : " rr");
    printf("[state-stream] %5dx%-5d %-17s %s: %d arrays read + rewritten in place (%d streams)%s: %6.1f us  %5.2f TB/s\n", M, N, FLAT ? (XM ? "flat, XCD-contig" : "flat 32-KB chunks") : shape,
           NT ? "nt   " : "plain", NA, 2 * NA, SH ? " + bf16 copy" : "            ", us, bytes / us * 1e-6);
}

static void run_state_stream()
{
    const int shapes[2][2] = {{20108, 600}, {600, 20108}};
    const size_t P = (size_t)20108 * 600, stride = (P + 1023) / 1024 * 1024;
    const int sets = 3;                                        // 3 x 3 x 48 MB: no set is still in the 256-MB cache when its turn comes again
    float* arr[3];
    bf16_t* sh;
    for (int k = 0; k < 3; ++k) { CK(hipMalloc(&arr[k], stride * sets * 4)); CK(hipMemset(arr[k], 0, stride * sets * 4)); }
    CK(hipMalloc(&sh, (size_t)rtx_pad(20108) * rtx_pad(20108 > 600 ? 600 : 600) * 2 + (size_t)rtx_pad(600) * rtx_pad(20108) * 2));
    for (int rep = 0; rep < 1; ++rep)
        for (auto& shp : shapes) {
            const int M = shp[0], N = shp[1];
            state_stream_case<1, false, true, true, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<1, false, false, true, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<2, false, true, true, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, false, true, true, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, true, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, true, 128, true>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false, 128, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false, 1024, false>(arr, sh, M, N, stride, sets);
            state_stream_case<1, false, true, false>(arr, sh, M, N, stride, sets);
            state_stream_case<2, false, true, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, false, true, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, false, false, false>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false>(arr, sh, M, N, stride, sets);
            // the same 8192 elements per workgroup as tiles with longer rows (what a workgroup owning several adjacent MFMA tiles would touch per epilogue pass)
            state_stream_case<3, true, true, false, 256>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false, 512>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false, 1024>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, true, false, 2048>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, false, false, 512>(arr, sh, M, N, stride, sets);
            state_stream_case<3, true, false, false, 2048>(arr, sh, M, N, stride, sets);
        }
    for (int k = 0; k < 3; ++k) hipFree(arr[k]);
    hipFree(sh);
}

// the fused epilogue's hardware sqrt / rcp (dw_adam.hip dw_finish4) isolated from operand rounding.  100 Adam
// steps of rtx_dw_tn<..., ADAM> on fresh bf16 operands each step; the gradient the epilogue used leaves through `gkeep`, and the
// EXACT update (k_adam's expressions: IEEE sqrtf and division, float32, torch.optim.Adam semantics -- reference models.py:768-770)
// is applied by a kernel of this file to a second copy of (p, m, v) from the SAME gradient bits.  The two copies never exchange
// values: the drift accumulates as it would in training.  Bounds: after 100 steps the parameters differ by at most 2e-6 of the
// distance an update of size lr per step can cover (measured: 2e-8 absolute = 5e-7 of the motion); the moments carry no approximation
// (last-bit differences from FMA contraction are bounded at rounding level).
__global__ void k_exact_adam(float* p, float* m, float* v, const float* g, size_t n, float step_size, float bc2_sqrt, float b1, float b2, float eps)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float gg = g[i];
    const float m1 = m[i] + (gg - m[i]) * (1.f - b1);
    const float v1 = v[i] * b2 + (1.f - b2) * gg * gg;
    const float denom = sqrtf(v1) / bc2_sqrt + eps;
    p[i] = p[i] - step_size * (m1 / denom);
    m[i] = m1;
    v[i] = v1;
}

static int run_adam_approx_isolation(int cfg)
{
    const int M_real = 130, N_real = 600, K_real = 500, STEPS = 100;
    const int Mp = rtx_pad(M_real), Np = rtx_pad(N_real), Kp = rtx_pad_batch(K_real);
    const size_t P = (size_t)M_real * N_real;
    std::vector<float> hp(P);
    for (auto& w : hp) w = frand() * 0.05f;
    bf16_t *D, *X, *sh;
    float *p, *m, *v, *gk, *gb, *xp, *xm, *xv;
    CK(hipMalloc(&D, (size_t)Kp * Mp * 2)); CK(hipMalloc(&X, (size_t)Kp * Np * 2));
    CK(hipMalloc(&p, P * 4)); CK(hipMalloc(&m, P * 4)); CK(hipMalloc(&v, P * 4)); CK(hipMalloc(&gk, P * 4)); CK(hipMalloc(&gb, Mp * 4));
    CK(hipMalloc(&xp, P * 4)); CK(hipMalloc(&xm, P * 4)); CK(hipMalloc(&xv, P * 4));
    CK(hipMalloc(&sh, (size_t)Mp * Np * 2));
    CK(hipMemcpy(p, hp.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemset(m, 0, P * 4)); CK(hipMemset(v, 0, P * 4));
    CK(hipMemcpy(xp, hp.data(), P * 4, hipMemcpyHostToDevice)); CK(hipMemset(xm, 0, P * 4)); CK(hipMemset(xv, 0, P * 4));
    const float lr = 1e-3f, b1 = 0.9f, b2 = 0.999f, eps = 1e-8f;
    std::vector<bf16_t> hD((size_t)Kp * Mp, 0), hX((size_t)Kp * Np, 0);
    std::vector<float> gp(P), gm(P), gv(P), ep(P), em(P), ev(P);
    double worst_growth = 0, prev = 0;
    for (int step = 1; step <= STEPS; ++step) {
        // gradients of realistic spread: most entries small (sqrt(v) near eps matters), every 17th row 100x larger
        for (int k = 0; k < K_real; ++k) {
            for (int mm = 0; mm < M_real; ++mm) hD[(size_t)k * Mp + mm] = f32_to_bf16(frand() * (mm % 17 == 0 ? 2e-2f : 2e-4f));
            for (int n = 0; n < N_real; ++n) hX[(size_t)k * Np + n] = f32_to_bf16(frand());
            hX[(size_t)k * Np + N_real] = f32_to_bf16(1.f);
        }
        CK(hipMemcpy(D, hD.data(), hD.size() * 2, hipMemcpyHostToDevice));
        CK(hipMemcpy(X, hX.data(), hX.size() * 2, hipMemcpyHostToDevice));
        RtxDw d = {};
        d.A = D; d.lda = Mp; d.B = X; d.ldb = Np;
        d.m_tiles = Mp / rtx_dw_tile_rows(cfg); d.n_tiles = (Np + rtx_dw_tile_cols(cfg) - 1) / rtx_dw_tile_cols(cfg); d.k_slices = Kp / 64;
        d.M_real = M_real; d.N_real = N_real; d.gbias = gb;
        const float step_size = (float)(lr / (1.0 - pow((double)b1, step))), bc2 = (float)sqrt(1.0 - pow((double)b2, step));
        d.adam.p = p; d.adam.m = m; d.adam.v = v; d.adam.gkeep = gk; d.adam.sh = sh; d.adam.ld_sh = Np;
        d.adam.step_size = step_size; d.adam.bc2_sqrt = bc2; d.adam.beta1 = b1; d.adam.beta2 = b2; d.adam.eps = eps;
        if (rtx_dw_launch(d, RTX_DW_ADAM, cfg, 0)) { printf("[adam-approx] launch failed: %s\n", rtx_last_error_str()); return 1; }
        hipLaunchKernelGGL(k_exact_adam, dim3((unsigned)((P + 255) / 256)), dim3(256), 0, 0, xp, xm, xv, gk, P, step_size, bc2, b1, b2, eps);
        CK(hipDeviceSynchronize());
        if (step % 10 == 0 || step == 1) {
            CK(hipMemcpy(gp.data(), p, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(ep.data(), xp, P * 4, hipMemcpyDeviceToHost));
            double dd = 0;
            for (size_t i = 0; i < P; ++i) dd = fmax(dd, fabs((double)gp[i] - ep[i]));
            worst_growth = fmax(worst_growth, (dd - prev) / (step == 1 ? 1 : 10));
            prev = dd;
        }
    }
    CK(hipMemcpy(gp.data(), p, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(gm.data(), m, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gv.data(), v, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ep.data(), xp, P * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(em.data(), xm, P * 4, hipMemcpyDeviceToHost));
    CK(hipMemcpy(ev.data(), xv, P * 4, hipMemcpyDeviceToHost));
    double drift = 0, moved = 0, dm = 0, dv = 0, mx_m = 0, mx_v = 0;
    long moment_bits = 0;
    for (size_t i = 0; i < P; ++i) {
        drift = fmax(drift, fabs((double)gp[i] - ep[i]));
        moved = fmax(moved, fabs((double)ep[i] - hp[i]));
        if (memcmp(&gm[i], &em[i], 4) || memcmp(&gv[i], &ev[i], 4)) ++moment_bits;
        dm = fmax(dm, fabs((double)gm[i] - em[i])); mx_m = fmax(mx_m, fabs((double)em[i]));
        dv = fmax(dv, fabs((double)gv[i] - ev[i])); mx_v = fmax(mx_v, fabs((double)ev[i]));
    }
    // each step's update is <= ~lr; 1-ulp sqrt and rcp put it off by <= ~3e-7 of itself, plus float32 roundings of the parameter
    // (|p| <= 0.15 -> spacing <= 1.5e-8) that can fall differently in the two copies.  The moments carry no approximation; they may
    // still differ in their last bits because the compiler contracts  m + (g - m)(1 - b1)  and  v b2 + (1 - b2) g g  into different
    // fused multiply-adds in the two kernels: bounded at float32 rounding level of the tensors' scale.
    const double total_bound = 2e-6 * lr * STEPS + 1.5e-8 * 10;
    const bool ok = drift <= total_bound && dm <= 1e-6 * mx_m && dv <= 4e-6 * mx_v;
    printf("[adam-approx-isolation] cfg%d %dx%d K=%d, %d steps: max |p_fused - p_exact| %.3e (bound %.3e; parameters moved by up to %.3e -> %.1e of the motion), "
           "largest growth per step %.2e; moments: max |dm| %.1e of max |m| %.1e, max |dv| %.1e of max |v| %.1e (%ld of %zu words differ in their last bits: "
           "FMA contraction) -> %s\n",
           cfg, M_real, N_real, K_real, STEPS, drift, total_bound, moved, drift / fmax(moved, 1e-30), worst_growth, dm, mx_m, dv, mx_v, moment_bits, 2 * P, ok ? "ok" : "FAIL");
    hipFree(D); hipFree(X); hipFree(p); hipFree(m); hipFree(v); hipFree(gk); hipFree(gb); hipFree(sh); hipFree(xp); hipFree(xm); hipFree(xv);
    return ok ? 0 : 1;
}

// ---- "mfmaclk": what the matrix pipes deliver from registers alone, and the shader clock they run at (round 4) -----------------
// Every wave issues independent-accumulator MFMAs with register operands for a few hundred microseconds: no LDS, no memory.  Per
// workgroup: shader-clock cycles and the 100-MHz real-time counter across the loop -> the clock under sustained matrix load; from
// HIP events around the launch: the TFLOP/s a perfect kernel could reach on THIS device (the denominator behind every
// "fraction of the MFMA peak" in DESIGN.md).
typedef __attribute__((ext_vector_type(16))) float tg_f32x16;
typedef __attribute__((ext_vector_type(8))) __bf16 tg_bf16x8;
template <int F32>
__global__ __launch_bounds__(256) void k_mfma_clock(unsigned long long* out, int iters, float seed)
{
    tg_f32x16 acc[4];
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[u][e] = 0.f;
    const float a = seed * (float)(threadIdx.x & 7), b = seed * 0.5f;
    tg_bf16x8 a8, b8;
#pragma unroll
    for (int e = 0; e < 8; ++e) { a8[e] = (__bf16)a; b8[e] = (__bf16)b; }
    const unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (F32) acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[u & 3], 0, 0, 0);
            else acc[u & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8, b8, acc[u & 3], 0, 0, 0);
        }
    }
    float sum = 0.f;
#pragma unroll
    for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 16; ++e) sum += acc[u][e];
    const unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) { out[blockIdx.x * 2] = c1 - c0; out[blockIdx.x * 2 + 1] = r1 - r0; }
    if (sum == 12345.678f) out[0] = 0;   // (keeps the accumulators alive)
}

static void mfma_clock(int f32, int wgs_per_cu, int cus)
{
    const int wgs = wgs_per_cu * cus, iters = f32 ? 400 : 800;
    unsigned long long* dst;
    CK(hipMalloc(&dst, (size_t)wgs * 16));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    float ms = 0.f;
    for (int rep = 0; rep < 3; ++rep) {   // the last of three back-to-back launches is reported (clocks settle)
        CK(hipEventRecord(e0));
        if (f32) hipLaunchKernelGGL(k_mfma_clock<1>, dim3(wgs), dim3(256), 0, 0, dst, iters, 1e-3f);
        else hipLaunchKernelGGL(k_mfma_clock<0>, dim3(wgs), dim3(256), 0, 0, dst, iters, 1e-3f);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        CK(hipEventElapsedTime(&ms, e0, e1));
    }
    std::vector<unsigned long long> h((size_t)wgs * 2);
    CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
    double lo = 1e9, hi = 0, mean = 0;
    for (int w = 0; w < wgs; ++w) {
        const double ghz = (double)h[w * 2] / ((double)h[w * 2 + 1] * 10.0);
        lo = std::min(lo, ghz); hi = std::max(hi, ghz); mean += ghz / wgs;
    }
    const double flop_per_mfma = f32 ? 32.0 * 32 * 2 * 2 : 32.0 * 32 * 16 * 2;
    const double flops = (double)wgs * 4 * iters * 16 * flop_per_mfma;
    const double cyc_per_mfma = 0.0;
    (void)cyc_per_mfma;
    double mc = 0;
    for (int w = 0; w < wgs; ++w) mc += (double)h[w * 2] / wgs;
    printf("mfmaclk %s  %d workgroup(s)/CU x 4 waves: %.1f us, %.1f TFLOP/s (%.3f of the %s peak); shader clock under load %.3f GHz (min %.3f, max %.3f); "
           "%.1f cycles per MFMA and wave\n", f32 ? "f32 32x32x2 " : "bf16 32x32x16", wgs_per_cu, ms * 1e3, flops / (ms * 1e-3) * 1e-12,
           flops / (ms * 1e-3) * 1e-12 / (f32 ? 157.3 : 2500.0), f32 ? "157.3-TF f32" : "2.5-PF bf16", mean, lo, hi, mc / ((double)iters * 16));
    hipFree(dst);
}

int main(int argc, char** argv)
{
    int fails = 0;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s  CUs=%d  gcnArch=%s\n", prop.name, prop.multiProcessorCount, prop.gcnArchName);
    if (argc > 2 && !strcmp(argv[2], "xcd")) g_xcd_block = 1;
    if (argc > 1 && !strcmp(argv[1], "ingest")) {   // the loader-wave variant with the compute waves idle: what a CU can take in
        unsigned long long* dst;
        CK(hipMalloc(&dst, 72 * 8));
        for (int skip : {2, 1, 0}) {
            CK(hipMemset(dst, 0, 72 * 8));
            rtx_gemm_dma_set_stamps(dst);
            rtx_gemm_dma_set_skip(skip);
            perf_dma(skip == 2 ? "dma-only" : skip == 1 ? "dma+fragment reads" : "all", RTX_FORM_NT, RTX_DMA_256x256_LW, 4096, 4096, 4096, 1, RTX_EPI_STORE);
            rtx_gemm_dma_set_stamps(nullptr);
            rtx_gemm_dma_set_skip(0);
            unsigned long long h[72];
            CK(hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost));
            printf("  a LOADER wave:  wait for my DMA | barrier | issue of the next slice's 16 pieces | total\n");
            for (int k = 0; k < 7; ++k)
                printf("    slice %2d: %6llu %6llu %6llu   = %llu\n", 8 + k, h[32 + k * 4 + 1] - h[32 + k * 4], h[32 + k * 4 + 2] - h[32 + k * 4 + 1],
                       h[32 + k * 4 + 3] - h[32 + k * 4 + 2], h[32 + (k + 1) * 4] - h[32 + k * 4]);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "stamps")) {   // where a K slice of the LDS-DMA GEMM goes (wave 0 of workgroup 0, slices 8..15)
        unsigned long long* dst;
        CK(hipMalloc(&dst, 72 * 8));
        for (int cfg : {RTX_DMA_256x256_LW, RTX_DMA_256x256, RTX_DMA_256x256_W4, RTX_DMA_512x128, RTX_DMA_128x128}) {
            CK(hipMemset(dst, 0, 72 * 8));
            rtx_gemm_dma_set_stamps(dst);
            perf_dma("sq4k-stamped", RTX_FORM_NT, cfg, 4096, 4096, 4096, 1, RTX_EPI_STORE);
            rtx_gemm_dma_set_stamps(nullptr);
            unsigned long long h[72];
            CK(hipMemcpy(h, dst, sizeof(h), hipMemcpyDeviceToHost));
            printf("  cfg %d, cycles per slice:  wait for my DMA | barrier | fragment reads + MFMAs (+ next slice's DMA issue) | total\n", cfg);
            for (int k = 0; k < 7; ++k)
                printf("    slice %2d: %6llu %6llu %6llu   = %llu\n", 8 + k, h[k * 4 + 1] - h[k * 4], h[k * 4 + 2] - h[k * 4 + 1], h[k * 4 + 3] - h[k * 4 + 2],
                       h[(k + 1) * 4] - h[k * 4]);
            printf("  workgroup 0: entry -> main loop done %llu cycles, epilogue %llu cycles; whole %llu cycles in %.2f us = %.3f GHz shader clock\n", h[66] - h[64],
                   h[67] - h[66], h[67] - h[64], (h[68] - h[65]) * 0.01, (double)(h[67] - h[64]) / ((h[68] - h[65]) * 10.0));
            if (cfg == RTX_DMA_256x256_LW) {
                printf("  cfg %d, a LOADER wave:  wait for my DMA | barrier | issue of the next slice's 16 pieces | total\n", cfg);
                for (int k = 0; k < 7; ++k)
                    printf("    slice %2d: %6llu %6llu %6llu   = %llu\n", 8 + k, h[32 + k * 4 + 1] - h[32 + k * 4], h[32 + k * 4 + 2] - h[32 + k * 4 + 1],
                           h[32 + k * 4 + 3] - h[32 + k * 4 + 2], h[32 + (k + 1) * 4] - h[32 + k * 4]);
            }
        }
        hipFree(dst);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "mfmaclk")) {
        for (int f32 : {1, 0})
            for (int per_cu : {1, 2, 4}) mfma_clock(f32, per_cu, prop.multiProcessorCount);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "chan")) {   // power-of-two row strides against odd multiples of 128 B (L2 channel spread)
        for (int cfg : {RTX_DMA_256x256, RTX_DMA_256x256_W4, RTX_DMA_256x256_LW})
            for (int K : {4096, 4160, 4032, 8192, 8256}) perf_dma("chan", RTX_FORM_NT, cfg, 4096, 4096, K, 1, RTX_EPI_STORE);
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "dwx")) {   // where a weight-gradient + Adam workgroup spends its life (round 4)
        const int cfg = argc > 2 ? atoi(argv[2]) : 0;
        const bool base_only = argc > 3 && !strcmp(argv[3], "base");   // (the counter passes: one kernel variant per kernel name)
        const bool stamps_only = argc > 3 && !strcmp(argv[3], "stamps");
        for (int mask : {0, 16, 1, 2, 4, 3, 5, 6, 7}) {
            if (stamps_only) break;
            if (base_only && mask) break;
            rtx_dw_set_skip(mask);
            printf("skip mask %d (1 = no K walk, 2 = no p/m/v loads, 4 = no stores, 16 = DMA issued first)\n", mask);
            perf_dw("dW4+adam", cfg, RTX_DW_ADAM, 20108, 600, 500, 3);
            perf_dw("dW1+adam", cfg, RTX_DW_ADAM, 600, 20108, 500, 3);
        }
        rtx_dw_set_skip(0);
        for (int shape = 0; shape < 2 && !base_only; ++shape) {
            const int M = shape ? 600 : 20108, N = shape ? 20108 : 600;
            const int wgs = 8 * (((rtx_pad(M) / rtx_dw_tile_rows(cfg)) * ((rtx_pad(N) + rtx_dw_tile_cols(cfg) - 1) / rtx_dw_tile_cols(cfg)) + 7) / 8);
            unsigned long long* dst;
            CK(hipMalloc(&dst, (size_t)wgs * 64));
            CK(hipMemset(dst, 0, (size_t)wgs * 64));
            rtx_dw_set_stamps(dst);
            perf_dw(shape ? "dW1+adam stamped" : "dW4+adam stamped", cfg, RTX_DW_ADAM, M, N, 500, 3);
            rtx_dw_set_stamps(nullptr);
            std::vector<unsigned long long> h((size_t)wgs * 8);
            CK(hipMemcpy(h.data(), dst, h.size() * 8, hipMemcpyDeviceToHost));
            hipFree(dst);
            if (const char* dump = getenv("DWX_DUMP")) {   // raw stamps of the last launch: [workgroup][8] u64 (tools/dw_stamps.py)
                char path[512];
                snprintf(path, sizeof(path), "%s_%d.bin", dump, shape);
                if (FILE* f = fopen(path, "wb")) { fwrite(h.data(), 8, h.size(), f); fclose(f); }
            }
            // the LAST launch's stamps: phases averaged over workgroups, and the launch's profile in time
            unsigned long long t0 = ~0ull, t1 = 0;
            int n = 0;
            for (int w = 0; w < wgs; ++w)
                if (h[w * 8]) { t0 = std::min(t0, h[w * 8]); t1 = std::max(t1, h[w * 8 + 6]); ++n; }
            double ph[6] = {0, 0, 0, 0, 0, 0};
            for (int w = 0; w < wgs; ++w)
                if (h[w * 8])
                    for (int k = 0; k < 6; ++k) ph[k] += (double)(h[w * 8 + k + 1] - h[w * 8 + k]) * 0.01 / n;
            printf("  %d workgroups, launch span %.1f us; mean us per phase: issue %.2f | p/m/v + slice 0 landed %.2f | K walk %.2f | drain + park %.2f | Adam + store issue %.2f | "
                   "stores acknowledged %.2f | life %.2f\n", n, (t1 - t0) * 0.01, ph[0], ph[1], ph[2], ph[3], ph[4], ph[5], ph[0] + ph[1] + ph[2] + ph[3] + ph[4] + ph[5]);
            // resident workgroups and phase census every 5 us
            for (double t = 2.5; t < (t1 - t0) * 0.01; t += 5.0) {
                const unsigned long long tt = t0 + (unsigned long long)(t * 100);
                int c[6] = {0, 0, 0, 0, 0, 0};
                for (int w = 0; w < wgs; ++w)
                    if (h[w * 8] && h[w * 8] <= tt && tt < h[w * 8 + 6])
                        for (int k = 5; k >= 0; --k)
                            if (tt >= h[w * 8 + k]) { ++c[k]; break; }
                printf("    t = %5.1f us: resident %4d | issue %3d wait-first %4d K-walk %4d park %3d adam %4d store-ack %4d\n", t, c[0] + c[1] + c[2] + c[3] + c[4] + c[5], c[0], c[1],
                       c[2], c[3], c[4], c[5]);
            }
            // per-XCD workgroup counts and mean life
            double life[8] = {0}; int cnt[8] = {0};
            for (int w = 0; w < wgs; ++w)
                if (h[w * 8]) { const int x = (int)(h[w * 8 + 7] >> 32) & 7; life[x] += (h[w * 8 + 6] - h[w * 8]) * 0.01; ++cnt[x]; }
            printf("    per XCC_ID: ");
            for (int x = 0; x < 8; ++x) printf("%d: %d wgs, life %.1f us | ", x, cnt[x], cnt[x] ? life[x] / cnt[x] : 0.0);
            printf("\n");
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "dwk")) {   // the weight-gradient + Adam kernel at a LONG K (a batch of thousands of rows), every tile
        for (int cfg : {0, 3, 4}) {
            perf_dw("netflix dW4+adam K=4096", cfg, RTX_DW_ADAM, 17769, 600, 4096, 3);
            perf_dw("netflix dW1+adam K=4096", cfg, RTX_DW_ADAM, 600, 17769, 4096, 3);
            perf_dw("ml20m dW4+adam K=500", cfg, RTX_DW_ADAM, 20108, 600, 500, 3);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "dwperf")) {   // the two n_items x 600 launches of the step, warm and cold
        for (int rep = 0; rep < 3; ++rep)
            for (int cfg : {0, 3, 8}) {   // 64 x 128; 32 x 256 on two / three stages (round 6)
                perf_dw("dW4+adam", cfg, RTX_DW_ADAM, 20108, 600, 500, 3);
                perf_dw("dW1+adam", cfg, RTX_DW_ADAM, 600, 20108, 500, 3);
            }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "dw")) {   // only the weight-gradient (+ Adam) kernels
        fails = run_dw_cases();
        printf("%s (%d failing cases)\n", fails ? "GEMM TESTS FAILED" : "GEMM TESTS PASSED", fails);
        return fails ? 1 : 0;
    }
#ifdef RTX_GEMM_ABLATE
    if (argc > 1 && !strcmp(argv[1], "ablate")) {   // round 6: where the first-layer product's 22 us go (parts of the loop removed; results are wrong by design)
        const int M = 512, N = 640, K = 20224, splits = 24;
        bf16_t *A, *B;
        float* C;
        std::vector<bf16_t> hA((size_t)M * K), hB((size_t)N * K);
        for (auto& v : hA) v = f32_to_bf16(frand());
        for (auto& v : hB) v = f32_to_bf16(frand());
        CK(hipMalloc(&A, hA.size() * 2)); CK(hipMalloc(&B, hB.size() * 2)); CK(hipMalloc(&C, (size_t)splits * M * N * 4));
        CK(hipMemcpy(A, hA.data(), hA.size() * 2, hipMemcpyHostToDevice)); CK(hipMemcpy(B, hB.data(), hB.size() * 2, hipMemcpyHostToDevice));
        RtxGemm g = {};
        g.A = A; g.B = B; g.lda = K; g.ldb = K; g.tile_shape = 0; g.m_tiles = M / 128; g.n_tiles = N / 128; g.k_slices = K * 2 / 128;
        g.splits = splits; g.C = C; g.ldc = N; g.slab_stride = (long)M * N; g.M_real = M; g.N_real = N;
        const int masks[] = {0, 64, 0, 64, 100, 3, 103, 7, 107, 1, 2, 4, 8, 11, 16, 32, 63};
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        for (int rep = 0; rep < 2; ++rep)
            for (int mk : masks) {
                for (int i = 0; i < 3; ++i) rtx_gemm_ablate_launch(g, mk, 0);
                CK(hipEventRecord(e0, 0));
                for (int i = 0; i < 20; ++i) rtx_gemm_ablate_launch(g, mk, 0);
                CK(hipEventRecord(e1, 0));
                CK(hipEventSynchronize(e1));
                float ms;
                CK(hipEventElapsedTime(&ms, e0, e1));
                if (mk >= 100) { printf("[ablate fwd1 512x640x20224 / 24] HOISTED fragment reads, mask %d: %.1f us\n", mk - 100, ms * 50.0); continue; }
                printf("[ablate fwd1 512x640x20224 / 24] mask %2d (%s%s%s%s%s%s): %.1f us\n", mk, mk & 1 ? "no-gload " : "", mk & 2 ? "no-ldswrite " : "", mk & 4 ? "no-mfma " : "",
                       mk & 8 ? "no-ldsread+mfma " : "", mk & 16 ? "no-barrier " : "", mk & 64 ? "second workgroup of a CU staggered by half a slice " : (mk & 32 ? "no-store " : ""), ms * 50.0);
            }
        return 0;
    }
#endif
    if (argc > 1 && !strcmp(argv[1], "skinny")) {   // round 6: the two K = n_items products on every kernel that can run them
        for (int K : {64, 128, 192, 256, 320, 384, 448, 704, 1408})
            fails += run_case<bf16_t>("store-d3", 256, 256, K, 1, RTX_EPI_STORE, 256, 256, RTX_TILE_128x128_D3);
        fails += run_case<bf16_t>("splitk11-d3", 256, 512, 1408, 11, RTX_EPI_STORE, 256, 512, RTX_TILE_128x128_D3);
        fails += run_case<bf16_t>("bias-d3", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, RTX_TILE_128x128_D3);
        fails += run_logits16_case(512, 768, 640, 410, 701, RTX_TILE_128x128_D3, 1.f);
        {   // round 6: the same product with K-BLOCKED operand images (a tile's slice = one contiguous 16-KB run): what the access pattern costs
            const int M = 512, N = 640, K = 20224;
            bf16_t *A, *B;
            float* C;
            CK(hipMalloc(&A, (size_t)M * K * 2)); CK(hipMalloc(&B, (size_t)N * K * 2)); CK(hipMalloc(&C, (size_t)24 * M * N * 4));
            CK(hipMemset(A, 0x11, (size_t)M * K * 2)); CK(hipMemset(B, 0x22, (size_t)N * K * 2));
            for (int blocked = 0; blocked < 2; ++blocked)
                for (int shape : {0, 4}) {
                    RtxGemm g = {};
                    g.A = A; g.B = B; g.tile_shape = shape; g.m_tiles = M / 128; g.n_tiles = N / 128; g.k_slices = K * 2 / 128;
                    g.splits = 24; g.C = C; g.ldc = N; g.slab_stride = (long)M * N; g.M_real = M; g.N_real = N;
                    if (blocked) { g.lda = 64; g.ldb = 64; g.a_slice_stride = (long)M * 128; g.b_slice_stride = (long)N * 128; }
                    else { g.lda = K; g.ldb = K; }
                    hipEvent_t e0, e1;
                    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
                    for (int i = 0; i < 3; ++i) rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_STORE, 0);
                    CK(hipEventRecord(e0, 0));
                    for (int i = 0; i < 20; ++i) rtx_gemm_launch(g, RTX_DT_BF16, RTX_EPI_STORE, 0);
                    CK(hipEventRecord(e1, 0));
                    CK(hipEventSynchronize(e1));
                    float ms;
                    CK(hipEventElapsedTime(&ms, e0, e1));
                    printf("[perf fwd1 layout] %s operands, tile%d, 24 slabs: %.1f us\n", blocked ? "K-BLOCKED [K/64][rows][64]" : "row-major                 ", shape, ms * 50.0);
                }
            hipFree(A); hipFree(B); hipFree(C);
        }
        for (int rep = 0; rep < 2; ++rep) {
            perf_case<bf16_t>("fwd1 regstage", 512, 640, 20224, 24, RTX_EPI_STORE, 0);
            perf_case<bf16_t>("fwd1 depth-3 ", 512, 640, 20224, 24, RTX_EPI_STORE, RTX_TILE_128x128_D3);
            perf_logits16(512, 20224, 640, 0);
            perf_logits16(512, 20224, 640, RTX_TILE_128x128_D3);
            for (int sp : {12, 16, 20, 23}) perf_dma("fwd1", RTX_FORM_NT, RTX_DMA_128x128_S2, 512, 640, 20224, sp, RTX_EPI_STORE);
            for (int sp : {12, 16, 20, 23}) perf_dma("dH3", RTX_FORM_NN, RTX_DMA_128x128_S2, 512, 640, 20224, sp, RTX_EPI_STORE);
            for (int sp : {8, 12}) perf_dma("fwd1", RTX_FORM_NT, RTX_DMA_128x128, 512, 640, 20224, sp, RTX_EPI_STORE);
        }
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "streams")) {   // round 6: the optimizer state's access pattern alone (2 / 4 / 6 streams, tiles vs flat)
        run_state_stream();
        return 0;
    }
    if (argc > 1 && !strcmp(argv[1], "adamiso")) {
        fails += run_adam_approx_isolation(RTX_DW_64x128);
        fails += run_adam_approx_isolation(RTX_DW_128x128);
        printf("%s (%d failing cases)\n", fails ? "GEMM TESTS FAILED" : "GEMM TESTS PASSED", fails);
        return fails ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "logits")) {   // round 6: the logits product, 128-byte slices (2 workgroups/CU) vs 64-byte slices (3/CU)
        fails += run_case<bf16_t>("bias-k32", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, RTX_TILE_128x128_K32);
        fails += run_case<bf16_t>("bias-wide-k32", 256, 2304, 64, 1, RTX_EPI_BIAS_ROWS, 250, 2300, RTX_TILE_128x128_K32);
        fails += run_case<bf16_t>("bias-odd-k32", 128, 256, 192, 1, RTX_EPI_BIAS_ROWS, 77, 131, RTX_TILE_128x128_K32);
        fails += run_logits16_case(512, 768, 640, 410, 701, RTX_TILE_128x128_K32, 1.f);
        fails += run_logits16_case(256, 2304, 128, 250, 2300, RTX_TILE_128x128_K32, 4000.f);
        for (int rep = 0; rep < 3; ++rep)
            for (int shape : {0, 3, 1}) perf_logits16(512, 20224, 640, shape);
        for (int shape : {0, 3}) perf_logits16(4096, 17920, 640, shape);
        printf("%s (%d failing cases)\n", fails ? "GEMM TESTS FAILED" : "GEMM TESTS PASSED", fails);
        return fails ? 1 : 0;
    }
    if (argc > 1 && !strcmp(argv[1], "big")) {   // only the big-tile GEMM lines (round 3)
        for (int cfg : {RTX_DMA_256x256, RTX_DMA_256x256_W4, RTX_DMA_512x128}) {
            for (int form : {RTX_FORM_NT, RTX_FORM_NN}) {
                fails += run_dma_case("store", form, cfg, 512, 768, 704, 1, RTX_EPI_STORE, 512, 768, 0);
                fails += run_dma_case("bias", form, cfg, 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, 0);
            }
            perf_dma("sq4k", RTX_FORM_NT, cfg, 4096, 4096, 4096, 1, RTX_EPI_STORE);
            perf_dma("sq4k", RTX_FORM_NN, cfg, 4096, 4096, 4096, 1, RTX_EPI_STORE);
            perf_dma("sq8k", RTX_FORM_NT, cfg, 8192, 8192, 8192, 1, RTX_EPI_STORE);
            perf_dma("nflx-logits", RTX_FORM_NT, cfg, 4096, 17920, 640, 1, RTX_EPI_BIAS_ROWS);
            perf_dma("nflx-fwd1", RTX_FORM_NT, cfg, 4096, 768, 17920, 5, RTX_EPI_STORE);
            perf_dma("nflx-dH3", RTX_FORM_NN, cfg, 4096, 768, 17920, 5, RTX_EPI_STORE);
            perf_dma("logits", RTX_FORM_NT, cfg, 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS);
        }
        printf("%s (%d failing cases)\n", fails ? "GEMM TESTS FAILED" : "GEMM TESTS PASSED", fails);
        return fails ? 1 : 0;
    }
    for (int shape = 0; shape < 3; ++shape) {   // 128x128, 256x128, 128x256
        fails += run_case<bf16_t>("store", 512, 768, 704, 1, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<bf16_t>("splitk3", 512, 768, 704, 3, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<bf16_t>("splitk5", 256, 256, 704, 5, RTX_EPI_STORE, 256, 256, shape);
        fails += run_case<bf16_t>("splitk11", 256, 512, 1408, 11, RTX_EPI_STORE, 256, 512, shape);
        fails += run_case<bf16_t>("bias", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, shape);
        fails += run_case<bf16_t>("grad", 768, 512, 512, 1, RTX_EPI_GRAD, 700, 300, shape);
        fails += run_case<bf16_t>("grad-tall", 2304, 256, 128, 1, RTX_EPI_GRAD, 2300, 200, shape);
        fails += run_case<bf16_t>("bias-wide", 256, 2304, 64, 1, RTX_EPI_BIAS_ROWS, 250, 2300, shape);
        fails += run_logits16_case(512, 768, 640, 410, 701, shape, 1.f);
        fails += run_logits16_case(256, 2304, 128, 250, 2300, shape, 4000.f);   // products beyond the half range: clamped to +-65504
        fails += run_case<float>("store", 512, 768, 352, 1, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<float>("splitk3", 512, 768, 352, 3, RTX_EPI_STORE, 512, 768, shape);
        fails += run_case<float>("bias", 512, 768, 320, 1, RTX_EPI_BIAS_ROWS, 410, 701, shape);
        fails += run_case<float>("grad", 768, 512, 256, 1, RTX_EPI_GRAD, 700, 300, shape);
    }
    // the depth-3 pipeline of the 128 x 128 tile: every slice count mod 6, split-K with short and empty last splits, both epilogues
    for (int K : {64, 128, 192, 256, 320, 384, 448, 704, 1408})
        fails += run_case<bf16_t>("store-d3", 256, 256, K, 1, RTX_EPI_STORE, 256, 256, RTX_TILE_128x128_D3);
    fails += run_case<bf16_t>("splitk3-d3", 512, 768, 704, 3, RTX_EPI_STORE, 512, 768, RTX_TILE_128x128_D3);
    fails += run_case<bf16_t>("splitk5-d3", 256, 256, 704, 5, RTX_EPI_STORE, 256, 256, RTX_TILE_128x128_D3);
    fails += run_case<bf16_t>("splitk11-d3", 256, 512, 1408, 11, RTX_EPI_STORE, 256, 512, RTX_TILE_128x128_D3);
    fails += run_case<bf16_t>("bias-d3", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, RTX_TILE_128x128_D3);
    fails += run_logits16_case(512, 768, 640, 410, 701, RTX_TILE_128x128_D3, 1.f);
    // the 64-byte-slice tile (three workgroups per CU): bias epilogue only, float32 and half logits, ragged edges, K of one slice
    fails += run_case<bf16_t>("bias-k32", 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, RTX_TILE_128x128_K32);
    fails += run_case<bf16_t>("bias-wide-k32", 256, 2304, 64, 1, RTX_EPI_BIAS_ROWS, 250, 2300, RTX_TILE_128x128_K32);
    fails += run_case<bf16_t>("bias-odd-k32", 128, 256, 192, 1, RTX_EPI_BIAS_ROWS, 77, 131, RTX_TILE_128x128_K32);
    fails += run_logits16_case(512, 768, 640, 410, 701, RTX_TILE_128x128_K32, 1.f);
    fails += run_logits16_case(256, 2304, 128, 250, 2300, RTX_TILE_128x128_K32, 4000.f);
    fails += tr_probe();
    for (int cfg = 0; cfg < 6; ++cfg) {   // 128x128 (4 waves, 3 stages), 512x128 and 256x256 (8 waves, 2 stages), 128x128 (2 stages), 256x256 on 4 waves, 256x256 on 8 + 4 loader waves
        for (int form : {RTX_FORM_NT, RTX_FORM_NN}) {
            fails += run_dma_case("store", form, cfg, 512, 768, 704, 1, RTX_EPI_STORE, 512, 768, 0);
            fails += run_dma_case("splitk3", form, cfg, 512, 768, 704, 3, RTX_EPI_STORE, 512, 768, 0);
            fails += run_dma_case("splitk11", form, cfg, 512, 512, 1408, 11, RTX_EPI_STORE, 512, 512, 0);
            fails += run_dma_case("k1", form, cfg, 512, 256, 64, 1, RTX_EPI_STORE, 512, 256, 0);
            fails += run_dma_case("bias", form, cfg, 512, 768, 640, 1, RTX_EPI_BIAS_ROWS, 410, 701, 0);
            fails += run_dma_case("bias-oddld", form, cfg, 512, 768, 128, 1, RTX_EPI_BIAS_ROWS, 500, 703, 1);
        }
        fails += run_dma_case("bias-wide", RTX_FORM_NT, cfg, 512, 2304, 64, 1, RTX_EPI_BIAS_ROWS, 500, 2300, 0);
    }
    fails += run_dw_cases();
    fails += run_adam_approx_isolation(RTX_DW_64x128);
    for (int form : {RTX_FORM_NN, RTX_FORM_TN}) {
        fails += run_f32_case("store", form, 256, 384, 352, 1, RTX_EPI_STORE, 256, 384);
        fails += run_f32_case("splitk3", form, 256, 384, 352, 3, RTX_EPI_STORE, 256, 384);
        fails += run_f32_case("k1", form, 128, 128, 32, 1, RTX_EPI_STORE, 128, 128);
        fails += run_f32_case("grad", form, 384, 256, 256, 1, RTX_EPI_GRAD, 300, 200);
    }
    if (argc > 1) {
        // the step's big contractions on the LDS-DMA kernels: logits, fwd-1 / dH3 (split-K), fused dW + Adam
        perf_dma("logits", RTX_FORM_NT, RTX_DMA_512x128, 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS);
        perf_dma("logits", RTX_FORM_NT, RTX_DMA_128x128, 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS);
        perf_dma("logits", RTX_FORM_NT, RTX_DMA_256x256, 512, 20224, 640, 1, RTX_EPI_BIAS_ROWS);
        perf_dma("logits-store", RTX_FORM_NT, RTX_DMA_512x128, 512, 20224, 640, 1, RTX_EPI_STORE);
        for (int sp : {23, 32, 46}) perf_dma("fwd1", RTX_FORM_NT, RTX_DMA_512x128, 512, 640, 20224, sp, RTX_EPI_STORE);
        for (int sp : {23, 32, 46}) perf_dma("dH3", RTX_FORM_NN, RTX_DMA_512x128, 512, 640, 20224, sp, RTX_EPI_STORE);
        perf_dma("fwd1", RTX_FORM_NT, RTX_DMA_128x128, 512, 640, 20224, 12, RTX_EPI_STORE);
        perf_dma("hidden", RTX_FORM_NT, RTX_DMA_128x128, 512, 512, 640, 5, RTX_EPI_STORE);
        perf_dma("hidden", RTX_FORM_NT, RTX_DMA_128x128, 512, 512, 640, 1, RTX_EPI_STORE);
        perf_dma("hidden", RTX_FORM_NN, RTX_DMA_128x128, 512, 640, 512, 4, RTX_EPI_STORE);
        for (int sp : {12, 24}) perf_dma("dH3", RTX_FORM_NN, RTX_DMA_128x128_S2, 512, 640, 20224, sp, RTX_EPI_STORE);
        perf_dma("dH3", RTX_FORM_NN, RTX_DMA_128x128, 512, 640, 20224, 12, RTX_EPI_STORE);
        perf_dma("sq4k", RTX_FORM_NT, RTX_DMA_256x256, 4096, 4096, 4096, 1, RTX_EPI_STORE);
        perf_dma("sq4k", RTX_FORM_NN, RTX_DMA_256x256, 4096, 4096, 4096, 1, RTX_EPI_STORE);
        perf_dma("sq4k", RTX_FORM_NT, RTX_DMA_512x128, 4096, 4096, 4096, 1, RTX_EPI_STORE);
        for (int cfg : {0, 2}) {
            perf_dw("dW4+adam", cfg, RTX_DW_ADAM, 20108, 600, 500);
            perf_dw("dW1+adam", cfg, RTX_DW_ADAM, 600, 20108, 500);
            perf_dw("dW4+adam", cfg, RTX_DW_ADAM, 20108, 600, 500, 3);
            perf_dw("dW1+adam", cfg, RTX_DW_ADAM, 600, 20108, 500, 3);
            perf_dw("dW4 grad", cfg, RTX_DW_GRAD, 20108, 600, 500);
        }
        perf_dw("dW-hidden+adam", 0, RTX_DW_ADAM, 400, 600, 500);
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
