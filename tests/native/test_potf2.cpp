// Native (no Python) check of the EASE solver's leaf (potf2.hip): W = inv(chol(A)) of one 128x128 SPD block, both leaves
// (round 3's blocked one, round 1's column-by-column one) against a host double-precision Cholesky + triangular inverse;
// not-positive-definite input must raise the status flag.  `perf` times a chain of 158 leaves (the ml-20m fit has that many).
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

#include "../../rectorch_amd/csrc/rtx_dgemm.h"
const char* rtx_last_error_str();

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(2); } } while (0)

static double frand() { return (double)rand() / RAND_MAX - 0.5; }

static int run_case(const char* name, int blocked, int ld, double shift, bool expect_fail)
{
    const int n = 128;
    std::vector<double> B(n * n), A((size_t)n * ld, 123.0), L(n * n, 0.0), Wref(n * n, 0.0);
    for (auto& v : B) v = frand();
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = 0; k < n; ++k) s += B[i * n + k] * B[j * n + k];
            A[(size_t)i * ld + j] = s + (i == j ? shift : 0.0);
            // the upper triangle keeps garbage (123): the leaf must read the lower triangle only
        }
    bool host_ok = true;
    for (int j = 0; j < n && host_ok; ++j) {       // host Cholesky (lower) ...
        double d = A[(size_t)j * ld + j];
        for (int k = 0; k < j; ++k) d -= L[j * n + k] * L[j * n + k];
        if (!(d > 0)) { host_ok = false; break; }
        L[j * n + j] = sqrt(d);
        for (int i = j + 1; i < n; ++i) {
            double s = A[(size_t)i * ld + j];
            for (int k = 0; k < j; ++k) s -= L[i * n + k] * L[j * n + k];
            L[i * n + j] = s / L[j * n + j];
        }
    }
    if (host_ok)
        for (int j = 0; j < n; ++j) {               // ... and its inverse, column by column
            Wref[j * n + j] = 1.0 / L[j * n + j];
            for (int i = j + 1; i < n; ++i) {
                double s = 0;
                for (int k = j; k < i; ++k) s -= L[i * n + k] * Wref[k * n + j];
                Wref[i * n + j] = s / L[i * n + i];
            }
        }
    double *dA, *dW, *dWT;
    int* dst;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dW, (size_t)n * ld * 8)); CK(hipMalloc(&dWT, (size_t)n * ld * 8)); CK(hipMalloc(&dst, 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dW, 0xff, (size_t)n * ld * 8)); CK(hipMemset(dWT, 0xff, (size_t)n * ld * 8)); CK(hipMemset(dst, 0, 4));
    rtx_potf2_set_blocked(blocked);
    int rc = rtx_potf2_inv_launch(dA, ld, dW, dWT, ld, dst, 0);
    if (rc) { printf("[%s] launch failed: %s\n", name, rtx_last_error_str()); return 1; }
    CK(hipDeviceSynchronize());
    int st = 0;
    CK(hipMemcpy(&st, dst, 4, hipMemcpyDeviceToHost));
    std::vector<double> W((size_t)n * ld), WT((size_t)n * ld);
    CK(hipMemcpy(W.data(), dW, W.size() * 8, hipMemcpyDeviceToHost));
    CK(hipMemcpy(WT.data(), dWT, WT.size() * 8, hipMemcpyDeviceToHost));
    hipFree(dA); hipFree(dW); hipFree(dWT); hipFree(dst);
    if (expect_fail) {
        const bool ok = st == 1 && !host_ok;
        printf("[%s] blocked=%d not positive definite: status=%d host_ok=%d -> %s\n", name, blocked, st, (int)host_ok, ok ? "ok" : "FAIL");
        return ok ? 0 : 1;
    }
    double err = 0, errT = 0, scale = 0, resid = 0;
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) {
            const double ref = (j <= i) ? Wref[i * n + j] : 0.0;
            err = fmax(err, fabs(W[(size_t)i * ld + j] - ref));
            errT = fmax(errT, fabs(WT[(size_t)j * ld + i] - ref));
            scale = fmax(scale, fabs(ref));
        }
    for (int i = 0; i < n; ++i)                     // W L = I
        for (int j = 0; j <= i; ++j) {
            double s = 0;
            for (int k = j; k <= i; ++k) s += W[(size_t)i * ld + k] * L[k * n + j];
            resid = fmax(resid, fabs(s - (i == j ? 1.0 : 0.0)));
        }
    const bool ok = st == 0 && err <= 1e-11 * scale * n && errT <= 1e-11 * scale * n && resid < 1e-10;
    printf("[%s] blocked=%d ld=%d shift=%g: max|W - ref| %.2e (W^T %.2e) of %.2e, max|W L - I| %.2e, status %d -> %s\n", name, blocked, ld, shift, err, errT, scale,
           resid, st, ok ? "ok" : "FAIL");
    return ok ? 0 : 1;
}

static void perf(int blocked)
{
    const int n = 128, ld = 20224, reps = 158;
    std::vector<double> A((size_t)n * ld, 0.0);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) A[(size_t)i * ld + j] = (i == j) ? 200.0 + i : 0.3 * frand();
    double *dA, *dW, *dWT;
    int* dst;
    CK(hipMalloc(&dA, A.size() * 8)); CK(hipMalloc(&dW, A.size() * 8)); CK(hipMalloc(&dWT, A.size() * 8)); CK(hipMalloc(&dst, 4));
    CK(hipMemcpy(dA, A.data(), A.size() * 8, hipMemcpyHostToDevice));
    CK(hipMemset(dst, 0, 4));
    rtx_potf2_set_blocked(blocked);
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int w = 0; w < 2; ++w) {
        CK(hipEventRecord(e0, 0));
        for (int r = 0; r < reps; ++r) rtx_potf2_inv_launch(dA, ld, dW, dWT, ld, dst, 0);
        CK(hipEventRecord(e1, 0));
        CK(hipDeviceSynchronize());
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    if (blocked) {   // phase stamps of one launch (100-MHz clock)
        unsigned long long* dst_;
        CK(hipMalloc(&dst_, 16 * 8));
        CK(hipMemset(dst_, 0, 16 * 8));
        rtx_potf2_set_stamps(dst_);
        rtx_potf2_inv_launch(dA, ld, dW, dWT, ld, dst, 0);
        CK(hipDeviceSynchronize());
        rtx_potf2_set_stamps(nullptr);
        unsigned long long h[16];
        CK(hipMemcpy(h, dst_, sizeof(h), hipMemcpyDeviceToHost));
        printf("[perf] phase stamps (us since kernel start): ");
        for (int k = 1; k < 16 && h[k]; ++k) printf("%.1f ", (double)(h[k] - h[0]) / 100.0);
        printf("\n   (loaded | after potrf panel 0, 1, 2 | potrf done | trtri done | stored)\n");
        hipFree(dst_);
    }
    printf("[perf] blocked=%d: %d leaves back to back %.2f ms = %.1f us per leaf\n", blocked, reps, ms, 1e3 * ms / reps);
    hipFree(dA); hipFree(dW); hipFree(dWT); hipFree(dst);
}

int main(int argc, char** argv)
{
    int fails = 0;
    srand(7);
    for (int blocked : {1, 0}) {
        fails += run_case("spd", blocked, 128, 128.0, false);
        fails += run_case("spd-wide-ld", blocked, 1000, 64.0, false);
        fails += run_case("spd-ill", blocked, 384, 1e-3, false);     // condition number ~1e5: the tolerance scales with |W|
        fails += run_case("not-pd", blocked, 128, -5.0, true);
    }
    if (argc > 1) { perf(1); perf(0); }
    printf("%s (%d failing cases)\n", fails ? "POTF2 TESTS FAILED" : "POTF2 TESTS PASSED", fails);
    return fails ? 1 : 0;
}
