// ease.hip -- EASE closed-form solver on MI355X (BASELINE.json configs[2]).
//
// Reference (rectorch/models.py:1015-1025, numpy float64 on the host):
//     X = train.toarray(); G = X^T X; G[diag] += lam; P = inv(G); B = P / (-diag(P)); B[diag] = 0; model = X B
// Here:
//   1. X^T as a dense K(=users)-contiguous operand, Gram matrix by ONE MFMA SYRK launch
//        - integer-valued data with max|x|^2 * n_users < 2^24 (implicit feedback): bf16 operands are exact and the f32
//          accumulators hold exact integers -> G is exact;  otherwise f64 MFMA.
//   2. G + lam I padded to a multiple of 128 with an identity block, then a RECURSIVE blocked Cholesky that carries
//      the inverse of every factored diagonal range along (ease_factor(lo, hi), below):
//        leaves   one 128x128 block: unscaled right-looking Cholesky + in-place triangular inverse in LDS, one
//                 workgroup of 1024 threads, one barrier per column (k_potf2_inv)
//        panel    L21 = A21 * inv(L11)^T          (f64 MFMA NT GEMM, K = width of the left half)
//        update   A22 -= L21 L21^T                (lower tiles only)
//        inverse  W21 = -W22 * (L21 * W11)        (two GEMMs; W = L^-1 is kept in both orientations)
//      Every GEMM of the recursion has K >= half its range, so the f64 MFMA pipe -- not the read-modify-write of C --
//      is the limit (a flat nb = 128 right-looking sweep measured 14 TFLOP/s; the big nodes of this form run at ~60).
//   3. P = W^T W as one GEMM on W^T (K from the diagonal block on), B = P / (-diag P) with a zero diagonal.
//   4. scores S_u = X_u B as a sparse-row x dense f64 product (k_ease_scores), -inf at the user's own items.
#include "../../include/rectorch_hip.h"
#include "rtx_dgemm.h"
#include "rtx_gemm.h"
#include "rtx_kernels.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#include <vector>

struct rtx_ease {
    int n = 0;
    double lam = 0;
    double* B = nullptr;   // [n][n] row-major
    int status = 0;        // 0 ok, 1 = matrix not positive definite
    double fit_ms = 0, gram_ms = 0, chol_ms = 0, inv_ms = 0;
};

// ------------------------------------------------------------------------------------------------ kernels
__global__ __launch_bounds__(256) void k_ease_scatter_T16(const int64_t* indptr, const int32_t* indices, const float* values, float vscale, long ldu,
                                                          bf16_t* XT)
{
    const int64_t u = blockIdx.x;
    for (int64_t k = indptr[u] + threadIdx.x; k < indptr[u + 1]; k += 256)
        XT[(size_t)indices[k] * ldu + u] = f32_to_bf16(values ? values[k] * vscale : 1.f);
}
// fp8 flavour: the hardware conversion produces the encoding the MFMA consumes (exact for integers |v| <= 16)
__global__ __launch_bounds__(256) void k_ease_scatter_T8(const int64_t* indptr, const int32_t* indices, const float* values, float vscale, long ldu,
                                                         uint8_t* XT)
{
    const int64_t u = blockIdx.x;
    for (int64_t k = indptr[u] + threadIdx.x; k < indptr[u + 1]; k += 256)
        XT[(size_t)indices[k] * ldu + u] = (uint8_t)(__builtin_amdgcn_cvt_pk_fp8_f32(values ? values[k] * vscale : 1.f, 0.f, 0, false) & 0xff);
}
__global__ __launch_bounds__(256) void k_ease_scatter_T64(const int64_t* indptr, const int32_t* indices, const float* values, long ldu, double* XT)
{
    const int64_t u = blockIdx.x;
    for (int64_t k = indptr[u] + threadIdx.x; k < indptr[u + 1]; k += 256)
        XT[(size_t)indices[k] * ldu + u] = values ? (double)values[k] : 1.0;
}

// A (f64 [np][np]) = G (+ lam on the real diagonal, 1 on the padded diagonal, 0 elsewhere in the pad)
template <typename TG>
__global__ __launch_bounds__(256) void k_ease_init(const TG* G, long ldg, double* A, int n, int np, double lam, double ginv)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)np * np) return;
    const int i = (int)(idx / np), j = (int)(idx % np);
    double v = 0.0;
    if (i < n && j < n) {   // the Gram kernels fill the lower triangle and the whole diagonal tiles
        if (i >= j || (i >> 7) == (j >> 7)) v = (double)G[(size_t)i * ldg + j] * ginv + (i == j ? lam : 0.0);   // ginv = 1 / s^2: a power of two
    }
    else if (i == j) v = 1.0;
    A[idx] = v;
}

// B[i][j] = P[i][j] / (-P[j][j]), B[i][i] = 0   (P symmetric, only its lower triangle is valid).  One workgroup per
// 64x64 tile of the lower triangle: the tile is written in place and, transposed through LDS, at its mirror position.
__global__ __launch_bounds__(256) void k_ease_B(const double* P, long ldp, double* B, int n)
{
    __shared__ double tile[64][65];
    __shared__ double drow[64], dcol[64];
    const int bi = blockIdx.y, bj = blockIdx.x;
    if (bj > bi) return;
    const int r0 = bi * 64, c0 = bj * 64, tid = threadIdx.x;
    if (tid < 64) drow[tid] = (r0 + tid < n) ? -P[(size_t)(r0 + tid) * ldp + r0 + tid] : 1.0;
    else if (tid < 128) dcol[tid - 64] = (c0 + tid - 64 < n) ? -P[(size_t)(c0 + tid - 64) * ldp + c0 + tid - 64] : 1.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = k * 256 + tid, rr = idx >> 6, cc = idx & 63;
        const int i = r0 + rr, j = c0 + cc;
        // inside a diagonal tile the upper half is read from its mirror
        tile[rr][cc] = (i < n && j < n) ? ((i >= j) ? P[(size_t)i * ldp + j] : P[(size_t)j * ldp + i]) : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = k * 256 + tid, rr = idx >> 6, cc = idx & 63;
        const int i = r0 + rr, j = c0 + cc;
        if (i < n && j < n) B[(size_t)i * n + j] = (i == j) ? 0.0 : tile[rr][cc] / dcol[cc];
    }
    if (bi != bj) {
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int idx = k * 256 + tid, cc = idx >> 6, rr = idx & 63;   // B[j][i] = P[i][j] / (-P[i][i])
            const int i = r0 + rr, j = c0 + cc;
            if (i < n && j < n) B[(size_t)j * n + i] = tile[rr][cc] / drow[rr];
        }
    }
}

// scores[b][:] = sum over the stored entries (i, v) of user row u_b of v * B[i][:]   (models.py:1025, 1054-1057)
// each workgroup owns 512 columns of one user; -inf at the non-zero entries of the mask row falling in its columns
__global__ __launch_bounds__(256) void k_ease_scores(const RtxCsrView x, const RtxCsrView mask, const double* B, int n, double* out)
{
    const int b = blockIdx.y;
    const int j0 = (blockIdx.x * 256 + threadIdx.x) * 2;
    const int64_t u = x.row_ids ? (int64_t)x.row_ids[b] : (int64_t)b;
    const int64_t beg = x.indptr[u], end = x.indptr[u + 1];
    if (j0 < n) {
        double s0 = 0.0, s1 = 0.0;
        const bool two = j0 + 1 < n;
        for (int64_t k = beg; k < end; ++k) {
            const double v = x.values ? (double)x.values[k] : 1.0;
            const double* row = B + (size_t)x.indices[k] * n + j0;
            s0 += v * row[0];
            if (two) s1 += v * row[1];
        }
        out[(size_t)b * n + j0] = s0;
        if (two) out[(size_t)b * n + j0 + 1] = s1;
    }
    if (mask.indptr) {
        __syncthreads();
        const int64_t mu = mask.row_ids ? (int64_t)mask.row_ids[b] : (int64_t)b;
        const int lo = blockIdx.x * 512, hi = lo + 512;
        for (int64_t k = mask.indptr[mu] + threadIdx.x; k < mask.indptr[mu + 1]; k += 256) {
            const int i = mask.indices[k];
            const float v = mask.values ? mask.values[k] : 1.f;
            if (v != 0.f && i >= lo && i < hi) out[(size_t)b * n + i] = -INFINITY;
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
static int dalloc(void** p, size_t bytes, std::vector<void*>& pool)
{
    hipError_t rc = hipMalloc(p, bytes ? bytes : 16);
    if (rc != hipSuccess) {
        rtx_set_error("ease: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(rc));
        return RTX_ENOMEM;
    }
    pool.push_back(*p);
    return RTX_OK;
}

struct EaseWork {
    double *A, *L, *W, *WT;   // [np][np] each: Gram matrix (lower, updated in place), panels of L, L^-1 and its transpose
    int np;
    int* status;
    hipStream_t st;
};

static int dgemm(const double* A, long lda, const double* B, long ldb, int m_tiles, int n_tiles, int k_slices, double* C, long ldc,
                 double* CT, long ldct, double alpha, double beta, int lower_only, int k_lo, int k_hi, hipStream_t st)
{
    if (m_tiles <= 0 || n_tiles <= 0) return RTX_OK;
    RtxDgemm g = {};
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.m_tiles = m_tiles; g.n_tiles = n_tiles; g.k_slices = k_slices;
    g.C = C; g.ldc = ldc; g.CT = CT; g.ldct = ldct; g.alpha = alpha; g.beta = beta; g.lower_only = lower_only;
    g.k_lo = k_lo; g.k_hi = k_hi;
    // all but the top two levels of the recursion have too few 128x128 tiles to fill 256 compute units: quarter tiles
    // put them on 4x as many (the triangular K ranges and the lower-only rule are tile-relative, so they carry over
    // with the finer grid).  Measured fit time by threshold: 0 -> 252.7 ms, 16 -> 241.4, 400..1600 -> 230.8, all -> 238.3
    constexpr int small_max = 1024;
    if (m_tiles * n_tiles <= small_max) { g.small_tile = 1; g.m_tiles = 2 * m_tiles; g.n_tiles = 2 * n_tiles; }
    return rtx_dgemm_launch(g, st);
}

// Block rows/columns [lo, hi) of the padded matrix (units of 128).  On return W[lo:hi, lo:hi] = inv(L[lo:hi, lo:hi])
// (W^T likewise) and the strictly-lower block panels of L inside the range are in w.L.
static int ease_factor(const EaseWork& w, int lo, int hi)
{
    const long np = w.np;
    if (hi - lo == 1) {
        const size_t d = (size_t)lo * 128 * np + (size_t)lo * 128;
        return rtx_potf2_inv_launch(w.A + d, np, w.W + d, w.WT + d, np, w.status, w.st);
    }
    const int mid = lo + (hi - lo + 1) / 2, n1 = mid - lo, n2 = hi - mid;
    RTX_TRY(ease_factor(w, lo, mid));
    const size_t d1 = (size_t)lo * 128 * np + (size_t)lo * 128;     // block (lo, lo)
    const size_t d2 = (size_t)mid * 128 * np + (size_t)mid * 128;   // block (mid, mid)
    const size_t o21 = (size_t)mid * 128 * np + (size_t)lo * 128;   // block (mid, lo)
    const size_t o12 = (size_t)lo * 128 * np + (size_t)mid * 128;   // block (lo, mid)
    // panel: L21[m][n] = sum_k A21[m][k] W11[n][k]   (W11 lower triangular: k < 128 (tn + 1))
    RTX_TRY(dgemm(w.A + o21, np, w.W + d1, np, n2, n1, 8 * n1, w.L + o21, np, nullptr, 0, 1.0, 0.0, 0, RTX_DK_ALL, RTX_DK_TN, w.st));
    // update: A22 -= L21 L21^T (lower tiles)
    RTX_TRY(dgemm(w.L + o21, np, w.L + o21, np, n2, n2, 8 * n1, w.A + d2, np, nullptr, 0, -1.0, 1.0, 1, RTX_DK_ALL, RTX_DK_ALL, w.st));
    RTX_TRY(ease_factor(w, mid, hi));
    // T^T[n][m] = sum_k W11^T[n][k] L21[m][k]   (W11^T upper triangular: k >= 128 tm); scratch = the unused mirror block of A
    double* TT = w.A + o12;
    RTX_TRY(dgemm(w.WT + d1, np, w.L + o21, np, n1, n2, 8 * n1, TT, np, nullptr, 0, 1.0, 0.0, 0, RTX_DK_TM, RTX_DK_ALL, w.st));
    // W21[m][n] = -sum_j W22[m][j] T^T[n][j]     (W22 lower triangular: j < 128 (tm + 1)); also stored into W^T
    RTX_TRY(dgemm(w.W + d2, np, TT, np, n2, n1, 8 * n2, w.W + o21, np, w.WT + o12, np, -1.0, 0.0, 0, RTX_DK_ALL, RTX_DK_TM, w.st));
    return RTX_OK;
}

static double elapsed_ms(hipEvent_t a, hipEvent_t b)
{
    float ms = 0.f;
    (void)hipEventElapsedTime(&ms, a, b);
    return ms;
}

extern "C" {

int rtx_ease_fit(const rtx_csr* X, double lam, rtx_ease** out, void* stream)
{
    RTX_CHECK(X && out, RTX_EINVAL, "ease_fit: NULL argument");
    RTX_CHECK(X->n_rows > 0 && X->n_cols > 0, RTX_EINVAL, "ease_fit: empty matrix");
    hipStream_t st = (hipStream_t)stream;
    const int n = X->n_cols;
    const long U = X->n_rows;
    const int np = ((n + 127) / 128) * 128;
    const int KB = np / 128;
    std::vector<void*> pool;
    int rc = RTX_OK;
    hipEvent_t e0, e1, e2, e3, e4;
    RTX_HIP(hipEventCreate(&e0)); RTX_HIP(hipEventCreate(&e1)); RTX_HIP(hipEventCreate(&e2)); RTX_HIP(hipEventCreate(&e3)); RTX_HIP(hipEventCreate(&e4));
    rtx_ease* h = new rtx_ease();
    h->n = n; h->lam = lam;
#define EASE_TRY(x) do { rc = (x); if (rc) goto done; } while (0)
#define EASE_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { rtx_set_error("ease: %s -> %s", #x, hipGetErrorString(e_)); rc = RTX_EHIP; goto done; } } while (0)
    {
        double *A = nullptr, *L = nullptr, *W = nullptr, *WT = nullptr;
        int* d_status = nullptr;
        // ---- exactness test for the low-precision Gram paths (host pass over the values; binary data has values == NULL):
        //      integer-valued entries whose products sum below 2^24 are exact in f32 accumulators; the operands are exact
        //      in fp8 e4m3 up to |v| = 16 and in bf16 up to |v| = 256
        //      Dyadic ratings (half or quarter stars) become integers after a power-of-two scale s: G = (sX)^T (sX) / s^2,
        //      with the division exact in float64 -- so explicit-feedback data take the MFMA path too.
        int gram = RTX_DT_FP8;   // RTX_DT_FP8 / RTX_DT_BF16, or RTX_DT_F32 meaning "no: use the f64 path"
        float vscale = 1.f;      // s
        {
            double mx = 1.0;
            if (X->values && X->nnz > 0) {
                std::vector<float> hv((size_t)X->nnz);
                EASE_HIP(hipMemcpy(hv.data(), X->values, sizeof(float) * X->nnz, hipMemcpyDeviceToHost));
                gram = RTX_DT_F32;
                for (float s : {1.f, 2.f, 4.f, 8.f}) {
                    bool ok = true;
                    double m = 0;
                    for (float v : hv) {
                        const float sv = v * s;
                        if (sv != rintf(sv) || fabsf(sv) > 256.f) { ok = false; break; }
                        m = fmax(m, fabs((double)sv));
                    }
                    if (ok) { gram = RTX_DT_FP8; vscale = s; mx = m; break; }
                }
            }
            if (gram != RTX_DT_F32 && mx * mx * (double)U >= 16777216.0) gram = RTX_DT_F32;
            if (gram == RTX_DT_FP8 && mx > 16.0) gram = RTX_DT_BF16;
            if (gram == RTX_DT_F32) vscale = 1.f;
        }
        EASE_TRY(dalloc((void**)&A, sizeof(double) * (size_t)np * np, pool));
        EASE_HIP(hipEventRecord(e0, st));
        // ---- 1. Gram matrix
        if (gram != RTX_DT_F32) {
            const int esz = (gram == RTX_DT_FP8) ? 1 : 2;
            const long Up = ((U + 127) / 128) * 128;
            const long np256 = ((np + 255) / 256) * 256;   // the Gram kernel works on 256-row tiles
            void* XT = nullptr;
            float* G32 = nullptr;
            EASE_TRY(dalloc(&XT, (size_t)esz * np256 * Up, pool));
            EASE_TRY(dalloc((void**)&G32, sizeof(float) * (size_t)np256 * np, pool));
            EASE_HIP(hipMemsetAsync(XT, 0, (size_t)esz * np256 * Up, st));
            if (gram == RTX_DT_FP8)
                hipLaunchKernelGGL(k_ease_scatter_T8, dim3((unsigned)U), dim3(256), 0, st, X->indptr, X->indices, X->values, vscale, Up, (uint8_t*)XT);
            else
                hipLaunchKernelGGL(k_ease_scatter_T16, dim3((unsigned)U), dim3(256), 0, st, X->indptr, X->indices, X->values, vscale, Up, (bf16_t*)XT);
            EASE_TRY(rtx_syrk_lower_launch(XT, Up * esz, 128, (int)(np256 / 256), KB, (int)(Up * esz / 128), gram == RTX_DT_FP8, G32, np, st));
            hipLaunchKernelGGL(k_ease_init<float>, dim3((unsigned)(((long)np * np + 255) / 256)), dim3(256), 0, st, G32, (long)np, A, n, np, lam,
                               1.0 / ((double)vscale * vscale));
        } else {
            const long Up = ((U + 15) / 16) * 16;
            double* XT = nullptr;
            double* G64 = nullptr;
            EASE_TRY(dalloc((void**)&XT, sizeof(double) * (size_t)np * Up, pool));
            EASE_TRY(dalloc((void**)&G64, sizeof(double) * (size_t)np * np, pool));
            EASE_HIP(hipMemsetAsync(XT, 0, sizeof(double) * (size_t)np * Up, st));
            hipLaunchKernelGGL(k_ease_scatter_T64, dim3((unsigned)U), dim3(256), 0, st, X->indptr, X->indices, X->values, Up, XT);
            EASE_TRY(dgemm(XT, Up, XT, Up, KB, KB, (int)(Up / 16), G64, np, nullptr, 0, 1.0, 0.0, 1, RTX_DK_ALL, RTX_DK_ALL, st));
            // mirror is not needed: only the lower triangle of A is read below; init copies what is there
            hipLaunchKernelGGL(k_ease_init<double>, dim3((unsigned)(((long)np * np + 255) / 256)), dim3(256), 0, st, G64, (long)np, A, n, np, lam, 1.0);
        }
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e1, st));
        // ---- 2. recursive Cholesky + inverse of the factor
        EASE_TRY(dalloc((void**)&L, sizeof(double) * (size_t)np * np, pool));
        EASE_TRY(dalloc((void**)&W, sizeof(double) * (size_t)np * np, pool));
        EASE_TRY(dalloc((void**)&WT, sizeof(double) * (size_t)np * np, pool));
        EASE_TRY(dalloc((void**)&d_status, sizeof(int), pool));
        EASE_HIP(hipMemsetAsync(d_status, 0, sizeof(int), st));
        // W (lower) and WT (upper) need no zero fill: every block that is read -- W11 / W22 up to the diagonal (k_hi), WT11
        // from the diagonal on (k_lo), WT from max(tm, tn) on in the final product -- is written first, by a leaf (full
        // 128x128 diagonal blocks) or by a merge (W21 and its transpose)
        {
            EaseWork w = {A, L, W, WT, np, d_status, st};
            EASE_TRY(ease_factor(w, 0, KB));
        }
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e2, st));
        // ---- 3. P (into the lower tiles of A) = W^T W : P[i][j] = sum_{k >= max(i,j)} WT[i][k] WT[j][k]
        EASE_TRY(dgemm(WT, np, WT, np, KB, KB, np / 16, A, np, nullptr, 0, 1.0, 0.0, 1, RTX_DK_MAX, RTX_DK_ALL, st));
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e3, st));
        // ---- 4. B
        {
            hipError_t e_ = hipMalloc((void**)&h->B, sizeof(double) * (size_t)n * n);
            if (e_ != hipSuccess) { rtx_set_error("ease: hipMalloc(B) failed: %s", hipGetErrorString(e_)); rc = RTX_ENOMEM; goto done; }
        }
        hipLaunchKernelGGL(k_ease_B, dim3((n + 63) / 64, (n + 63) / 64), dim3(256), 0, st, A, (long)np, h->B, n);
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e4, st));
        EASE_HIP(hipStreamSynchronize(st));
        EASE_HIP(hipMemcpy(&h->status, d_status, sizeof(int), hipMemcpyDeviceToHost));
        h->gram_ms = elapsed_ms(e0, e1); h->chol_ms = elapsed_ms(e1, e2); h->inv_ms = elapsed_ms(e2, e3); h->fit_ms = elapsed_ms(e0, e4);
        if (h->status != 0) { rtx_set_error("ease_fit: X^T X + lam I is not positive definite (lam = %g)", lam); rc = RTX_EINVAL; }
    }
done:
    for (void* p : pool) (void)hipFree(p);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipEventDestroy(e2); (void)hipEventDestroy(e3); (void)hipEventDestroy(e4);
    if (rc) {
        if (h->B) (void)hipFree(h->B);
        delete h;
        return rc;
    }
    *out = h;
    return RTX_OK;
#undef EASE_TRY
#undef EASE_HIP
}

int rtx_ease_destroy(rtx_ease* h)
{
    if (!h) return RTX_OK;
    if (h->B) (void)hipFree(h->B);
    delete h;
    return RTX_OK;
}

int rtx_ease_weights(const rtx_ease* h, double** B_dev, int32_t* n)
{
    RTX_CHECK(h, RTX_EINVAL, "ease: NULL handle");
    if (B_dev) *B_dev = h->B;
    if (n) *n = h->n;
    return RTX_OK;
}

int rtx_ease_copy_weights(const rtx_ease* h, double* dst_dev, void* stream)
{
    RTX_CHECK(h && dst_dev, RTX_EINVAL, "ease: NULL argument");
    RTX_HIP(hipMemcpyAsync(dst_dev, h->B, sizeof(double) * (size_t)h->n * h->n, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return RTX_OK;
}

int rtx_ease_timings(const rtx_ease* h, double* fit_ms, double* gram_ms, double* chol_ms, double* inv_ms)
{
    RTX_CHECK(h, RTX_EINVAL, "ease: NULL handle");
    if (fit_ms) *fit_ms = h->fit_ms;
    if (gram_ms) *gram_ms = h->gram_ms;
    if (chol_ms) *chol_ms = h->chol_ms;
    if (inv_ms) *inv_ms = h->inv_ms;
    return RTX_OK;
}

int rtx_ease_scores(const rtx_ease* h, const rtx_csr* X, const int32_t* row_ids, int32_t batch, const rtx_csr* mask,
                    const int32_t* mask_row_ids, double* out, void* stream)
{
    RTX_CHECK(h && X && out, RTX_EINVAL, "ease_scores: NULL argument");
    RTX_CHECK(X->n_cols == h->n, RTX_EINVAL, "ease_scores: matrix has %d columns, model has %d items", X->n_cols, h->n);
    RTX_CHECK(batch >= 0 && (row_ids || batch <= X->n_rows), RTX_EINVAL, "ease_scores: bad batch %d", batch);
    RTX_CHECK(!mask || (mask->n_cols == h->n && (mask_row_ids || batch <= mask->n_rows)), RTX_EINVAL,
              "ease_scores: mask matrix does not match (%d columns, %lld rows)", mask ? mask->n_cols : 0,
              mask ? (long long)mask->n_rows : 0LL);
    if (batch == 0) return RTX_OK;
    RtxCsrView v = {X->indptr, X->indices, X->values, row_ids};
    RtxCsrView mv = {nullptr, nullptr, nullptr, nullptr};
    if (mask) mv = RtxCsrView{mask->indptr, mask->indices, mask->values, mask_row_ids};
    hipLaunchKernelGGL(k_ease_scores, dim3((h->n + 511) / 512, batch), dim3(256), 0, (hipStream_t)stream, v, mv, h->B, h->n, out);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

}  // extern "C"
