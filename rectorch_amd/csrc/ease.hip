// ease.hip -- EASE closed-form solver on MI355X (BASELINE.json configs[2]).
//
// Reference (rectorch/models.py:1015-1025, numpy float64 on the host):
//     X = train.toarray(); G = X^T X; G[diag] += lam; P = inv(G); B = P / (-diag(P)); B[diag] = 0; model = X B
// Here:
//   1. X^T as a dense K(=users)-contiguous operand, Gram matrix by ONE MFMA SYRK launch
//        - integer-valued data with max|x|^2 * n_users < 2^24 (implicit feedback): bf16 operands are exact and the f32
//          accumulators hold exact integers -> G is exact;  otherwise f64 MFMA.
//   2. G + lam I padded to a multiple of 128 with an identity block, blocked Cholesky (nb = 128) in f64:
//        diagonal block factor + its triangular inverse in LDS (one workgroup), panel = A21 * inv(L11)^T and the
//        trailing update A22 -= L21 L21^T as f64 MFMA NT GEMMs (rtx_dgemm).
//   3. W = L^-1 by blocked triangular inversion (two GEMMs per block column), P = W^T W (GEMM on W^T, summed from the
//      diagonal block on), B = P / (-diag P) with a zero diagonal.
//   4. scores S_u = X_u B as a sparse-row x dense f64 product (k_ease_scores), -inf at the user's own items.
#include "../../include/rectorch_hip.h"
#include "rtx_dgemm.h"
#include "rtx_gemm.h"
#include "rtx_kernels.h"

#include <math.h>
#include <vector>

struct rtx_ease {
    int n = 0;
    double lam = 0;
    double* B = nullptr;   // [n][n] row-major
    int status = 0;        // 0 ok, 1 = matrix not positive definite
    double fit_ms = 0, gram_ms = 0, chol_ms = 0, inv_ms = 0;
};

// ------------------------------------------------------------------------------------------------ kernels
__global__ __launch_bounds__(256) void k_ease_scatter_T16(const int64_t* indptr, const int32_t* indices, const float* values, long ldu, bf16_t* XT)
{
    const int64_t u = blockIdx.x;
    for (int64_t k = indptr[u] + threadIdx.x; k < indptr[u + 1]; k += 256)
        XT[(size_t)indices[k] * ldu + u] = f32_to_bf16(values ? values[k] : 1.f);
}
__global__ __launch_bounds__(256) void k_ease_scatter_T64(const int64_t* indptr, const int32_t* indices, const float* values, long ldu, double* XT)
{
    const int64_t u = blockIdx.x;
    for (int64_t k = indptr[u] + threadIdx.x; k < indptr[u + 1]; k += 256)
        XT[(size_t)indices[k] * ldu + u] = values ? (double)values[k] : 1.0;
}

// A (f64 [np][np]) = G (+ lam on the real diagonal, 1 on the padded diagonal, 0 elsewhere in the pad)
template <typename TG>
__global__ __launch_bounds__(256) void k_ease_init(const TG* G, long ldg, double* A, int n, int np, double lam)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)np * np) return;
    const int i = (int)(idx / np), j = (int)(idx % np);
    double v = 0.0;
    if (i < n && j < n) v = (double)G[(size_t)i * ldg + j] + (i == j ? lam : 0.0);
    else if (i == j) v = 1.0;
    A[idx] = v;
}

// One 128x128 diagonal block: Cholesky factor L (lower) in place, W = L^-1 and W^T into compact [128][128] buffers.
// LDS tile of 128 x 129 doubles: L in the lower triangle, the inverse is built transposed in the free upper triangle.
__global__ __launch_bounds__(256) void k_potf2_inv(double* Akk, long ld, double* Winv, double* WinvT, int* status)
{
    extern __shared__ __attribute__((aligned(16))) double t[];   // [128][129]
    __shared__ double dinv[128];
    const int tid = threadIdx.x;
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        t[i * 129 + j] = Akk[(size_t)i * ld + j];
    }
    __syncthreads();
    for (int j = 0; j < 128; ++j) {
        const double ajj = t[j * 129 + j];
        if (!(ajj > 0.0)) {
            if (tid == 0) *status = 1;
            return;   // uniform: every thread reads the same value
        }
        const double d = sqrt(ajj);
        __syncthreads();
        for (int i = j + tid; i < 128; i += 256) t[i * 129 + j] = (i == j) ? d : t[i * 129 + j] / d;
        __syncthreads();
        // trailing update of the lower triangle: a[i][c] -= l[i][j] * l[c][j],  j < c <= i
        const int m = 127 - j;   // rows / cols j+1 .. 127
        for (int e = tid; e < m * m; e += 256) {
            const int c = j + 1 + e / m, i = j + 1 + e % m;
            if (i >= c) t[i * 129 + c] -= t[i * 129 + j] * t[c * 129 + j];
        }
        __syncthreads();
    }
    if (tid < 128) dinv[tid] = 1.0 / t[tid * 129 + tid];
    __syncthreads();
    // inverse by forward substitution, one column per thread: x_c = 1/l_cc, x_i = -(sum_{j=c}^{i-1} l_ij x_j) / l_ii
    // x_i (i > c) is stored at t[c][i] (upper triangle)
    if (tid < 128) {
        const int c = tid;
        for (int i = c + 1; i < 128; ++i) {
            double s = t[i * 129 + c] * dinv[c];
            for (int j = c + 1; j < i; ++j) s += t[i * 129 + j] * t[c * 129 + j];
            t[c * 129 + i] = -s * dinv[i];
        }
    }
    __syncthreads();
    for (int e = tid; e < 128 * 128; e += 256) {
        const int i = e >> 7, j = e & 127;
        const double l = (i >= j) ? t[i * 129 + j] : 0.0;
        const double w = (i > j) ? t[j * 129 + i] : (i == j ? dinv[i] : 0.0);   // W[i][j], lower triangular
        Akk[(size_t)i * ld + j] = l;
        Winv[i * 128 + j] = w;
        WinvT[j * 128 + i] = w;
    }
}

// copy the 128x128 inverse block onto the diagonal of W
__global__ __launch_bounds__(256) void k_copy_block(const double* src, double* dst, long ld)
{
    for (int e = threadIdx.x; e < 128 * 128; e += 256) dst[(size_t)(e >> 7) * ld + (e & 127)] = src[e];
}

__global__ __launch_bounds__(256) void k_transpose_f64(const double* __restrict__ in, long ld_in, double* __restrict__ out, long ld_out)
{
    __shared__ double tile[64][65];
    const int c0 = blockIdx.x * 64, r0 = blockIdx.y * 64, tid = threadIdx.x;
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = k * 256 + tid, rr = idx >> 6, cc = idx & 63;
        tile[rr][cc] = in[(size_t)(r0 + rr) * ld_in + c0 + cc];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int idx = k * 256 + tid, cc = idx >> 6, rr = idx & 63;
        out[(size_t)(c0 + cc) * ld_out + r0 + rr] = tile[rr][cc];
    }
}

// B[i][j] = P[i][j] / (-P[j][j]), B[i][i] = 0   (P symmetric, only its lower triangle is valid)
__global__ __launch_bounds__(256) void k_ease_B(const double* P, long ldp, double* B, int n)
{
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (long)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    const double pij = (i >= j) ? P[(size_t)i * ldp + j] : P[(size_t)j * ldp + i];
    B[idx] = (i == j) ? 0.0 : pij / (-P[(size_t)j * ldp + j]);
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

static int dgemm(const double* A, long lda, const double* B, long ldb, int m_tiles, int n_tiles, int k_slices, double* C, long ldc,
                 double alpha, double beta, int lower_only, int k_from, int k_to, hipStream_t st)
{
    if (m_tiles <= 0 || n_tiles <= 0) return RTX_OK;
    RtxDgemm g = {};
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb; g.m_tiles = m_tiles; g.n_tiles = n_tiles; g.k_slices = k_slices;
    g.C = C; g.ldc = ldc; g.alpha = alpha; g.beta = beta; g.lower_only = lower_only; g.k_from_tile = k_from; g.k_to_tile = k_to;
    return rtx_dgemm_launch(g, st);
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
        double *A = nullptr, *W = nullptr, *WT = nullptr, *T1T = nullptr, *Winv = nullptr, *WinvT = nullptr;
        int* d_status = nullptr;
        // ---- exactness test for the bf16 Gram path (host pass over the values; binary data has values == NULL)
        bool use_bf16 = true;
        if (X->values && X->nnz > 0) {
            std::vector<float> hv((size_t)X->nnz);
            EASE_HIP(hipMemcpy(hv.data(), X->values, sizeof(float) * X->nnz, hipMemcpyDeviceToHost));
            double mx = 0;
            for (float v : hv) {
                if (v != rintf(v) || fabsf(v) > 256.f) { use_bf16 = false; break; }
                mx = fmax(mx, fabs((double)v));
            }
            if (mx * mx * (double)U >= 16777216.0) use_bf16 = false;
        } else if ((double)U >= 16777216.0) {
            use_bf16 = false;
        }
        EASE_TRY(dalloc((void**)&A, sizeof(double) * (size_t)np * np, pool));
        EASE_HIP(hipEventRecord(e0, st));
        // ---- 1. Gram matrix
        if (use_bf16) {
            const long Up = ((U + 127) / 128) * 128;
            bf16_t* XT = nullptr;
            float* G32 = nullptr;
            EASE_TRY(dalloc((void**)&XT, sizeof(bf16_t) * (size_t)np * Up, pool));
            EASE_TRY(dalloc((void**)&G32, sizeof(float) * (size_t)np * np, pool));
            EASE_HIP(hipMemsetAsync(XT, 0, sizeof(bf16_t) * (size_t)np * Up, st));
            hipLaunchKernelGGL(k_ease_scatter_T16, dim3((unsigned)U), dim3(256), 0, st, X->indptr, X->indices, X->values, Up, XT);
            RtxGemm g = {};
            g.A = XT; g.B = XT; g.lda = Up; g.ldb = Up; g.tile_shape = RTX_TILE_128x128;
            g.m_tiles = KB; g.n_tiles = KB; g.k_slices = (int)(Up * 2 / 128); g.splits = 1;
            g.C = G32; g.ldc = np; g.slab_stride = 0; g.M_real = np; g.N_real = np;
            EASE_TRY(rtx_gemm_launch(g, 1, RTX_EPI_STORE, st));
            hipLaunchKernelGGL(k_ease_init<float>, dim3((unsigned)(((long)np * np + 255) / 256)), dim3(256), 0, st, G32, (long)np, A, n, np, lam);
        } else {
            const long Up = ((U + 15) / 16) * 16;
            double* XT = nullptr;
            double* G64 = nullptr;
            EASE_TRY(dalloc((void**)&XT, sizeof(double) * (size_t)np * Up, pool));
            EASE_TRY(dalloc((void**)&G64, sizeof(double) * (size_t)np * np, pool));
            EASE_HIP(hipMemsetAsync(XT, 0, sizeof(double) * (size_t)np * Up, st));
            hipLaunchKernelGGL(k_ease_scatter_T64, dim3((unsigned)U), dim3(256), 0, st, X->indptr, X->indices, X->values, Up, XT);
            EASE_TRY(dgemm(XT, Up, XT, Up, KB, KB, (int)(Up / 16), G64, np, 1.0, 0.0, 1, 0, 0, st));
            // mirror is not needed: only the lower triangle of A is read below; init copies what is there
            hipLaunchKernelGGL(k_ease_init<double>, dim3((unsigned)(((long)np * np + 255) / 256)), dim3(256), 0, st, G64, (long)np, A, n, np, lam);
        }
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e1, st));
        // ---- 2. blocked Cholesky, nb = 128 (lower)
        EASE_TRY(dalloc((void**)&Winv, sizeof(double) * (size_t)KB * 128 * 128, pool));
        EASE_TRY(dalloc((void**)&WinvT, sizeof(double) * (size_t)KB * 128 * 128, pool));
        EASE_TRY(dalloc((void**)&d_status, sizeof(int), pool));
        EASE_HIP(hipMemsetAsync(d_status, 0, sizeof(int), st));
        EASE_HIP(hipFuncSetAttribute((const void*)k_potf2_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8));
        for (int k = 0; k < KB; ++k) {
            double* Akk = A + (size_t)k * 128 * np + (size_t)k * 128;
            hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(256), 128 * 129 * 8, st, Akk, (long)np, Winv + (size_t)k * 16384, WinvT + (size_t)k * 16384, d_status);
            const int r = KB - k - 1;
            if (r == 0) break;
            double* A21 = A + (size_t)(k + 1) * 128 * np + (size_t)k * 128;
            // panel: L21 = A21 * inv(L11)^T  (in place: one 128-wide tile column, every workgroup owns its rows)
            EASE_TRY(dgemm(A21, np, Winv + (size_t)k * 16384, 128, r, 1, 8, A21, np, 1.0, 0.0, 0, 0, 0, st));
            // trailing update: A22 -= L21 L21^T  (lower tiles)
            double* A22 = A + (size_t)(k + 1) * 128 * np + (size_t)(k + 1) * 128;
            EASE_TRY(dgemm(A21, np, A21, np, r, r, 8, A22, np, -1.0, 1.0, 1, 0, 0, st));
        }
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e2, st));
        // ---- 3. W = L^-1 (blocked, from the last block column up), then P = W^T W
        EASE_TRY(dalloc((void**)&W, sizeof(double) * (size_t)np * np, pool));
        EASE_TRY(dalloc((void**)&T1T, sizeof(double) * (size_t)128 * np, pool));
        EASE_HIP(hipMemsetAsync(W, 0, sizeof(double) * (size_t)np * np, st));
        for (int k = KB - 1; k >= 0; --k) {
            hipLaunchKernelGGL(k_copy_block, dim3(1), dim3(256), 0, st, Winv + (size_t)k * 16384, W + (size_t)k * 128 * np + (size_t)k * 128, (long)np);
            const int r = KB - k - 1;
            if (r == 0) continue;
            const double* L21 = A + (size_t)(k + 1) * 128 * np + (size_t)k * 128;
            // T1^T [128][r*128] = W11^T-rows x L21-rows:  T1^T[n][m] = sum_j W11[j][n] L21[m][j]
            EASE_TRY(dgemm(WinvT + (size_t)k * 16384, 128, L21, np, 1, r, 8, T1T, np, 1.0, 0.0, 0, 0, 0, st));
            // W21 = -W22 * T1   (W22 lower triangular: row tile tm only needs k < 128 (tm + 1))
            const double* W22 = W + (size_t)(k + 1) * 128 * np + (size_t)(k + 1) * 128;
            double* W21 = W + (size_t)(k + 1) * 128 * np + (size_t)k * 128;
            EASE_TRY(dgemm(W22, np, T1T, np, r, 1, r * 8, W21, np, -1.0, 0.0, 0, 0, 1, st));
        }
        EASE_TRY(dalloc((void**)&WT, sizeof(double) * (size_t)np * np, pool));
        hipLaunchKernelGGL(k_transpose_f64, dim3(np / 64, np / 64), dim3(256), 0, st, W, (long)np, WT, (long)np);
        // P (into A) = W^T W : P[i][j] = sum_{k >= max(i,j)} WT[i][k] WT[j][k]
        EASE_TRY(dgemm(WT, np, WT, np, KB, KB, np / 16, A, np, 1.0, 0.0, 1, 1, 0, st));
        EASE_HIP(hipGetLastError());
        EASE_HIP(hipEventRecord(e3, st));
        // ---- 4. B
        {
            hipError_t e_ = hipMalloc((void**)&h->B, sizeof(double) * (size_t)n * n);
            if (e_ != hipSuccess) { rtx_set_error("ease: hipMalloc(B) failed: %s", hipGetErrorString(e_)); rc = RTX_ENOMEM; goto done; }
        }
        hipLaunchKernelGGL(k_ease_B, dim3((unsigned)(((long)n * n + 255) / 256)), dim3(256), 0, st, A, (long)np, h->B, n);
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
