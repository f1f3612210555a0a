// potf2.hip -- the leaf of the EASE solver's recursive Cholesky (ease.hip): one 128x128 diagonal block.
// Kept in its own translation unit: the fully unrolled kernel takes minutes to compile.
#include "rtx_dgemm.h"

// One 128x128 diagonal block: W = inv(chol(Akk)) written to the diagonal blocks of W (lower) and W^T (upper).
// 1024 threads = (row r, column phase q); thread (r, q) keeps a[k] = element (r, q + 8k) of the block in REGISTERS for
// the whole kernel.  Every step publishes one column (and, for the inverse, one row) through a double-buffered 1-KB
// LDS line, so a step is one barrier + one round of independent LDS reads + <= 16 register FMAs:
//   phase 1  right-looking Cholesky with UNSCALED columns (a_rc keeps a_rc^(j); l_rc = a_rc / sqrt(a_cc))
//   phase 2  columns scaled to L
//   phase 3  in-place inverse: step j folds row j of W into the rows below; a_rc, c <= j, is then the unnormalised
//            row r of W (it replaces the consumed column of L); W[r][c] = a_rc / l_rr is applied on the way out
// Both j loops are fully unrolled: register indices and the live range of k are compile-time.
__device__ __forceinline__ double rtx_rcp_f64(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    return y * (2.0 - x * y);
}

__global__ __launch_bounds__(1024) void k_potf2_inv(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status)
{
    extern __shared__ __attribute__((aligned(16))) double t[];   // [128][129] staging tile
    __shared__ double colbuf[2][128], rowbuf[2][128], dinv[128];
    const int tid = threadIdx.x, r = tid & 127, q = tid >> 7;
    for (int e = tid; e < 128 * 128; e += 1024) t[(e >> 7) * 129 + (e & 127)] = Akk[(size_t)(e >> 7) * ld + (e & 127)];
    __syncthreads();
    double a[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) a[k] = t[r * 129 + q + 8 * k];

#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = kb * 8 + jj;
            double* cb = colbuf[j & 1];
            if (q == jj && r >= j) cb[r] = a[kb];
            __syncthreads();
            const double ajj = cb[j];
            if (!(ajj > 0.0)) {   // uniform: every thread reads the same value
                if (tid == 0) *status = 1;
                return;
            }
            if (q == jj && r == j) dinv[j] = 1.0 / sqrt(ajj);
            if (r > j) {
                const double m = cb[r] * rtx_rcp_f64(ajj);
#pragma unroll
                for (int k = kb; k < 16; ++k) {
                    const int c = q + 8 * k;
                    if (c > j && c <= r) a[k] -= m * cb[c];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = q + 8 * k;
        if (c < r) a[k] *= dinv[c];   // l_rc = a_rc / sqrt(a_cc)
    }
    const double dr = dinv[r];
#pragma unroll
    for (int kb = 0; kb < 16; ++kb) {
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int j = kb * 8 + jj;
            double* cb = colbuf[j & 1];
            double* rb = rowbuf[j & 1];
            if (q == jj && r > j) cb[r] = a[kb];
            if (r == j) {
#pragma unroll
                for (int k = 0; k <= kb; ++k) {
                    const int c = q + 8 * k;
                    if (c < j) rb[c] = a[k] * dr;
                    else if (c == j) rb[c] = dr;
                }
            }
            __syncthreads();
            if (r > j) {
                const double lrj = cb[r];
#pragma unroll
                for (int k = 0; k <= kb; ++k) {
                    const int c = q + 8 * k;
                    if (c < j) a[k] -= lrj * rb[c];
                    else if (c == j) a[k] = -lrj * rb[c];
                }
            }
        }
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = q + 8 * k;
        t[r * 129 + c] = (c < r) ? a[k] * dr : (c == r ? dr : 0.0);
    }
    __syncthreads();
    for (int e = tid; e < 128 * 128; e += 1024) {
        const int i = e >> 7, j = e & 127;
        Wkk[(size_t)i * ldw + j] = t[i * 129 + j];
        WTkk[(size_t)i * ldw + j] = t[j * 129 + i];
    }
}

int rtx_potf2_inv_launch(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status, hipStream_t stream)
{
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)k_potf2_inv, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 129 * 8));
        configured = true;
    }
    hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(1024), 128 * 129 * 8, stream, Akk, ld, Wkk, WTkk, ldw, status);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
