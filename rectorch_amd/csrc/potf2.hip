// potf2.hip -- the leaf of the EASE solver's recursive Cholesky (ease.hip): one 128x128 diagonal block.
#include "rtx_dgemm.h"

// One 128x128 diagonal block: W = inv(chol(Akk)) written to the diagonal blocks of W (lower) and W^T (upper).
// 256 threads as a 16 x 16 grid; thread (tr, tc) keeps the 8 x 8 elements (tr + 16 i, tc + 16 jx) of the block in
// REGISTERS for the whole kernel.  Every step publishes one column (and, for the inverse, one row) through a
// double-buffered 1-KB LDS line, so a step is one barrier + <= 17 independent LDS reads + <= 36 register FMAs per thread:
//   phase 1  right-looking Cholesky with UNSCALED columns (a_rc keeps a_rc^(j); l_rc = a_rc / sqrt(a_cc))
//   phase 2  columns scaled to L
//   phase 3  in-place inverse: step j folds row j of W into the rows below; a_rc, c <= j, is then the unnormalised
//            row r of W (it replaces the consumed column of L); W[r][c] = a_rc / l_rr is applied on the way out
// The loops over the 16-wide column blocks are unrolled (register indices and the live sub-blocks are compile-time),
// the 16 columns inside a block are a run-time loop.  (A first version with 1024 threads x 16 elements took 208 us:
// its 16 waves issued 17 LDS reads per 16 FMAs and the LDS pipe, not the FMA pipe, set the pace.)
__device__ __forceinline__ double rtx_rcp_f64(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    y = y * (2.0 - x * y);
    return y * (2.0 - x * y);
}

// phase 1, columns 16 JB .. 16 JB + 15.  false: the block is not positive definite
template <int JB>
__device__ __forceinline__ bool potf2_chol_block(double (&a)[8][8], double (*colbuf)[128], double* dinv, int tid, int tr, int tc)
{
#pragma nounroll
    for (int jc = 0; jc < 16; ++jc) {
        const int j = JB * 16 + jc;
        double* cb = colbuf[j & 1];
        if (tc == jc) {
#pragma unroll
            for (int i = JB; i < 8; ++i) {
                const int r = tr + 16 * i;
                if (r >= j) cb[r] = a[i][JB];
            }
        }
        __syncthreads();
        const double ajj = cb[j];
        if (!(ajj > 0.0)) return false;   // uniform: every thread reads the same value
        if (tid == 0) dinv[j] = 1.0 / sqrt(ajj);
        const double rinv = rtx_rcp_f64(ajj);
        double mi[8], cv[8];
#pragma unroll
        for (int i = JB; i < 8; ++i) mi[i] = cb[tr + 16 * i] * rinv;
#pragma unroll
        for (int jx = JB; jx < 8; ++jx) cv[jx] = cb[tc + 16 * jx];
#pragma unroll
        for (int i = JB; i < 8; ++i)
#pragma unroll
            for (int jx = JB; jx <= i; ++jx) {
                // element (r, c) = (tr + 16 i, tc + 16 jx): updated iff j < c <= r
                const bool right = (jx > JB) || (tc > jc);
                const bool lower = (i > jx) || (tc <= tr);
                if (right && lower) a[i][jx] -= mi[i] * cv[jx];
            }
    }
    return true;
}

// phase 3, columns 16 JB .. 16 JB + 15
template <int JB>
__device__ __forceinline__ void potf2_inv_block(double (&a)[8][8], const double (&dr)[8], double (*colbuf)[128], double (*rowbuf)[128], int tr,
                                                int tc)
{
#pragma nounroll
    for (int jc = 0; jc < 16; ++jc) {
        const int j = JB * 16 + jc;
        double* cb = colbuf[j & 1];
        double* rb = rowbuf[j & 1];
        if (tc == jc) {   // column j of L below the diagonal
#pragma unroll
            for (int i = JB; i < 8; ++i) {
                const int r = tr + 16 * i;
                if (r > j) cb[r] = a[i][JB];
            }
        }
        if (tr == jc) {   // row j (= tr + 16 JB) of W, normalised
#pragma unroll
            for (int jx = 0; jx <= JB; ++jx) {
                const int c = tc + 16 * jx;
                if (c < j) rb[c] = a[JB][jx] * dr[JB];
                else if (c == j) rb[c] = dr[JB];
            }
        }
        __syncthreads();
        double wv[8];
#pragma unroll
        for (int jx = 0; jx <= JB; ++jx) wv[jx] = rb[tc + 16 * jx];
#pragma unroll
        for (int i = JB; i < 8; ++i) {
            const int r = tr + 16 * i;
            if (r > j) {
                const double lrj = cb[r];
#pragma unroll
                for (int jx = 0; jx <= JB; ++jx) {
                    const int c = tc + 16 * jx;
                    if (c < j) a[i][jx] -= lrj * wv[jx];
                    else if (c == j) a[i][jx] = -lrj * wv[jx];
                }
            }
        }
    }
}

__global__ __launch_bounds__(256) void k_potf2_inv(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status)
{
    extern __shared__ __attribute__((aligned(16))) double t[];   // [128][129] staging tile
    __shared__ double colbuf[2][128], rowbuf[2][128], dinv[128];
    const int tid = threadIdx.x, tr = tid >> 4, tc = tid & 15;
    for (int e = tid; e < 128 * 128; e += 256) t[(e >> 7) * 129 + (e & 127)] = Akk[(size_t)(e >> 7) * ld + (e & 127)];
    __syncthreads();
    double a[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) a[i][jx] = t[(tr + 16 * i) * 129 + tc + 16 * jx];

    bool ok = potf2_chol_block<0>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<1>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<2>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<3>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<4>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<5>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<6>(a, colbuf, dinv, tid, tr, tc);
    ok = ok && potf2_chol_block<7>(a, colbuf, dinv, tid, tr, tc);
    if (!ok) {
        if (tid == 0) *status = 1;
        return;
    }
    __syncthreads();
    // ---- phase 2: l_rc = a_rc / sqrt(a_cc) for c < r
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
        const double dc = dinv[tc + 16 * jx];
#pragma unroll
        for (int i = jx; i < 8; ++i)
            if ((i > jx) || (tc < tr)) a[i][jx] *= dc;
    }
    double dr[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) dr[i] = dinv[tr + 16 * i];
    potf2_inv_block<0>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<1>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<2>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<3>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<4>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<5>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<6>(a, dr, colbuf, rowbuf, tr, tc);
    potf2_inv_block<7>(a, dr, colbuf, rowbuf, tr, tc);
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int jx = 0; jx < 8; ++jx) {
            const int r = tr + 16 * i, c = tc + 16 * jx;
            t[r * 129 + c] = (c < r) ? a[i][jx] * dr[i] : (c == r ? dr[i] : 0.0);
        }
    __syncthreads();
    for (int e = tid; e < 128 * 128; e += 256) {
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
    hipLaunchKernelGGL(k_potf2_inv, dim3(1), dim3(256), 128 * 129 * 8, stream, Akk, ld, Wkk, WTkk, ldw, status);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
