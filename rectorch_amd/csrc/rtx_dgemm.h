// rtx_dgemm.h -- f64 MFMA NT GEMM used by the EASE closed-form solver (ease.hip):
//
//   C[m][n] = beta * C[m][n] + alpha * sum_{k >= k0(m,n)} A[m][k] * B[n][k]
//
// v_mfma_f64_16x16x4_f64; 128x128 tile, 4 waves, 64x64 of C per wave as 4x4 blocks of 16x16; K consumed in
// 128-byte slices (16 doubles) through the same global->VGPR->LDS staging as the bf16/f32 GEMM (144-B padded rows,
// conflict-free ds_read_b64 fragment reads).  All dimensions are multiples of 128 (the solver pads the Gram matrix
// with an identity block), so there are no bounds checks.
#pragma once
#include "rtx_common.h"

struct RtxDgemm {
    const double* A;   // [M][lda]
    const double* B;   // [N][ldb]
    long lda, ldb;
    int m_tiles, n_tiles;
    int k_slices;      // K / 16
    double* C;
    long ldc;
    double alpha, beta;
    int lower_only;    // skip tiles strictly above the diagonal (tn > tm)
    int k_from_tile;   // 1: the sum starts at k = 128 * max(tm, tn) (products of lower-triangular factors)
    int k_to_tile;     // 1: the sum ends at k = 128 * (tm + 1) (A lower triangular: nothing beyond its diagonal block)
};

int rtx_dgemm_launch(const RtxDgemm& g, hipStream_t stream);
