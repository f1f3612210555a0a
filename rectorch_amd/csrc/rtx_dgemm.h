// rtx_dgemm.h -- f64 MFMA NT GEMM used by the EASE closed-form solver (ease.hip):
//
//   C[m][n] = beta * C[m][n] + alpha * sum_{k0(m,n) <= k < k1(m,n)} A[m][k] * B[n][k]
//
// v_mfma_f64_16x16x4_f64; 128x128 tile (or 64x64 for the small nodes), 4 waves, 64x64 of C per wave as 4x4 blocks of 16x16; K consumed in
// 128-byte slices (16 doubles) through the same global->VGPR->LDS staging as the bf16/f32 GEMM (144-B padded rows,
// conflict-free ds_read_b64 fragment reads).  All dimensions are multiples of 128 (the solver pads the Gram matrix
// with an identity block), so there are no bounds checks.
#pragma once
#include "rtx_common.h"

enum { RTX_DK_ALL = 0, RTX_DK_TM = 1, RTX_DK_TN = 2, RTX_DK_MAX = 3 };

struct RtxDgemm {
    const double* A;   // [M][lda]
    const double* B;   // [N][ldb]
    long lda, ldb;
    int m_tiles, n_tiles;
    int k_slices;      // K / 16
    double* C;
    long ldc;
    double* CT;        // nullable: the result is also stored transposed, CT[n][m] (ldct)
    long ldct;
    double alpha, beta;
    int lower_only;    // skip tiles strictly above the diagonal (tn > tm)
    // triangular operands: the K range of tile (tm, tn) starts at 128 * {0, tm, tn, max(tm, tn)}  (k_lo) and ends
    // at 128 * ({tm, tn} + 1) (k_hi: RTX_DK_TM / RTX_DK_TN) instead of covering all of K
    int k_lo, k_hi;
    int small_tile;    // 1: 64x64 workgroup tiles (m_tiles / n_tiles and the tile-relative K ranges count 64s)
};

int rtx_dgemm_launch(const RtxDgemm& g, hipStream_t stream);

// leaf of the recursive Cholesky: W = inv(chol(Akk)) of one 128x128 block into Wkk (lower) and WTkk (upper), both with
// leading dimension ldw; *status = 1 if the block is not positive definite (potf2.hip)
int rtx_potf2_inv_launch(const double* Akk, long ld, double* Wkk, double* WTkk, long ldw, int* status, hipStream_t stream);
// measurement knob: 1 (default) = the blocked leaf (four 32-column panels, ~30 barriers), 0 = one barrier per column (round 1)
void rtx_potf2_set_blocked(int on);
void rtx_potf2_set_stamps(unsigned long long* dev);   // measurement: >= 16 device entries receive 100-MHz clock stamps of the blocked leaf's phases
