// gemm_f32.hip -- float32 MFMA GEMM (parity mode) for the operand forms that read a K-MAJOR matrix (gfx950).
//
//   NN : C[m][n] = sum_k A[m][k] * B[k][n]      backward-data:   dOut[B,out] x W[out,in]
//   TN : C[m][n] = sum_k A[k][m] * B[k][n]      weight gradient: dOut[B,out]^T x Act[B,in]
//
// (NT, both operands K-contiguous, stays in gemm.hip.)  v_mfma_f32_32x32x2_f32: exact f32 products and sums, the
// arithmetic the 1e-5 logits criterion is measured on.  With 4-byte elements a K-major operand needs no transposing
// read: the MFMA A / B operand of lane (i = lane & 31, k = lane >> 5) is ONE float, and in a [k][m] image the 32 lanes
// of a half-wave read 32 consecutive floats of one k-row -- a conflict-free ds_read_b32.
//
// 128 x 128 tile, 4 waves (64 x 64 of C each as 2 x 2 accumulators), K slices of 32, global -> registers -> LDS staging
// with the next slice's loads in flight under the current slice's MFMAs, one barrier per slice.  This mode is bound by
// the f32 MFMA rate (1/16 of bf16), so the staging is kept simple.
//   K-contiguous operand image: [128 rows][32 k], 144-byte rows (16-byte pad: conflict-free ds_read_b128 of 4 k)
//   K-major operand image     : [32 k][128 cols], 512-byte rows
// Both operands must agree on which k an MFMA step multiplies: step (kk8, e) of lane-half g uses k = kk8*8 + g*4 + e.
#include "rtx_gemm.h"

typedef __attribute__((ext_vector_type(16))) float gf_f32x16;
typedef __attribute__((ext_vector_type(4))) float gf_f32x4;

#define GF_ROW 144

template <int AKM /* A is K-major */, int EPI>
__global__ __launch_bounds__(256, 2) void rtx_gemm_f32_km(const RtxGemm p)
{
    // B is always K-major here (NN and TN); A is K-contiguous (NN) or K-major (TN)
    constexpr int ASZ = AKM ? 32 * 512 : 128 * GF_ROW, BSZ = 32 * 512, STAGE = ASZ + BSZ;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * STAGE (68 KB in the NN form: dynamic)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, g = lane >> 5;

    int tm, tn, split;
    bool tail = false;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        if (p.tail_splits > 1 && L >= p.tail_block0) {
            // an extra workgroup of the last partial wave: tile t of the tail region, K piece `split`
            tail = true;
            const int idx = L - p.tail_block0, t = idx / p.tail_splits;
            split = idx - t * p.tail_splits;
            if (p.m_tiles > p.n_tiles) { tm = p.tail_t0 + t / p.n_tiles; tn = t % p.n_tiles; if (tm >= p.m_tiles) return; }
            else { tn = p.tail_t0 + t / p.m_tiles; tm = t % p.m_tiles; if (tn >= p.n_tiles) return; }
        } else if (p.splits > 1) {
            const int tiles = p.m_tiles * p.n_tiles;
            split = xcd + 8 * (j / tiles);
            if (split >= p.splits) return;
            const int t = j % tiles;
            tm = t % p.m_tiles;
            tn = t / p.m_tiles;
        } else if (p.m_tiles <= p.n_tiles) {
            split = 0;
            tn = xcd + 8 * (j / p.m_tiles);
            tm = j % p.m_tiles;
            if (tn >= p.n_tiles) return;
        } else {
            split = 0;
            tm = xcd + 8 * (j / p.n_tiles);
            tn = j % p.n_tiles;
            if (tm >= p.m_tiles) return;
        }
        if (!tail && p.tail_splits > 1 && (p.m_tiles > p.n_tiles ? tm : tn) >= p.tail_t0) return;   // the tail's tiles belong to the extra workgroups
    }
    const int per = tail ? (p.k_slices + p.tail_splits - 1) / p.tail_splits : (p.k_slices + p.splits - 1) / p.splits;
    const int ks0 = split * per;
    const int nk = min(ks0 + per, p.k_slices) - ks0;

    // staging maps.  K-contiguous: thread -> (row = tid >> 3 (+32 per pass), 16-byte chunk tid & 7), 4 passes.
    //                K-major     : thread -> (k-row = tid >> 5 (+8 per pass), 16-byte chunk tid & 31), 4 passes.
    const size_t rowA = (size_t)p.lda * 4, rowB = (size_t)p.ldb * 4;
    const unsigned char* gA = AKM ? (const unsigned char*)p.A + (size_t)(tid >> 5) * rowA + (size_t)tm * 512 + (tid & 31) * 16
                                  : (const unsigned char*)p.A + ((size_t)tm * 128 + (tid >> 3)) * rowA + (tid & 7) * 16;
    const unsigned char* gB = (const unsigned char*)p.B + (size_t)(tid >> 5) * rowB + (size_t)tn * 512 + (tid & 31) * 16;
    const int lds_a = AKM ? (tid >> 5) * 512 + (tid & 31) * 16 : (tid >> 3) * GF_ROW + (tid & 7) * 16;
    const int lds_b = ASZ + (tid >> 5) * 512 + (tid & 31) * 16;
    constexpr int A_PASS_G = AKM ? 8 : 32;                 // rows per staging pass (global rows: k-rows or tile rows)
    constexpr int A_PASS_L = AKM ? 8 * 512 : 32 * GF_ROW;  // LDS bytes per staging pass

    gf_f32x4 ra[4], rb[4];
    auto gload = [&](int t) __attribute__((always_inline)) {
        const size_t ka = AKM ? (size_t)(ks0 + t) * 32 * rowA : (size_t)(ks0 + t) * 128;
        const size_t kb = (size_t)(ks0 + t) * 32 * rowB;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            ra[q] = *(const gf_f32x4*)(gA + ka + (size_t)q * A_PASS_G * rowA);
            rb[q] = *(const gf_f32x4*)(gB + kb + (size_t)q * 8 * rowB);
        }
    };
    auto lstore = [&](int st) __attribute__((always_inline)) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            *(gf_f32x4*)(smem + st * STAGE + lds_a + q * A_PASS_L) = ra[q];
            *(gf_f32x4*)(smem + st * STAGE + lds_b + q * 8 * 512) = rb[q];
        }
    };

    gf_f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto compute = [&](int st) __attribute__((always_inline)) {
        const unsigned char* sA = smem + st * STAGE;
        const unsigned char* sB = smem + st * STAGE + ASZ;
#pragma unroll
        for (int kk8 = 0; kk8 < 4; ++kk8) {
            float a[2][4], b[2][4];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (AKM) {
#pragma unroll
                    for (int e = 0; e < 4; ++e) a[i][e] = *(const float*)(sA + (kk8 * 8 + g * 4 + e) * 512 + (wm * 64 + i * 32 + r) * 4);
                } else {
                    const gf_f32x4 t = *(const gf_f32x4*)(sA + (wm * 64 + i * 32 + r) * GF_ROW + kk8 * 32 + g * 16);
                    a[i][0] = t[0]; a[i][1] = t[1]; a[i][2] = t[2]; a[i][3] = t[3];
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) b[j][e] = *(const float*)(sB + (kk8 * 8 + g * 4 + e) * 512 + (wn * 64 + j * 32 + r) * 4);
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i][e], b[j][e], acc[i][j], 0, 0, 0);
        }
    };

    if (nk > 0) {
        gload(0);
        lstore(0);
        __syncthreads();
        for (int t = 0; t < nk; ++t) {
            const int st = t & 1;
            if (t + 1 < nk) gload(t + 1);
            compute(st);
            if (t + 1 < nk) lstore(1 - st);
            __syncthreads();
        }
    }

    if (EPI == RTX_EPI_GRAD && tail) {
        // partial sums of a tail tile, tile-local coordinates of the tail region
        const int lr = (p.m_tiles > p.n_tiles ? (tm - p.tail_t0) : tm) * 128 + wm * 64 + 4 * g;
        const int lc = (p.m_tiles > p.n_tiles ? tn : (tn - p.tail_t0)) * 128 + wn * 64 + r;
        float* tp = p.tail_C + (size_t)split * p.tail_slab_stride + (size_t)lr * p.tail_ldc + lc;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) tp[(long)(i * 32 + (e & 3) + 8 * (e >> 2)) * p.tail_ldc + j * 32] = acc[i][j][e];
        return;
    }
    // epilogue (as gemm.hip): C/D layout col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int row_base = tm * 128 + wm * 64 + 4 * g;
    const int col_base = tn * 128 + wn * 64 + r;
    const long ld = (EPI == RTX_EPI_GRAD) ? (long)p.N_real : p.ldc;
    float* cp = p.C + (EPI == RTX_EPI_STORE ? (size_t)split * p.slab_stride : (size_t)0) + (size_t)row_base * ld + col_base;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col_base + j * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dr = i * 32 + (e & 3) + 8 * (e >> 2);
                const int row = row_base + dr;
                const float v = acc[i][j][e];
                float* dst = cp + (long)dr * ld + j * 32;
                if (EPI == RTX_EPI_STORE) {
                    *dst = v;
                } else {   // RTX_EPI_GRAD
                    if (row < p.M_real) {
                        if (col < p.N_real) *dst = v;
                        else if (col == p.N_real && p.gbias) p.gbias[row] = v;
                    }
                }
            }
        }
}

// float32 operands, 128 x 128 tiles (g.m_tiles = M_pad / 128, g.n_tiles = N_pad / 128), g.k_slices = K_pad / 32.
// g.form RTX_FORM_NN (A [M_pad][lda]) or RTX_FORM_TN (A [K_pad][lda]); B [K_pad][ldb].  RTX_EPI_STORE (split-K slabs) / RTX_EPI_GRAD.
int rtx_gemm_f32_km_launch(const RtxGemm& g, int epilogue, hipStream_t stream)
{
    RTX_CHECK(g.form == RTX_FORM_NN || g.form == RTX_FORM_TN, RTX_EINVAL, "gemm_f32_km: form %d not supported", g.form);
    RTX_CHECK(epilogue == RTX_EPI_STORE || epilogue == RTX_EPI_GRAD, RTX_EINVAL, "gemm_f32_km: bad epilogue %d", epilogue);
    RTX_CHECK(g.m_tiles > 0 && g.n_tiles > 0 && g.k_slices > 0 && g.splits > 0, RTX_EINVAL, "gemm_f32_km: empty problem");
    RTX_CHECK(epilogue == RTX_EPI_STORE || g.splits == 1, RTX_EINVAL, "gemm_f32_km: split-K only with EPI_STORE");
    RTX_CHECK((g.splits - 1) * ((g.k_slices + g.splits - 1) / g.splits) < g.k_slices, RTX_EINVAL, "gemm_f32_km: %d splits leave an empty split of %d slices",
              g.splits, g.k_slices);
    const int tiles = g.m_tiles * g.n_tiles;
    int groups, gsize;
    if (g.splits > 1) { groups = g.splits; gsize = tiles; }
    else if (g.m_tiles <= g.n_tiles) { groups = g.n_tiles; gsize = g.m_tiles; }
    else { groups = g.m_tiles; gsize = g.n_tiles; }
    unsigned n_blocks = (unsigned)(8 * ((groups + 7) / 8) * gsize);
    RtxGemm gt = g;
    if (g.tail_splits > 1) {
        RTX_CHECK(g.splits == 1 && epilogue == RTX_EPI_GRAD && g.tail_C && g.tail_t0 >= 0 && g.tail_t0 < groups, RTX_EINVAL,
                  "gemm_f32_km: the tail split needs a single-pass gradient product and a tail start inside the long tile dimension");
        RTX_CHECK((g.tail_splits - 1) * ((g.k_slices + g.tail_splits - 1) / g.tail_splits) < g.k_slices, RTX_EINVAL,
                  "gemm_f32_km: %d tail splits leave an empty split of %d slices", g.tail_splits, g.k_slices);
        gt.tail_block0 = (int)n_blocks;
        n_blocks += (unsigned)((groups - g.tail_t0) * gsize * g.tail_splits);
    }
    const RtxGemm& gg = gt;
    const dim3 grid(n_blocks), block(256);
    constexpr int LDS_NN = 2 * (128 * GF_ROW + 32 * 512), LDS_TN = 2 * (32 * 512 + 32 * 512);
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_gemm_f32_km<0, RTX_EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_NN));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_gemm_f32_km<0, RTX_EPI_GRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_NN));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_gemm_f32_km<1, RTX_EPI_STORE>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TN));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_gemm_f32_km<1, RTX_EPI_GRAD>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_TN));
        configured = true;
    }
    if (g.form == RTX_FORM_NN) {
        if (epilogue == RTX_EPI_STORE) hipLaunchKernelGGL((rtx_gemm_f32_km<0, RTX_EPI_STORE>), grid, block, LDS_NN, stream, gg);
        else hipLaunchKernelGGL((rtx_gemm_f32_km<0, RTX_EPI_GRAD>), grid, block, LDS_NN, stream, gg);
    } else {
        if (epilogue == RTX_EPI_STORE) hipLaunchKernelGGL((rtx_gemm_f32_km<1, RTX_EPI_STORE>), grid, block, LDS_TN, stream, gg);
        else hipLaunchKernelGGL((rtx_gemm_f32_km<1, RTX_EPI_GRAD>), grid, block, LDS_TN, stream, gg);
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
