// dgemm.hip -- f64 MFMA NT GEMM for the EASE solver (see rtx_dgemm.h).
#include "rtx_dgemm.h"

typedef __attribute__((ext_vector_type(4))) double f64x4_t;

#define RTX_DROW 144                   // 128 B of K (16 doubles) + 16 B pad

// BLK = 16x16 blocks per wave and dimension: 4 -> 128x128 workgroup tile (64x64 per wave), 2 -> 64x64 tile (32x32 per
// wave).  The small tile serves the deep nodes of the EASE recursion: a 128-wide node is ONE 128x128 tile, i.e. one
// compute unit's f64 pipe for the whole product; as four 64x64 tiles it runs on four.
template <int BLK>
__global__ __launch_bounds__(256, 2) void rtx_dgemm_nt(const RtxDgemm p)
{
    constexpr int TILE = 32 * BLK;                 // tile rows = tile columns
    constexpr int STAGE = 2 * TILE * RTX_DROW;     // A rows | B rows of one stage
    constexpr int SPT = TILE / 16;                 // K slices (16 doubles) per tile of K
    constexpr int NLD = TILE / 32;                 // staging loads per operand and thread (32 rows per pass)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 15, kq = lane >> 4;
    const int tn = blockIdx.x, tm = blockIdx.y;
    if (p.lower_only && tn > tm) return;

    int ks0 = 0, ks1 = p.k_slices;   // TILE rows of K per tile = SPT slices of 16 doubles
    if (p.k_lo == RTX_DK_TM) ks0 = SPT * tm;
    else if (p.k_lo == RTX_DK_TN) ks0 = SPT * tn;
    else if (p.k_lo == RTX_DK_MAX) ks0 = SPT * max(tm, tn);
    if (p.k_hi == RTX_DK_TM) ks1 = min(ks1, SPT * (tm + 1));
    else if (p.k_hi == RTX_DK_TN) ks1 = min(ks1, SPT * (tn + 1));
    const int nk = ks1 - ks0;

    const size_t rowA = (size_t)p.lda * 8, rowB = (size_t)p.ldb * 8;
    const int st_row = tid >> 3, st_ch = tid & 7;
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * TILE + st_row) * rowA + st_ch * 16;
    const unsigned char* gB = (const unsigned char*)p.B + ((size_t)tn * TILE + st_row) * rowB + st_ch * 16;
    const int lds_a = st_row * RTX_DROW + st_ch * 16;
    const int lds_b = (TILE + st_row) * RTX_DROW + st_ch * 16;

    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    f64x4_t acc[BLK][BLK];
#pragma unroll
    for (int i = 0; i < BLK; ++i)
#pragma unroll
        for (int j = 0; j < BLK; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

#define RTX_DGL(X, q, base, row, ks) r##X##q = *(const uint4*)((base) + (size_t)(q) * 32 * (row) + (size_t)(ks) * 128);
#define RTX_DGLOAD(ks)                                                                                   \
    RTX_DGL(a, 0, gA, rowA, ks) RTX_DGL(a, 1, gA, rowA, ks)                                              \
    if (NLD == 4) { RTX_DGL(a, 2, gA, rowA, ks) RTX_DGL(a, 3, gA, rowA, ks) }                            \
    RTX_DGL(b, 0, gB, rowB, ks) RTX_DGL(b, 1, gB, rowB, ks)                                              \
    if (NLD == 4) { RTX_DGL(b, 2, gB, rowB, ks) RTX_DGL(b, 3, gB, rowB, ks) }
#define RTX_DLS(X, q, off, st) *(uint4*)(smem + (st) * STAGE + (off) + (q) * 32 * RTX_DROW) = r##X##q;
#define RTX_DLSTORE(st)                                                                                  \
    RTX_DLS(a, 0, lds_a, st) RTX_DLS(a, 1, lds_a, st)                                                    \
    if (NLD == 4) { RTX_DLS(a, 2, lds_a, st) RTX_DLS(a, 3, lds_a, st) }                                  \
    RTX_DLS(b, 0, lds_b, st) RTX_DLS(b, 1, lds_b, st)                                                    \
    if (NLD == 4) { RTX_DLS(b, 2, lds_b, st) RTX_DLS(b, 3, lds_b, st) }

    if (nk > 0) {
        RTX_DGLOAD(ks0)
        RTX_DLSTORE(0)
    }
    __syncthreads();
    for (int t = 0; t < nk; ++t) {
        const int st = t & 1;
        if (t + 1 < nk) { RTX_DGLOAD(ks0 + t + 1) }
        // lane l supplies A[i = l & 15][k = l >> 4] and B[k = l >> 4][j = l & 15] of each 16x16x4 product
        const unsigned char* sA = smem + st * STAGE + (wm * (16 * BLK) + li) * RTX_DROW + kq * 8;
        const unsigned char* sB = smem + st * STAGE + (TILE + wn * (16 * BLK) + li) * RTX_DROW + kq * 8;
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            double a[BLK], b[BLK];
#pragma unroll
            for (int i = 0; i < BLK; ++i) {
                a[i] = *(const double*)(sA + i * 16 * RTX_DROW + s4 * 32);
                b[i] = *(const double*)(sB + i * 16 * RTX_DROW + s4 * 32);
            }
#pragma unroll
            for (int i = 0; i < BLK; ++i)
#pragma unroll
                for (int j = 0; j < BLK; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nk) {
            if (st) { RTX_DLSTORE(0) } else { RTX_DLSTORE(1) }
        }
        __syncthreads();
    }
#undef RTX_DGL
#undef RTX_DGLOAD
#undef RTX_DLS
#undef RTX_DLSTORE

    // C/D layout of v_mfma_f64_16x16x4_f64: col = lane & 15, row = (lane >> 4) + 4 * reg   (NOT the f32 map)
    double* cp = p.C + ((size_t)tm * TILE + wm * (16 * BLK) + kq) * p.ldc + (size_t)tn * TILE + wn * (16 * BLK) + li;
#pragma unroll
    for (int i = 0; i < BLK; ++i)
#pragma unroll
        for (int j = 0; j < BLK; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                double* dst = cp + (size_t)(i * 16 + 4 * e) * p.ldc + j * 16;
                double v = p.alpha * acc[i][j][e];
                if (p.beta != 0.0) v += p.beta * (*dst);
                *dst = v;
                acc[i][j][e] = v;
            }
    if (p.CT) {   // transposed copy: 4 consecutive rows of C per register quad -> 32-byte runs along CT's rows
        double* ct = p.CT + ((size_t)tn * TILE + wn * (16 * BLK) + li) * p.ldct + (size_t)tm * TILE + wm * (16 * BLK) + kq;
#pragma unroll
        for (int i = 0; i < BLK; ++i)
#pragma unroll
            for (int j = 0; j < BLK; ++j)
#pragma unroll
                for (int e = 0; e < 4; ++e) ct[(size_t)(j * 16) * p.ldct + i * 16 + 4 * e] = acc[i][j][e];
    }
}

int rtx_dgemm_launch(const RtxDgemm& g, hipStream_t stream)
{
    RTX_CHECK(g.m_tiles > 0 && g.n_tiles > 0 && g.k_slices >= 0, RTX_EINVAL, "dgemm: empty problem");
    if (g.small_tile)   // m_tiles / n_tiles count 64x64 tiles
        hipLaunchKernelGGL(rtx_dgemm_nt<2>, dim3(g.n_tiles, g.m_tiles), dim3(256), 0, stream, g);
    else
        hipLaunchKernelGGL(rtx_dgemm_nt<4>, dim3(g.n_tiles, g.m_tiles), dim3(256), 0, stream, g);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
