// gemm.hip -- MFMA NT GEMM for gfx950 (see rtx_gemm.h for the design notes).
#include "rtx_gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

#define RTX_LDS_ROW 144                  // 128 B of K + 16 B pad
#define RTX_LDS_TILE (128 * RTX_LDS_ROW) // one operand, one stage

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
    // one ds_read_b128 per operand = 8 bf16 = the whole K=16 fragment of v_mfma_f32_32x32x16_bf16
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    }
};

template <> struct Mma<float> {
    // 16 B = 4 consecutive k of this lane's row; lanes 0-31 hold k = 0..3, lanes 32-63 k = 4..7 of the
    // 8-float sub-slice.  v_mfma_f32_32x32x2_f32 pairs element e of both half-waves; A and B use the same
    // (lane-half, e) -> k map, so the products are summed over all 8 k.  Exact f32 (parity mode).
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.x), __builtin_bit_cast(float, b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.y), __builtin_bit_cast(float, b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.z), __builtin_bit_cast(float, b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__builtin_bit_cast(float, a.w), __builtin_bit_cast(float, b.w), acc, 0, 0, 0);
    }
};

template <typename T, int EPI>
__global__ __launch_bounds__(256, 2) void rtx_gemm_nt(const RtxGemm p)
{
    // [stage][operand] tiles; all LDS in one array (cdna guide: a second __shared__ object de-pipelines)
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * RTX_LDS_TILE];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, g = lane >> 5;

    int tm, tn;
    if (p.n_major) {
        tn = blockIdx.x % p.n_tiles;
        tm = blockIdx.x / p.n_tiles;
    } else {
        tm = blockIdx.x % p.m_tiles;
        tn = blockIdx.x / p.m_tiles;
    }
    const int split = blockIdx.y;
    const int per = (p.k_slices + p.splits - 1) / p.splits;
    const int ks0 = split * per;
    const int ks1 = min(ks0 + per, p.k_slices);
    const int nk = ks1 - ks0;

    const size_t rowA = (size_t)p.lda * sizeof(T), rowB = (size_t)p.ldb * sizeof(T);
    const int st_row = tid >> 3, st_ch = tid & 7;  // staging: 8 lanes x 16 B = one 128-B row slice
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * 128 + st_row) * rowA + st_ch * 16;
    const unsigned char* gB = (const unsigned char*)p.B + ((size_t)tn * 128 + st_row) * rowB + st_ch * 16;
    const int lds_st = st_row * RTX_LDS_ROW + st_ch * 16;

    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;  // named (not an array): keeps the prefetch in VGPRs
    f32x16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define RTX_GLOAD1(q, ks)                                                                          \
    ra##q = *(const uint4*)(gA + (size_t)(q) * 32 * rowA + (size_t)(ks) * 128);                    \
    rb##q = *(const uint4*)(gB + (size_t)(q) * 32 * rowB + (size_t)(ks) * 128);
#define RTX_GLOAD(ks) RTX_GLOAD1(0, ks) RTX_GLOAD1(1, ks) RTX_GLOAD1(2, ks) RTX_GLOAD1(3, ks)
#define RTX_LSTORE1(q, s)                                                                          \
    *(uint4*)(smem + (2 * (s)) * RTX_LDS_TILE + lds_st + (q) * 32 * RTX_LDS_ROW) = ra##q;          \
    *(uint4*)(smem + (2 * (s) + 1) * RTX_LDS_TILE + lds_st + (q) * 32 * RTX_LDS_ROW) = rb##q;
#define RTX_LSTORE(s) RTX_LSTORE1(0, s) RTX_LSTORE1(1, s) RTX_LSTORE1(2, s) RTX_LSTORE1(3, s)

    if (nk > 0) {
        RTX_GLOAD(ks0);
        RTX_LSTORE(0);
    }
    __syncthreads();

    for (int t = 0; t < nk; ++t) {
        const int s = t & 1;
        if (t + 1 < nk) { RTX_GLOAD(ks0 + t + 1); }
        const unsigned char* sA = smem + (2 * s) * RTX_LDS_TILE + (wm * 64 + r) * RTX_LDS_ROW + g * 16;
        const unsigned char* sB = smem + (2 * s + 1) * RTX_LDS_TILE + (wn * 64 + r) * RTX_LDS_ROW + g * 16;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
            const uint4 a0 = *(const uint4*)(sA + kk * 32);
            const uint4 a1 = *(const uint4*)(sA + 32 * RTX_LDS_ROW + kk * 32);
            const uint4 b0 = *(const uint4*)(sB + kk * 32);
            const uint4 b1 = *(const uint4*)(sB + 32 * RTX_LDS_ROW + kk * 32);
            Mma<T>::run(acc[0][0], a0, b0);
            Mma<T>::run(acc[0][1], a0, b1);
            Mma<T>::run(acc[1][0], a1, b0);
            Mma<T>::run(acc[1][1], a1, b1);
        }
        if (t + 1 < nk) { RTX_LSTORE(s ^ 1); }
        __syncthreads();
    }
#undef RTX_GLOAD
#undef RTX_GLOAD1
#undef RTX_LSTORE
#undef RTX_LSTORE1

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    const int row_base = tm * 128 + wm * 64 + 4 * g;
    const int col_base = tn * 128 + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = col_base + j * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = row_base + i * 32 + (e & 3) + 8 * (e >> 2);
                const float v = acc[i][j][e];
                if (EPI == RTX_EPI_STORE) {
                    p.C[(size_t)split * p.slab_stride + (size_t)row * p.ldc + col] = v;
                } else if (EPI == RTX_EPI_BIAS_ROWS) {
                    if (row < p.M_real && col < p.N_real)
                        p.C[(size_t)row * p.ldc + col] = v + (p.bias ? p.bias[col] : 0.f);
                } else {  // RTX_EPI_GRAD
                    if (row < p.M_real) {
                        if (col < p.N_real)
                            p.C[(size_t)row * p.N_real + col] = v;
                        else if (col == p.N_real && p.gbias)
                            p.gbias[row] = v;
                    }
                }
            }
        }
    }
}

int rtx_gemm_launch(const RtxGemm& g, int is_bf16, int epilogue, hipStream_t stream)
{
    RTX_CHECK(g.m_tiles > 0 && g.n_tiles > 0 && g.k_slices > 0 && g.splits > 0, RTX_EINVAL, "gemm: empty problem");
    RTX_CHECK(epilogue == RTX_EPI_STORE || g.splits == 1, RTX_EINVAL, "gemm: split-K only with EPI_STORE");
    RTX_CHECK(g.splits <= 65535, RTX_EINVAL, "gemm: too many splits");
    const dim3 grid((unsigned)(g.m_tiles * g.n_tiles), (unsigned)g.splits), block(256);
#define RTX_LAUNCH(T, E) hipLaunchKernelGGL((rtx_gemm_nt<T, E>), grid, block, 0, stream, g)
    if (is_bf16) {
        switch (epilogue) {
        case RTX_EPI_STORE: RTX_LAUNCH(bf16_t, RTX_EPI_STORE); break;
        case RTX_EPI_BIAS_ROWS: RTX_LAUNCH(bf16_t, RTX_EPI_BIAS_ROWS); break;
        case RTX_EPI_GRAD: RTX_LAUNCH(bf16_t, RTX_EPI_GRAD); break;
        default: RTX_CHECK(false, RTX_EINVAL, "gemm: bad epilogue %d", epilogue);
        }
    } else {
        switch (epilogue) {
        case RTX_EPI_STORE: RTX_LAUNCH(float, RTX_EPI_STORE); break;
        case RTX_EPI_BIAS_ROWS: RTX_LAUNCH(float, RTX_EPI_BIAS_ROWS); break;
        case RTX_EPI_GRAD: RTX_LAUNCH(float, RTX_EPI_GRAD); break;
        default: RTX_CHECK(false, RTX_EINVAL, "gemm: bad epilogue %d", epilogue);
        }
    }
#undef RTX_LAUNCH
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
