// syrk.hip -- Gram matrix G = X^T X (lower triangle) of the EASE solver on fp8 / bf16 MFMA.
//
//   C[m][n] = sum_k A[m][k] * A[n][k],  n-tile <= m-tile,   A = X^T as a dense K(=users)-contiguous matrix
//
// K is the number of users (1e5..1e6): every tile walks a very long K, so the kernel is all main loop.  The general
// GEMM of this library (gemm.hip, 64x64 of C per wave) is bound by LDS bandwidth there: a wave reads 16 KB of LDS per
// 128-byte K slice for 32 MFMAs.  Here each wave owns 128x64 of C (4x2 MFMA 32x32 accumulators, 128 VGPRs): 24 KB of LDS
// reads per slice feed 64 MFMAs -- 1.33x the flops per LDS byte -- and eight waves share a 256x256 workgroup tile whose
// operand slices go global -> LDS by DMA.  The ladder that led here (register-staged 256x128 / 4 waves: 45 ms on the
// ml-20m Gram; LDS-DMA ring, 4 waves: 42; 8 waves: 36; the K = 64 fp8 instruction: 22.8) is in DESIGN.md; the losing
// variants are not kept in the source.
// Workgroups are ordered so that the 32 resident on one XCD (one per CU) form an 8 x 4 patch of tiles = a 2048 x 1024
// block of C: they walk K together and share their operand row-panels out of that XCD's L2.
#include "rtx_gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 sy_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float sy_f32x16_t;
typedef __attribute__((ext_vector_type(8))) int sy_i32x8_t;

// FP8 == 2: v_mfma_f32_32x32x64_f8f6f4 (gfx950's double-rate fp8 path; cbsz = blgp = 0 selects e4m3 for both operands,
// zero scale operands select the unscaled instruction).  A lane feeds 32 consecutive bytes of its row per instruction.
__device__ __forceinline__ void sy_mma_k64(sy_f32x16_t& acc, const uint4& alo, const uint4& ahi, const uint4& blo, const uint4& bhi)
{
    sy_i32x8_t a, b;
    a[0] = (int)alo.x; a[1] = (int)alo.y; a[2] = (int)alo.z; a[3] = (int)alo.w; a[4] = (int)ahi.x; a[5] = (int)ahi.y; a[6] = (int)ahi.z; a[7] = (int)ahi.w;
    b[0] = (int)blo.x; b[1] = (int)blo.y; b[2] = (int)blo.z; b[3] = (int)blo.w; b[4] = (int)bhi.x; b[5] = (int)bhi.y; b[6] = (int)bhi.z; b[7] = (int)bhi.w;
    acc = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a, b, acc, 0, 0, 0, 0, 0, 0);
}

template <int FP8> struct SyMma;
template <> struct SyMma<2> {   // placeholder: the K = 64 path does not go through run()
    static __device__ __forceinline__ void run(sy_f32x16_t&, const uint4&, const uint4&) {}
};
template <> struct SyMma<0> {
    static __device__ __forceinline__ void run(sy_f32x16_t& acc, const uint4& a, const uint4& b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(sy_bf16x8_t, a), __builtin_bit_cast(sy_bf16x8_t, b), acc, 0, 0, 0);
    }
};

struct RtxSyrk {
    const void* A;     // [rows >= 256 * m_tiles][lda elements], zero padded
    long row_bytes;    // bytes between consecutive rows of A
    long slice_bytes;  // bytes between consecutive 128-byte K slices of one row (128 = rows are K-contiguous)
    int m_tiles;       // 256-row tiles
    int n_tiles;       // 128-column tiles (the real matrix size / 128: may be one less than 2 * m_tiles)
    int k_slices;      // K bytes / 128
    float* C;          // [256 * m_tiles][ldc]
    long ldc;
};

// Same per-wave work (128 x 64 of C) and the same LDS-DMA staging, but eight waves share a 256 x 256 tile: two waves per
// SIMD, so one wave's barrier / fragment-read phases run under the other's MFMAs, and a slice of operand bytes feeds twice
// the flops.  Two LDS stages of 64 KB: slice t+1 is requested right behind the barrier of slice t and has that whole
// slice's compute time to land, so the wait at the top of an iteration is a plain vmcnt(0).
#define SY8_STAGE (512 * 128)

template <int FP8>
__global__ __launch_bounds__(512, 1) void rtx_syrk_lower_dma8(const RtxSyrk p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 2 * SY8_STAGE
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;
    const int r = lane & 31, g = lane >> 5;
    // patches of 8 x 4 tiles (2048 rows x 1024 columns of C) over the lower triangle: 32 tiles = one per CU of an XCD
    int tm, tn;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int patch = xcd + 8 * (j >> 5), within = j & 31;
        int pm = (int)((sqrtf(4.f * (float)patch + 1.f) - 1.f) * 0.5f);
        while ((pm + 1) * (pm + 2) <= patch) ++pm;
        while (pm * (pm + 1) > patch) --pm;
        const int pn = patch - pm * (pm + 1);      // 0 .. 2 pm + 1
        tm = pm * 8 + (within & 7);
        tn = pn * 4 + (within >> 3);
        if (tm >= p.m_tiles || tn > tm) return;    // m_tiles = 256-row tiles; C is square
    }
    const size_t rowb = (size_t)p.row_bytes, sliceb = (size_t)p.slice_bytes;
    const int brow = lane >> 3, chunk = (lane & 7) ^ brow;
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * 256 + wave * 8 + brow) * rowb + chunk * 16;   // + 64 rows per block step
    const unsigned char* gB = (const unsigned char*)p.A + ((size_t)tn * 256 + wave * 8 + brow) * rowb + chunk * 16;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte* lbase = (lds_byte*)smem;

    sy_f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // wave w moves the 8-row blocks w, w+8, w+16, w+24 of A and of B (256 rows each) per slice
#define SY8_LOAD(stage, ks)                                                                                          \
    {                                                                                                                \
        lds_byte* sb = lbase + (stage) * SY8_STAGE + wave * 1024;                                                    \
        const unsigned char* a_ = gA + (size_t)(ks) * sliceb;                                                        \
        const unsigned char* b_ = gB + (size_t)(ks) * sliceb;                                                        \
        _Pragma("unroll") for (int q = 0; q < 4; ++q) {                                                              \
            __builtin_amdgcn_global_load_lds((const void*)(a_ + (size_t)q * 64 * rowb), (void __attribute__((address_space(3)))*)(sb + q * 8192), 16, 0, 0); \
            __builtin_amdgcn_global_load_lds((const void*)(b_ + (size_t)q * 64 * rowb), (void __attribute__((address_space(3)))*)(sb + 256 * 128 + q * 8192), 16, 0, 0); \
        }                                                                                                            \
    }
#define SY8_RD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
#define SY8_FRAG(F, kk)                                                                               \
    {                                                                                                 \
        const unsigned sl = (unsigned)(((g + 2 * (kk)) ^ (r & 7)) * 16);                              \
        SY8_RD(F##b0, sB + sl) SY8_RD(F##b1, sB + 32 * 128 + sl)                                      \
        SY8_RD(F##a0, sA + sl) SY8_RD(F##a1, sA + 32 * 128 + sl)                                      \
        SY8_RD(F##a2, sA + 64 * 128 + sl) SY8_RD(F##a3, sA + 96 * 128 + sl)                           \
    }
#define SY8_WAIT(n)                                         \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);
#define SY8_MMA(F)                                                                                    \
    SyMma<FP8>::run(acc[0][0], F##a0, F##b0); SyMma<FP8>::run(acc[0][1], F##a0, F##b1);               \
    SyMma<FP8>::run(acc[1][0], F##a1, F##b0); SyMma<FP8>::run(acc[1][1], F##a1, F##b1);               \
    SyMma<FP8>::run(acc[2][0], F##a2, F##b0); SyMma<FP8>::run(acc[2][1], F##a2, F##b1);               \
    SyMma<FP8>::run(acc[3][0], F##a3, F##b0); SyMma<FP8>::run(acc[3][1], F##a3, F##b1);

    const int nk = p.k_slices;
    SY8_LOAD(0, 0)
    for (int t = 0; t < nk; ++t) {
        const int stage = t & 1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // my loads of slice t have landed
        __builtin_amdgcn_s_barrier();                        // everybody's have; everybody is done reading slice t-1
        SY8_LOAD(1 - stage, min(t + 1, nk - 1))              // past the end: a harmless reload
        {
            const unsigned sA = (unsigned)(size_t)(lbase + stage * SY8_STAGE + (wm * 128 + r) * 128);
            const unsigned sB = (unsigned)(size_t)(lbase + stage * SY8_STAGE + (256 + wn * 64 + r) * 128);
            if constexpr (FP8 == 2) {
                // v_mfma_f32_32x32x64_f8f6f4: a lane feeds 32 bytes (chunks 2c, 2c+1 with c = g + 2 kk) per operand
                uint4 xa0l, xa0h, xa1l, xa1h, xa2l, xa2h, xa3l, xa3h, xb0l, xb0h, xb1l, xb1h;
                uint4 ya0l, ya0h, ya1l, ya1h, ya2l, ya2h, ya3l, ya3h, yb0l, yb0h, yb1l, yb1h;
#define SY8_FRAG2(F, kk)                                                                              \
    {                                                                                                 \
        const unsigned s0 = (unsigned)((((2 * g + 4 * (kk)) ^ (r & 7)) * 16));                        \
        const unsigned s1 = (unsigned)((((2 * g + 4 * (kk) + 1) ^ (r & 7)) * 16));                    \
        SY8_RD(F##b0l, sB + s0) SY8_RD(F##b0h, sB + s1)                                               \
        SY8_RD(F##b1l, sB + 32 * 128 + s0) SY8_RD(F##b1h, sB + 32 * 128 + s1)                         \
        SY8_RD(F##a0l, sA + s0) SY8_RD(F##a0h, sA + s1)                                               \
        SY8_RD(F##a1l, sA + 32 * 128 + s0) SY8_RD(F##a1h, sA + 32 * 128 + s1)                         \
        SY8_RD(F##a2l, sA + 64 * 128 + s0) SY8_RD(F##a2h, sA + 64 * 128 + s1)                         \
        SY8_RD(F##a3l, sA + 96 * 128 + s0) SY8_RD(F##a3h, sA + 96 * 128 + s1)                         \
    }
#define SY8_MMA2(F)                                                                                   \
    sy_mma_k64(acc[0][0], F##a0l, F##a0h, F##b0l, F##b0h); sy_mma_k64(acc[0][1], F##a0l, F##a0h, F##b1l, F##b1h); \
    sy_mma_k64(acc[1][0], F##a1l, F##a1h, F##b0l, F##b0h); sy_mma_k64(acc[1][1], F##a1l, F##a1h, F##b1l, F##b1h); \
    sy_mma_k64(acc[2][0], F##a2l, F##a2h, F##b0l, F##b0h); sy_mma_k64(acc[2][1], F##a2l, F##a2h, F##b1l, F##b1h); \
    sy_mma_k64(acc[3][0], F##a3l, F##a3h, F##b0l, F##b0h); sy_mma_k64(acc[3][1], F##a3l, F##a3h, F##b1l, F##b1h);
                SY8_FRAG2(x, 0)
                SY8_FRAG2(y, 1)
                SY8_WAIT(12)
                SY8_MMA2(x)
                SY8_WAIT(0)
                SY8_MMA2(y)
#undef SY8_FRAG2
#undef SY8_MMA2
            } else {
                uint4 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
                SY8_FRAG(x, 0)
                SY8_FRAG(y, 1)
                SY8_WAIT(6)
                SY8_MMA(x)
                SY8_FRAG(x, 2)
                SY8_WAIT(6)
                SY8_MMA(y)
                SY8_FRAG(y, 3)
                SY8_WAIT(6)
                SY8_MMA(x)
                SY8_WAIT(0)
                SY8_MMA(y)
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef SY8_LOAD
#undef SY8_RD
#undef SY8_FRAG
#undef SY8_WAIT
#undef SY8_MMA
    const int ncols = p.n_tiles * 128;
    float* cp = p.C + ((size_t)tm * 256 + wm * 128 + 4 * g) * p.ldc + (size_t)tn * 256 + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (tn * 256 + wn * 64 + j * 32 >= ncols) continue;   // the last 256-column tile may overhang an odd number of 128-blocks
#pragma unroll
            for (int e = 0; e < 16; ++e) cp[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * p.ldc + j * 32] = acc[i][j][e];
        }
}

int rtx_syrk_lower_launch(const void* A, long row_bytes, long slice_bytes, int rows256, int cols128, int k_slices, int fp8, float* C, long ldc, hipStream_t stream)
{
    RTX_CHECK(A && C && rows256 > 0 && k_slices > 0, RTX_EINVAL, "syrk: bad arguments");
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY8_STAGE));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma8<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY8_STAGE));
        configured = true;
    }
    RtxSyrk p = {A, row_bytes, slice_bytes, rows256, cols128, k_slices, C, ldc};
    const int pm8 = (rows256 + 7) / 8;                // 2048-row patch rows
    const int patches8 = pm8 * (pm8 + 1);
    const dim3 grid8((unsigned)(8 * ((patches8 + 7) / 8) * 32));
    if (fp8) hipLaunchKernelGGL(rtx_syrk_lower_dma8<2>, grid8, dim3(512), 2 * SY8_STAGE, stream, p);
    else hipLaunchKernelGGL(rtx_syrk_lower_dma8<0>, grid8, dim3(512), 2 * SY8_STAGE, stream, p);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
