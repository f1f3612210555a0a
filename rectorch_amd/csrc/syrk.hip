// syrk.hip -- Gram matrix G = X^T X (lower triangle) of the EASE solver on fp8 / bf16 MFMA.
//
//   C[m][n] = sum_k A[m][k] * A[n][k],  n-tile <= m-tile,   A = X^T as a dense K(=users)-contiguous matrix
//
// K is the number of users (1e5..1e6): every tile walks a very long K, so the kernel is all main loop.  The general
// GEMM of this library (gemm.hip, 64x64 of C per wave) is bound by LDS bandwidth there: a wave reads 16 KB of LDS per
// 128-byte K slice for 32 MFMAs.  This kernel gives each wave 128x64 of C (4x2 MFMA 32x32 accumulators, 128 VGPRs): 24 KB
// of LDS reads per slice feed 64 MFMAs -- 1.33x the flops per LDS byte -- on a 256x128 workgroup tile (4 waves), with
// the same global -> register -> LDS staging, two register sets in flight and one barrier per slice (rtx_syrk_lower), or
// -- the default -- LDS-DMA staging through a ring of three stages with counted waits (rtx_syrk_lower_dma, below).
// (A slice-major operand layout, [K / 128 B][row][128 B], was measured as well -- the K-contiguous rows put the 384 rows a
// workgroup touches per slice on 384 different pages -- and made no difference; the kernel takes either through
// row_bytes / slice_bytes.)  With the 32x32x16 fp8 MFMA the kernel runs at 1.4 PFLOP/s on the lower triangle, 56 % of
// that instruction's rate; the K = 64 f8f6f4 instruction (RTX_SYRK_FP8_K64=1) is not faster in this loop.
// Workgroups are ordered so that the 32 resident on one XCD (one per CU) form a 4 x 8 patch of tiles = a 1024 x 1024
// block of C: they walk K together and share 12 operand row-panels out of that XCD's L2.
#include "rtx_gemm.h"

#include <stdlib.h>

typedef __attribute__((ext_vector_type(8))) __bf16 sy_bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float sy_f32x16_t;

#define SY_ROW 144                       // 128 B of K + 16 B pad: conflict-free ds_read_b128 fragment reads
#define SY_STAGE ((256 + 128) * SY_ROW)  // A rows | B rows of one stage

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
template <> struct SyMma<1> {
    static __device__ __forceinline__ void run(sy_f32x16_t& acc, const uint4& a, const uint4& b)
    {
        const long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
        const long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc, 0, 0, 0);
    }
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

template <int FP8>
__global__ __launch_bounds__(256, 1) void rtx_syrk_lower(const RtxSyrk p)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // 2 * SY_STAGE
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, g = lane >> 5;

    // 4 x 8 patches of (256 x 128) tiles over the lower triangle of 1024-blocks; workgroup b runs on XCD b % 8
    int tm, tn;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int patch = xcd + 8 * (j >> 5), within = j & 31;
        int pm = (int)((sqrtf(8.f * (float)patch + 1.f) - 1.f) * 0.5f);
        while ((pm + 1) * (pm + 2) / 2 <= patch) ++pm;
        while (pm * (pm + 1) / 2 > patch) --pm;
        const int pn = patch - pm * (pm + 1) / 2;
        tm = pm * 4 + (within & 3);
        tn = pn * 8 + (within >> 2);
        if (tm >= p.m_tiles || tn >= p.n_tiles || tn > 2 * tm + 1) return;
    }
    const size_t rowb = (size_t)p.row_bytes, sliceb = (size_t)p.slice_bytes;
    const int st_row = tid >> 3, st_ch = tid & 7;
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * 256 + st_row) * rowb + st_ch * 16;
    const unsigned char* gB = (const unsigned char*)p.A + ((size_t)tn * 128 + st_row) * rowb + st_ch * 16;
    const int lds_a = st_row * SY_ROW + st_ch * 16;
    const int lds_b = (256 + st_row) * SY_ROW + st_ch * 16;

    uint4 ra0, ra1, ra2, ra3, ra4, ra5, ra6, ra7, rb0, rb1, rb2, rb3;
    uint4 sa0, sa1, sa2, sa3, sa4, sa5, sa6, sa7, sb0, sb1, sb2, sb3;
    sy_f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define SY_GL(R, X, q, base, ks) R##X##q = *(const uint4*)((base) + (size_t)(q) * 32 * rowb + (size_t)(ks) * sliceb);
#define SY_GLOAD(R, ks)                                                                              \
    SY_GL(R, a, 0, gA, ks) SY_GL(R, a, 1, gA, ks) SY_GL(R, a, 2, gA, ks) SY_GL(R, a, 3, gA, ks)      \
    SY_GL(R, a, 4, gA, ks) SY_GL(R, a, 5, gA, ks) SY_GL(R, a, 6, gA, ks) SY_GL(R, a, 7, gA, ks)      \
    SY_GL(R, b, 0, gB, ks) SY_GL(R, b, 1, gB, ks) SY_GL(R, b, 2, gB, ks) SY_GL(R, b, 3, gB, ks)
#define SY_LS(R, X, q, off, st) *(uint4*)(smem + (st) * SY_STAGE + (off) + (q) * 32 * SY_ROW) = R##X##q;
#define SY_LSTORE(R, st)                                                                             \
    SY_LS(R, a, 0, lds_a, st) SY_LS(R, a, 1, lds_a, st) SY_LS(R, a, 2, lds_a, st) SY_LS(R, a, 3, lds_a, st) \
    SY_LS(R, a, 4, lds_a, st) SY_LS(R, a, 5, lds_a, st) SY_LS(R, a, 6, lds_a, st) SY_LS(R, a, 7, lds_a, st) \
    SY_LS(R, b, 0, lds_b, st) SY_LS(R, b, 1, lds_b, st) SY_LS(R, b, 2, lds_b, st) SY_LS(R, b, 3, lds_b, st)
// fragments of sub-slice kk + 1 are read from LDS while the MFMAs of sub-slice kk run: with one wave per SIMD nothing
// else hides the LDS latency
#define SY_FRAG(F, kk)                                                                                \
    F##b0 = *(const uint4*)(sB + (kk) * 32);                                                          \
    F##b1 = *(const uint4*)(sB + 32 * SY_ROW + (kk) * 32);                                            \
    F##a0 = *(const uint4*)(sA + (kk) * 32);                                                          \
    F##a1 = *(const uint4*)(sA + 32 * SY_ROW + (kk) * 32);                                            \
    F##a2 = *(const uint4*)(sA + 64 * SY_ROW + (kk) * 32);                                            \
    F##a3 = *(const uint4*)(sA + 96 * SY_ROW + (kk) * 32);
#define SY_MMA(F)                                                                                     \
    SyMma<FP8>::run(acc[0][0], F##a0, F##b0); SyMma<FP8>::run(acc[0][1], F##a0, F##b1);               \
    SyMma<FP8>::run(acc[1][0], F##a1, F##b0); SyMma<FP8>::run(acc[1][1], F##a1, F##b1);               \
    SyMma<FP8>::run(acc[2][0], F##a2, F##b0); SyMma<FP8>::run(acc[2][1], F##a2, F##b1);               \
    SyMma<FP8>::run(acc[3][0], F##a3, F##b0); SyMma<FP8>::run(acc[3][1], F##a3, F##b1);
#define SY_COMPUTE(st)                                                                                \
    if constexpr (FP8 == 2) {                                                                         \
        const unsigned char* sA = smem + (st) * SY_STAGE + (wm * 128 + r) * SY_ROW + g * 32;          \
        const unsigned char* sB = smem + (st) * SY_STAGE + (256 + wn * 64 + r) * SY_ROW + g * 32;     \
        _Pragma("unroll") for (int kk = 0; kk < 2; ++kk) {                                            \
            const uint4 b0l = *(const uint4*)(sB + kk * 64), b0h = *(const uint4*)(sB + kk * 64 + 16); \
            const uint4 b1l = *(const uint4*)(sB + 32 * SY_ROW + kk * 64), b1h = *(const uint4*)(sB + 32 * SY_ROW + kk * 64 + 16); \
            _Pragma("unroll") for (int i = 0; i < 4; ++i) {                                           \
                const uint4 al = *(const uint4*)(sA + i * 32 * SY_ROW + kk * 64);                     \
                const uint4 ah = *(const uint4*)(sA + i * 32 * SY_ROW + kk * 64 + 16);                \
                sy_mma_k64(acc[i][0], al, ah, b0l, b0h);                                              \
                sy_mma_k64(acc[i][1], al, ah, b1l, b1h);                                              \
            }                                                                                         \
        }                                                                                             \
    } else {                                                                                          \
        const unsigned char* sA = smem + (st) * SY_STAGE + (wm * 128 + r) * SY_ROW + g * 16;          \
        const unsigned char* sB = smem + (st) * SY_STAGE + (256 + wn * 64 + r) * SY_ROW + g * 16;     \
        uint4 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;                             \
        SY_FRAG(x, 0)                                                                                 \
        SY_FRAG(y, 1)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        SY_MMA(x)                                                                                     \
        SY_FRAG(x, 2)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        SY_MMA(y)                                                                                     \
        SY_FRAG(y, 3)                                                                                 \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        SY_MMA(x)                                                                                     \
        SY_MMA(y)                                                                                     \
    }

    const int nk = p.k_slices;
    // software pipeline, depth 2 (see gemm.hip): loads of the steady-state loop are unconditional
    if (nk == 1) {
        SY_GLOAD(r, 0)
        SY_LSTORE(r, 0)
        __syncthreads();
        SY_COMPUTE(0)
    } else if (nk > 1) {
        SY_GLOAD(r, 0)
        SY_LSTORE(r, 0)
        SY_GLOAD(r, 1)
        __syncthreads();
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            SY_GLOAD(s, t + 2)
            __builtin_amdgcn_sched_barrier(0);
            SY_COMPUTE(0)
            SY_LSTORE(r, 1)
            __syncthreads();
            SY_GLOAD(r, t + 3)
            __builtin_amdgcn_sched_barrier(0);
            SY_COMPUTE(1)
            SY_LSTORE(s, 0)
            __syncthreads();
        }
        const bool three = (nk - t) == 3;
        if (three) { SY_GLOAD(s, t + 2) }
        SY_COMPUTE(0)
        SY_LSTORE(r, 1)
        __syncthreads();
        SY_COMPUTE(1)
        if (three) {
            SY_LSTORE(s, 0)
            __syncthreads();
            SY_COMPUTE(0)
        }
    }
#undef SY_GL
#undef SY_GLOAD
#undef SY_LS
#undef SY_LSTORE
#undef SY_COMPUTE
#undef SY_FRAG
#undef SY_MMA

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* cp = p.C + ((size_t)tm * 256 + wm * 128 + 4 * g) * p.ldc + (size_t)tn * 128 + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) cp[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * p.ldc + j * 32] = acc[i][j][e];
}

// ---- LDS-DMA variant ------------------------------------------------------------------------------------------------
// Same tile geometry, but the operand slices go global -> LDS directly (global_load_lds_dwordx4: no staging registers,
// no ds_write pass), through a ring of THREE stages, with counted waits: at the top of iteration t a wave waits for its
// own loads of slice t (vmcnt(12): the 12 loads of slice t+1 stay in flight across the barrier), the workgroup barrier
// makes everybody's slice t visible and retires everybody's reads of slice t-1, whose stage is then refilled with
// slice t+2.  The DMA writes LDS lane-linear (wave-uniform base + lane * 16 B), so rows are 128 B apart with no padding;
// bank conflicts of the fragment reads are avoided by an XOR swizzle of the 16-byte chunk index with (row & 7), applied
// to the per-lane GLOBAL address (the 8 lanes of a row still read one 128-byte line) and to the ds_read address.
#define SYG_STAGE ((256 + 128) * 128)

template <int FP8>
__global__ __launch_bounds__(256, 1) void rtx_syrk_lower_dma(const RtxSyrk p)
{
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];   // 3 * SYG_STAGE
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, g = lane >> 5;
    int tm, tn;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        const int patch = xcd + 8 * (j >> 5), within = j & 31;
        int pm = (int)((sqrtf(8.f * (float)patch + 1.f) - 1.f) * 0.5f);
        while ((pm + 1) * (pm + 2) / 2 <= patch) ++pm;
        while (pm * (pm + 1) / 2 > patch) --pm;
        const int pn = patch - pm * (pm + 1) / 2;
        tm = pm * 4 + (within & 3);
        tn = pn * 8 + (within >> 2);
        if (tm >= p.m_tiles || tn >= p.n_tiles || tn > 2 * tm + 1) return;
    }
    const size_t rowb = (size_t)p.row_bytes, sliceb = (size_t)p.slice_bytes;
    // lane -> (row within an 8-row block, swizzled chunk) of the 1-KB block a wave instruction moves
    const int brow = lane >> 3, chunk = (lane & 7) ^ brow;
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * 256 + wave * 8 + brow) * rowb + chunk * 16;   // + 32 rows per block step
    const unsigned char* gB = (const unsigned char*)p.A + ((size_t)tn * 128 + wave * 8 + brow) * rowb + chunk * 16;
    typedef __attribute__((address_space(3))) unsigned char lds_byte;
    lds_byte* lbase = (lds_byte*)smem;

    sy_f32x16_t acc[4][2];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // wave w moves the 8-row blocks w, w+4, ... : 8 blocks of A (256 rows) and 4 of B (128 rows) per slice
#define SYG_LOAD(stage, ks)                                                                                          \
    {                                                                                                                \
        lds_byte* sb = lbase + (stage) * SYG_STAGE + wave * 1024;                                                    \
        const unsigned char* a_ = gA + (size_t)(ks) * sliceb;                                                        \
        const unsigned char* b_ = gB + (size_t)(ks) * sliceb;                                                        \
        _Pragma("unroll") for (int q = 0; q < 8; ++q)                                                                \
            __builtin_amdgcn_global_load_lds((const void*)(a_ + (size_t)q * 32 * rowb), (void __attribute__((address_space(3)))*)(sb + q * 4096), 16, 0, 0); \
        _Pragma("unroll") for (int q = 0; q < 4; ++q)                                                                \
            __builtin_amdgcn_global_load_lds((const void*)(b_ + (size_t)q * 32 * rowb), (void __attribute__((address_space(3)))*)(sb + 256 * 128 + q * 4096), 16, 0, 0); \
    }
    // fragment reads: row R of the tile at R * 128, 16-byte chunk c at slot c ^ (R & 7); R & 7 == r & 7 for every fragment row
// The fragment reads are inline asm: a compiler-visible LDS read placed after an LDS-DMA in flight gets an
// `s_waitcnt vmcnt(0)` in front of it (the DMA could alias it), which would drain the ring every iteration.  The asm reads
// are ordered by hand: LDS operations retire in order, so `lgkmcnt(6)` after issuing the next set of six means "the previous
// set has arrived"; a sched_barrier behind the wait keeps the machine scheduler from hoisting the MFMAs above it.
#define SYG_RD(dst, addr) asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr));
#define SYG_FRAG(F, kk)                                                                               \
    {                                                                                                 \
        const unsigned sl = (unsigned)(((g + 2 * (kk)) ^ (r & 7)) * 16);                              \
        SYG_RD(F##b0, sB + sl) SYG_RD(F##b1, sB + 32 * 128 + sl)                                      \
        SYG_RD(F##a0, sA + sl) SYG_RD(F##a1, sA + 32 * 128 + sl)                                      \
        SYG_RD(F##a2, sA + 64 * 128 + sl) SYG_RD(F##a3, sA + 96 * 128 + sl)                           \
    }
#define SYG_WAIT(F, n)                                      \
    asm volatile("s_waitcnt lgkmcnt(" #n ")" ::: "memory");   \
    __builtin_amdgcn_sched_barrier(0);
#define SYG_MMA(F)                                                                                    \
    SyMma<FP8>::run(acc[0][0], F##a0, F##b0); SyMma<FP8>::run(acc[0][1], F##a0, F##b1);               \
    SyMma<FP8>::run(acc[1][0], F##a1, F##b0); SyMma<FP8>::run(acc[1][1], F##a1, F##b1);               \
    SyMma<FP8>::run(acc[2][0], F##a2, F##b0); SyMma<FP8>::run(acc[2][1], F##a2, F##b1);               \
    SyMma<FP8>::run(acc[3][0], F##a3, F##b0); SyMma<FP8>::run(acc[3][1], F##a3, F##b1);

    const int nk = p.k_slices;
    SYG_LOAD(0, 0)
    SYG_LOAD(1, min(1, nk - 1))
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");   // my loads of slice t have landed (slice t+1 may be in flight)
        __builtin_amdgcn_s_barrier();                        // everybody's have; everybody is done reading slice t-1
        {
            const int nst = stage == 0 ? 2 : stage - 1;      // (t + 2) % 3 == (t - 1) % 3: the stage just released
            SYG_LOAD(nst, min(t + 2, nk - 1))                // past the end: a harmless reload, keeps the counts static
        }
        {
            // LDS byte addresses (the asm reads take the 32-bit LDS offset)
            const unsigned sA = (unsigned)(size_t)(lbase + stage * SYG_STAGE + (wm * 128 + r) * 128);
            const unsigned sB = (unsigned)(size_t)(lbase + stage * SYG_STAGE + (256 + wn * 64 + r) * 128);
            uint4 xa0, xa1, xa2, xa3, xb0, xb1, ya0, ya1, ya2, ya3, yb0, yb1;
            SYG_FRAG(x, 0)
            SYG_FRAG(y, 1)
            SYG_WAIT(x, 6)
            SYG_MMA(x)
            SYG_FRAG(x, 2)
            SYG_WAIT(y, 6)
            SYG_MMA(y)
            SYG_FRAG(y, 3)
            SYG_WAIT(x, 6)
            SYG_MMA(x)
            SYG_WAIT(y, 0)
            SYG_MMA(y)
        }
        stage = stage == 2 ? 0 : stage + 1;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");         // the two reloads past the end
#undef SYG_LOAD
#undef SYG_FRAG
#undef SYG_MMA
#undef SYG_RD
#undef SYG_WAIT
    float* cp = p.C + ((size_t)tm * 256 + wm * 128 + 4 * g) * p.ldc + (size_t)tn * 128 + wn * 64 + r;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) cp[(size_t)(i * 32 + (e & 3) + 8 * (e >> 2)) * p.ldc + j * 32] = acc[i][j][e];
}

// ---- 256 x 256 tile, 8 waves ------------------------------------------------------------------------------------------
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
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY_STAGE));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY_STAGE));
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY_STAGE));
        configured = true;
    }
    RtxSyrk p = {A, row_bytes, slice_bytes, rows256, cols128, k_slices, C, ldc};
    const int pr = (rows256 + 3) / 4;                 // 1024-row patch rows
    const int patches = pr * (pr + 1) / 2;
    const dim3 grid((unsigned)(8 * ((patches + 7) / 8) * 32));
    static int dma = -1;
    if (dma < 0) {
        const char* v = getenv("RTX_SYRK_DMA");
        dma = v ? atoi(v) : 8;   // default 8: 256x256 tiles / 8 waves; 1: 256x128 / 4 waves; 0: register-staged (measurement switch)
        if (dma) {
            RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SYG_STAGE));
            RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 3 * SYG_STAGE));
        }
    }
    if (dma == 8) {   // 256 x 256 tiles, 8 waves
        static bool conf8 = false;
        if (!conf8) {
            RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma8<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY8_STAGE));
            RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma8<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY8_STAGE));
            RTX_HIP(hipFuncSetAttribute((const void*)rtx_syrk_lower_dma8<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * SY8_STAGE));
            conf8 = true;
        }
        const int pm8 = (rows256 + 7) / 8;
        const int patches8 = pm8 * (pm8 + 1);
        const dim3 grid8((unsigned)(8 * ((patches8 + 7) / 8) * 32));
        static int k64_8 = -1;
        if (k64_8 < 0) { const char* v = getenv("RTX_SYRK_FP8_K64"); k64_8 = v ? (atoi(v) ? 1 : 0) : 1; }   // default: the double-rate instruction
        if (fp8 && k64_8) hipLaunchKernelGGL(rtx_syrk_lower_dma8<2>, grid8, dim3(512), 2 * SY8_STAGE, stream, p);
        else if (fp8) hipLaunchKernelGGL(rtx_syrk_lower_dma8<1>, grid8, dim3(512), 2 * SY8_STAGE, stream, p);
        else hipLaunchKernelGGL(rtx_syrk_lower_dma8<0>, grid8, dim3(512), 2 * SY8_STAGE, stream, p);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    if (dma) {
        if (fp8) hipLaunchKernelGGL(rtx_syrk_lower_dma<1>, grid, dim3(256), 3 * SYG_STAGE, stream, p);
        else hipLaunchKernelGGL(rtx_syrk_lower_dma<0>, grid, dim3(256), 3 * SYG_STAGE, stream, p);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    static int k64 = -1;
    if (k64 < 0) { const char* v = getenv("RTX_SYRK_FP8_K64"); k64 = (v && atoi(v)) ? 1 : 0; }   // 1: v_mfma_f32_32x32x64_f8f6f4
    if (fp8 && k64)
        hipLaunchKernelGGL(rtx_syrk_lower<2>, grid, dim3(256), 2 * SY_STAGE, stream, p);
    else if (fp8)
        hipLaunchKernelGGL(rtx_syrk_lower<1>, grid, dim3(256), 2 * SY_STAGE, stream, p);
    else
        hipLaunchKernelGGL(rtx_syrk_lower<0>, grid, dim3(256), 2 * SY_STAGE, stream, p);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
