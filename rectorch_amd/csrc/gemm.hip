// gemm.hip -- MFMA NT GEMM for gfx950 (see rtx_gemm.h for the design notes).
//
// Why several tile shapes.  Everything a workgroup multiplies passes through its CU's vector L1 at 64 B/clk,
// while the CU's four matrix pipes retire ~4650 bf16 flop/clk.  A 128x128 tile moves (128+128)*128 B per
// 128-byte K slice = 512 clk of L1 time for 451 clk of MFMA time: L1-bound below half of the MFMA rate
// (measured: SQ_VALU_MFMA_BUSY 11 % of wave-cycles, L2 hit traffic = the predicted 202 MB on the logits GEMM).
// 256x128 / 128x256 (8 waves) need 768 clk of L1 per 902 clk of MFMA, so the big contractions of the step use
// those; 128x128 (4 waves) stays for the small hidden-layer GEMMs.
#include "rtx_gemm.h"

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// LDS row = KB bytes of K + 16 B pad: conflict-free ds_read_b128 fragment reads for KB = 128 (36-dword stride) and KB = 64 (20-dword
// stride: rows r = 0..15 of a lane group start at banks 20 r mod 64 = 0,20,40,60,16,36,56,12,32,52,8,28,48,4,24,44 -- 16 disjoint quads)
#define RTX_LDS_ROW_OF(KB) ((KB) + 16)

template <typename T> struct Mma;

template <> struct Mma<bf16_t> {
    // one ds_read_b128 per operand = 8 bf16 = the whole K=16 fragment of v_mfma_f32_32x32x16_bf16
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b)
    {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b),
                                                      acc, 0, 0, 0);
    }
};

template <> struct Mma<fp8_t> {
    // 16 B = 16 fp8 of this lane's row = two K=16 fragments of v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3 on gfx950).
    // Which 8 of the 32 k of a 32-byte sub-slice a lane feeds to which instruction is irrelevant as long as A and B
    // agree, and they do: both operands use the same (lane, byte) -> k map.
    static __device__ __forceinline__ void run(f32x16_t& acc, const uint4& a, const uint4& b)
    {
        const long a0 = (long)(((unsigned long)a.y << 32) | a.x), a1 = (long)(((unsigned long)a.w << 32) | a.z);
        const long b0 = (long)(((unsigned long)b.y << 32) | b.x), b1 = (long)(((unsigned long)b.w << 32) | b.z);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a0, b0, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_fp8_fp8(a1, b1, acc, 0, 0, 0);
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

// WM x WN waves; every wave owns a 64 x (32*NB) block of C (2 x NB MFMA 32x32 accumulators).
// 128x128 = 2x2 waves, 256x128 = 4x2, 128x256 = 2x4.  (256x256 was tried as 4x2 waves of 64x128 and as 4x4 waves of
// 64x64: both spill accumulators at their VGPR budgets with this source structure, so it is not shipped.)
// KB: bytes of K per row and slice.  128 everywhere but the logits product (K = hidden width, 10 slices): at 128 its 128 x 128 tile needs
// 73.7 KB of LDS -- two workgroups per CU, 512 slots for 632 tiles: a second, quarter-full round (measured: each workgroup lives 14 us, the
// kernel 28).  At KB = 64 a workgroup takes 41 KB and <= 168 registers: THREE per CU, 768 slots, every tile resident at once.
template <int WM, int WN, int NB, int KB = 128> struct TileCfg {
    static constexpr int NT = WM * WN * 64;       // threads
    static constexpr int BM = WM * 64;            // tile rows
    static constexpr int BN = WN * NB * 32;       // tile columns
    static constexpr int LPR = KB / 16;           // lanes per row slice (16 B each)
    static constexpr int RPP = NT / LPR;          // rows staged per pass
    static constexpr int PA = BM / RPP;           // staging passes for A (2 or 4)
    static constexpr int PB = BN / RPP;           // staging passes for B (2 or 4)
    static constexpr int ROW = RTX_LDS_ROW_OF(KB);
    static constexpr int STAGE = (BM + BN) * ROW;
    static_assert(PA == 2 || PA == 4, "PA");
    static_assert(PB == 2 || PB == 4, "PB");
};

// ABL (measurement builds only, tests/native/test_gemm.cpp "ablate"): 1 = no global loads in the loop, 2 = no LDS stores, 4 = no MFMAs (fragment reads
// kept alive), 8 = no fragment reads and no MFMAs, 16 = no barriers, 32 = no epilogue stores.  0 in every shipped instantiation.
template <typename T, int EPI, int WM, int WN, int NB, int KB = 128, int ABL = 0, int SCH = 0>
__global__ __launch_bounds__(WM* WN * 64, KB == 64 ? 3 : ((WM * WN) / 4 > 2 ? (WM * WN) / 4 : 2)) void rtx_gemm_nt(const RtxGemm p)
{
    using Cfg = TileCfg<WM, WN, NB, KB>;
    constexpr int BM = Cfg::BM, BN = Cfg::BN, RPP = Cfg::RPP, PA = Cfg::PA, PB = Cfg::PB, STAGE = Cfg::STAGE;
    constexpr int RTX_LDS_ROW = Cfg::ROW, LPR_ST = Cfg::LPR;
    // [stage][A rows | B rows]; all LDS in one array (a second __shared__ object de-pipelines, cdna guide)
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * STAGE];

    if (p.wait_word) rtx_fold_wait(p.wait_word, p.wait_seq);   // a cross-stream dependency folded into this kernel (rtx_gemm.h)
    if (ABL & 64) {   // measurement: the second workgroup of a CU (odd wave slot of its SIMD) starts half a slice late: out of phase with the first
        const unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_ID: wave_id in bits 3:0
        if (hw & 1) __builtin_amdgcn_s_sleep(20);
    }
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, g = lane >> 5;

    // XCD-aware work mapping.  Workgroup b is dispatched to XCD b % 8 (observed, used for speed only), and each
    // XCD has a private 4 MB L2.  Workgroups that read the same operand panel are therefore placed on ONE
    // XCD, back to back in its dispatch sequence:
    //   split-K   : all output tiles of one K-split share that split's A and B K-slices
    //   otherwise : the (few) tiles along the short dimension share the long dimension's operand tile
    // (rocprof: TCC misses dropped to the compulsory bytes with this mapping.)
    int tm, tn, split;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        if (p.syrk_lower) {
            // A == B, lower tiles only (Gram matrix of the EASE solver).  K is far longer than any cache, so what
            // matters is that the 64 workgroups resident on one XCD at a time form an 8x8 PATCH of tiles: they walk K
            // together and 16 operand row-panels feed 64 tile products out of the XCD's L2.
            split = 0;
            const int patch = xcd + 8 * (j >> 6), within = j & 63;
            int pm = (int)((sqrtf(8.f * (float)patch + 1.f) - 1.f) * 0.5f);
            while ((pm + 1) * (pm + 2) / 2 <= patch) ++pm;
            while (pm * (pm + 1) / 2 > patch) --pm;
            const int pn = patch - pm * (pm + 1) / 2;
            tm = pm * 8 + (within & 7);
            tn = pn * 8 + (within >> 3);
            if (tm >= p.m_tiles || tn > tm) return;
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
    }
    const int k_sl = p.k_slices * (128 / KB);      // (p.k_slices counts 128-byte slices)
    const int per = (k_sl + p.splits - 1) / p.splits;
    const int ks0 = split * per;
    const int ks1 = min(ks0 + per, k_sl);
    const int nk = ks1 - ks0;

    const size_t rowA = (size_t)p.lda * sizeof(T), rowB = (size_t)p.ldb * sizeof(T);
    // bytes between two K slices of one row: KB for a row-major operand; RtxGemm::a_slice_stride / b_slice_stride for a K-BLOCKED image
    // ([K / slice][rows][slice]: lda = ldb = one slice, a slice of a tile is ONE contiguous run)
    const size_t kstA = p.a_slice_stride ? (size_t)p.a_slice_stride : (size_t)KB, kstB = p.b_slice_stride ? (size_t)p.b_slice_stride : (size_t)KB;
    const int st_row = tid / LPR_ST, st_ch = tid % LPR_ST;  // staging: LPR lanes x 16 B = one KB-byte row slice
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * BM + st_row) * rowA + st_ch * 16;
    const unsigned char* gB = (const unsigned char*)p.B + ((size_t)tn * BN + st_row) * rowB + st_ch * 16;
    const int lds_a = st_row * RTX_LDS_ROW + st_ch * 16;
    const int lds_b = (BM + st_row) * RTX_LDS_ROW + st_ch * 16;

    // two register sets (named, so they stay in VGPRs): K-slices t+1 and t+2 are in flight while slice t is
    // multiplied -- the loop is latency-bound otherwise (one slice of MFMA work is shorter than an L2 round trip)
    uint4 ra0, ra1, ra2, ra3, rb0, rb1, rb2, rb3;
    uint4 sa0, sa1, sa2, sa3, sb0, sb1, sb2, sb3;
    uint4 ua0, ua1, ua2, ua3, ub0, ub1, ub2, ub3;   // SCH == 2: the third set (three slices ahead)
    f32x16_t acc[2][NB];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NB; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

#define RTX_GL(R, X, q, base, row, ks) R##X##q = *(const uint4*)((base) + (size_t)(q) * RPP * (row) + (size_t)(ks) * ((row) == rowA ? kstA : kstB));
#define RTX_GLOAD(R, ks)                                                    \
    if (!(ABL & 1) || (ks) == ks0)                                           \
    RTX_GLOAD_(R, ks)
#define RTX_GLOAD_(R, ks) {                                                 \
    RTX_GL(R, a, 0, gA, rowA, ks) RTX_GL(R, a, 1, gA, rowA, ks)             \
    if (PA == 4) { RTX_GL(R, a, 2, gA, rowA, ks) RTX_GL(R, a, 3, gA, rowA, ks) } \
    RTX_GL(R, b, 0, gB, rowB, ks) RTX_GL(R, b, 1, gB, rowB, ks)             \
    if (PB == 4) { RTX_GL(R, b, 2, gB, rowB, ks) RTX_GL(R, b, 3, gB, rowB, ks) } }
#define RTX_LS(R, X, q, off, st) *(uint4*)(smem + (st) * STAGE + (off) + (q) * RPP * RTX_LDS_ROW) = R##X##q;
#define RTX_LSTORE(R, st)                                                   \
    if (!(ABL & 2))                                                          \
    RTX_LSTORE_(R, st)
#define RTX_LSTORE_(R, st) {                                                \
    RTX_LS(R, a, 0, lds_a, st) RTX_LS(R, a, 1, lds_a, st)                   \
    if (PA == 4) { RTX_LS(R, a, 2, lds_a, st) RTX_LS(R, a, 3, lds_a, st) }  \
    RTX_LS(R, b, 0, lds_b, st) RTX_LS(R, b, 1, lds_b, st)                   \
    if (PB == 4) { RTX_LS(R, b, 2, lds_b, st) RTX_LS(R, b, 3, lds_b, st) } }
#define RTX_COMPUTE(st)                                                                                       \
    if (!(ABL & 8)) {                                                                                         \
        const unsigned char* sA = smem + (st) * STAGE + (wm * 64 + r) * RTX_LDS_ROW + g * 16;                 \
        const unsigned char* sB = smem + (st) * STAGE + (BM + wn * (NB * 32) + r) * RTX_LDS_ROW + g * 16;     \
        if (SCH == 1) {                                                                                       \
            /* every fragment read of the slice in flight before the first MFMA (ablation, round 6: the reads' LATENCY chain -- */ \
            /* read, wait, multiply, read ... -- was half of this loop's time; the matrix pipe was never the limit) */         \
            uint4 fa[KB / 32][2], fb[KB / 32][NB];                                                            \
            _Pragma("unroll") for (int kk = 0; kk < KB / 32; ++kk) {                                          \
                fa[kk][0] = *(const uint4*)(sA + kk * 32);                                                    \
                fa[kk][1] = *(const uint4*)(sA + 32 * RTX_LDS_ROW + kk * 32);                                 \
                _Pragma("unroll") for (int j = 0; j < NB; ++j) fb[kk][j] = *(const uint4*)(sB + j * 32 * RTX_LDS_ROW + kk * 32); \
            }                                                                                                 \
            _Pragma("unroll") for (int kk = 0; kk < KB / 32; ++kk)                                            \
                _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                              \
                    if (ABL & 4) { asm volatile("" ::"v"(fa[kk][0].x), "v"(fa[kk][1].w), "v"(fb[kk][j].x)); continue; } \
                    Mma<T>::run(acc[0][j], fa[kk][0], fb[kk][j]);                                             \
                    Mma<T>::run(acc[1][j], fa[kk][1], fb[kk][j]);                                             \
                }                                                                                             \
        } else {                                                                                              \
        _Pragma("unroll") for (int kk = 0; kk < KB / 32; ++kk) {                                              \
            const uint4 a0 = *(const uint4*)(sA + kk * 32);                                                   \
            const uint4 a1 = *(const uint4*)(sA + 32 * RTX_LDS_ROW + kk * 32);                                \
            _Pragma("unroll") for (int j = 0; j < NB; ++j) {                                                  \
                const uint4 b = *(const uint4*)(sB + j * 32 * RTX_LDS_ROW + kk * 32);                         \
                if (ABL & 4) { asm volatile("" ::"v"(a0.x), "v"(a0.w), "v"(a1.x), "v"(a1.w), "v"(b.x), "v"(b.w)); continue; } \
                Mma<T>::run(acc[0][j], a0, b);                                                                \
                Mma<T>::run(acc[1][j], a1, b);                                                                \
            }                                                                                                 \
        }                                                                                                     \
        }                                                                                                     \
    }

    // Software pipeline, depth 2: while slice t is multiplied out of LDS, slices t+1 and t+2 are in flight in the
    // two register sets.  The steady-state loop issues its loads UNCONDITIONALLY: with a branch around a load,
    // hipcc's waitcnt pass merges the "issued" and "not issued" states and falls back to vmcnt(0) before the LDS
    // stores, which drains the younger prefetch every slice (seen in the .s; it cost the whole second stage).
#define RTX_SYNC() do { if (!(ABL & 16)) __syncthreads(); } while (0)
    if (SCH == 2) {
        // Depth 3 (round 6): THREE register sets rotate over the two LDS stages, so a slice's loads have three compute phases to land
        // instead of two (the ablation puts ~5 of the first-layer product's 24 us on exposed load latency: a slice takes ~1 us per
        // workgroup because its 256 row pieces come from as many DRAM pages).  Invariant at step t (t = 0 mod 6): LDS stage 0 = slice t,
        // set r = slice t + 1, set s = slice t + 2 (in flight).  Every load is unconditional (clamped to the split's last slice: a branch
        // around a load costs the younger prefetches, see above); the multiply and the LDS store of slices beyond the split are skipped.
        if (nk > 0) {
            RTX_GLOAD(r, ks0)
            RTX_LSTORE(r, 0)
            RTX_GLOAD(r, min(ks0 + 1, ks1 - 1))
            RTX_GLOAD(s, min(ks0 + 2, ks1 - 1))
            RTX_SYNC();
#define RTX_STEP(U, R, st, k)                                                   \
            RTX_GLOAD(U, min(ks0 + t + (k) + 3, ks1 - 1))                       \
            __builtin_amdgcn_sched_barrier(0);                                  \
            if (t + (k) < nk) { RTX_COMPUTE(st) }                               \
            if (t + (k) + 1 < nk) { RTX_LSTORE(R, 1 - (st)) }                   \
            RTX_SYNC();
            for (int t = 0; t < nk; t += 6) {
                RTX_STEP(u, r, 0, 0) RTX_STEP(r, s, 1, 1) RTX_STEP(s, u, 0, 2)
                RTX_STEP(u, r, 1, 3) RTX_STEP(r, s, 0, 4) RTX_STEP(s, u, 1, 5)
            }
#undef RTX_STEP
        }
    } else
    if (nk == 1) {
        RTX_GLOAD(r, ks0)
        RTX_LSTORE(r, 0)
        RTX_SYNC();
        RTX_COMPUTE(0)
    } else if (nk > 1) {
        RTX_GLOAD(r, ks0)
        RTX_LSTORE(r, 0)
        RTX_GLOAD(r, ks0 + 1)
        RTX_SYNC();
        int t = 0;
        for (; t + 3 < nk; t += 2) {
            // LDS stage 0 = slice t, set r = slice t+1 (in flight)
            RTX_GLOAD(s, ks0 + t + 2)
            __builtin_amdgcn_sched_barrier(0);   // keep the prefetch ahead of the MFMAs (the scheduler sinks it otherwise)
            RTX_COMPUTE(0)
            RTX_LSTORE(r, 1)
            RTX_SYNC();
            // LDS stage 1 = slice t+1, set s = slice t+2 (in flight)
            RTX_GLOAD(r, ks0 + t + 3)
            __builtin_amdgcn_sched_barrier(0);
            RTX_COMPUTE(1)
            RTX_LSTORE(s, 0)
            RTX_SYNC();
        }
        // 2 or 3 slices left: stage 0 = slice t, set r = slice t+1
        const bool three = (nk - t) == 3;
        if (three) { RTX_GLOAD(s, ks0 + t + 2) }
        RTX_COMPUTE(0)
        RTX_LSTORE(r, 1)
        RTX_SYNC();
        RTX_COMPUTE(1)
        if (three) {
            RTX_LSTORE(s, 0)
            RTX_SYNC();
            RTX_COMPUTE(0)
        }
    }
#undef RTX_SYNC
#undef RTX_GLOAD_
#undef RTX_LSTORE_
#undef RTX_GL
#undef RTX_GLOAD
#undef RTX_LS
#undef RTX_LSTORE
#undef RTX_COMPUTE

    // epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5).
    // Per-column values (bias, column validity) are fetched ONCE before the store loop: a load inside the loop
    // makes hipcc wait vmcnt(0) per element, which also drains the stores (CDNA counts them in vmcnt) and
    // serialised the whole epilogue (12 us of the logits GEMM).
    if constexpr (EPI == RTX_EPI_BIAS_ROWS) {
        // Logits epilogue through LDS (the operand stages are dead): each wave parks 32 x 64 of C at a time in a private
        // region and re-reads it row-wise -- two lanes per row, 32 columns each -- so the stores are 16-byte pieces of
        // 128-byte row segments and the online-softmax partial (max, sum exp) of a row over the wave's 64-column strip
        // needs ONE shuffle.  (Round 1 reduced in the MFMA layout: 32 shuffle chains, +12 us; stores were 4 bytes per lane.)
        typedef __attribute__((ext_vector_type(4))) float f32x4_t;
        constexpr int WCOLS = NB * 32, WLD = WCOLS + 4, HC = WCOLS / 2, NQ = HC / 4;
        static_assert(WM * WN * 32 * WLD * 4 <= 2 * STAGE, "epilogue scratch must fit in the stages");
        __syncthreads();   // every wave has finished reading the operand stages
        float* wreg = (float*)smem + wave * (32 * WLD);
        const int erow = lane >> 1, ehalf = lane & 1;
        constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;   // store pass: lanes per row, rows per store instruction
        const int srow = lane / LPR, scol = (lane % LPR) * 4;
        const int col0 = tn * BN + wn * WCOLS;
        float bj[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col0 + j * 32 + r;
            bj[j] = p.bias ? p.bias[col < p.N_real ? col : 0] : 0.f;
        }
        const long ld = p.ldc;
        const bool vec_ok = ((ld & 3) == 0) && (((uintptr_t)p.C & 15) == 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) {
#pragma unroll
            for (int j = 0; j < NB; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) wreg[((e & 3) + 8 * (e >> 2) + 4 * g) * WLD + j * 32 + r] = acc[i][j][e] + bj[j];
            __builtin_amdgcn_wave_barrier();
            f32x4_t v[NQ];
#pragma unroll
            for (int q = 0; q < NQ; ++q) v[q] = *(const f32x4_t*)(wreg + erow * WLD + ehalf * HC + q * 4);
            __builtin_amdgcn_wave_barrier();
            const int row = tm * BM + wm * 64 + i * 32 + erow;
            const int colh = col0 + ehalf * HC;
            if (p.lse_part) {
                float m = -INFINITY;
#pragma unroll
                for (int q = 0; q < NQ; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k)
                        if (colh + q * 4 + k < p.N_real) m = fmaxf(m, v[q][k]);
                const float mm = fmaxf(m, __shfl_xor(m, 1, 64));
                float sum = 0.f;
                if (mm != -INFINITY) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q)
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (colh + q * 4 + k < p.N_real) sum += __expf(v[q][k] - mm);
                }
                sum += __shfl_xor(sum, 1, 64);
                if (ehalf == 0 && row < p.M_real) p.lse_part[(size_t)row * p.lse_ld + (tn * WN + wn)] = make_float2(mm, sum);
            }
            // the stores take a second, column-major read of the parked block: LPR lanes side by side cover one row's WCOLS * 4
            // bytes, so a store instruction writes RPI whole row segments (full 128-byte lines) -- the row-per-lane-pair
            // registers above would put each lane's 16 bytes on a line of its own (64 partial-line requests per instruction)
            f32x4_t w[32 / RPI];
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) w[it] = *(const f32x4_t*)(wreg + (it * RPI + srow) * WLD + scol);
            __builtin_amdgcn_wave_barrier();
            if (p.C16) {
                // half-precision logits (the training step): 8 bytes per lane, LPR lanes = one row's WCOLS * 2 bytes (whole 128-byte
                // lines); padding columns (>= N_real, < ldc16) receive don't-care values -- the loss kernel masks and zeroes them
                typedef __attribute__((ext_vector_type(4))) _Float16 f16x4_t;
#pragma unroll
                for (int it = 0; it < 32 / RPI; ++it) {
                    const int orow = tm * BM + wm * 64 + i * 32 + it * RPI + srow;
                    if (orow < p.M_real) {
                        f16x4_t h;
#pragma unroll
                        for (int k = 0; k < 4; ++k) h[k] = (_Float16)__builtin_amdgcn_fmed3f(w[it][k], -65504.f, 65504.f);
                        *(f16x4_t*)((_Float16*)p.C16 + (size_t)orow * p.ldc16 + col0 + scol) = h;
                    }
                }
                continue;
            }
#pragma unroll
            for (int it = 0; it < 32 / RPI; ++it) {
                const int orow = tm * BM + wm * 64 + i * 32 + it * RPI + srow;
                const int c = col0 + scol;
                if (orow < p.M_real) {
                    float* dst = p.C + (size_t)orow * ld + c;
                    if (vec_ok && c + 3 < p.N_real) {
                        *(f32x4_t*)dst = w[it];
                    } else {
#pragma unroll
                        for (int k = 0; k < 4; ++k)
                            if (c + k < p.N_real) dst[k] = w[it][k];
                    }
                }
            }
        }
        return;
    }
    const int row_base = tm * BM + wm * 64 + 4 * g;
    const int col_base = tn * BN + wn * (NB * 32) + r;
    // one 64-bit base per lane; every element offset is (compile-time constant) * ld + constant -> scalar math
    const long ld = (EPI == RTX_EPI_GRAD) ? (long)p.N_real : p.ldc;
    float* cp = p.C + (EPI == RTX_EPI_STORE ? (size_t)split * p.slab_stride : (size_t)0) + (size_t)row_base * ld + col_base;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int j = 0; j < NB; ++j) {
            const int col = col_base + j * 32;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int dr = i * 32 + (e & 3) + 8 * (e >> 2);
                const int row = row_base + dr;
                const float v = acc[i][j][e];
                float* dst = cp + (long)dr * ld + j * 32;
                if (EPI == RTX_EPI_STORE) {
                    if (!(ABL & 32) || v == 12345.678f) *dst = v;
                } else {  // RTX_EPI_GRAD
                    if (row < p.M_real) {
                        if (col < p.N_real)
                            *dst = v;
                        else if (col == p.N_real && p.gbias)
                            p.gbias[row] = v;
                    }
                }
            }
        }
    }
}

#ifdef RTX_GEMM_ABLATE
template <int ABL> static void abl_launch(const RtxGemm& g, dim3 grid, hipStream_t st)
{
    hipLaunchKernelGGL((rtx_gemm_nt<bf16_t, RTX_EPI_STORE, 2, 2, 2, 128, ABL>), grid, dim3(256), 0, st, g);
}
// measurement: the register-staged 128 x 128 split-K product with parts of its loop removed (ABL mask above)
int rtx_gemm_ablate_launch(const RtxGemm& g, int abl, hipStream_t st)
{
    const int tiles = g.m_tiles * g.n_tiles;
    const dim3 grid((unsigned)(8 * ((g.splits + 7) / 8) * tiles));
    if (abl == 100) { hipLaunchKernelGGL((rtx_gemm_nt<bf16_t, RTX_EPI_STORE, 2, 2, 2, 128, 0, 1>), grid, dim3(256), 0, st, g); RTX_HIP(hipGetLastError()); return RTX_OK; }
    if (abl == 107) { hipLaunchKernelGGL((rtx_gemm_nt<bf16_t, RTX_EPI_STORE, 2, 2, 2, 128, 7, 1>), grid, dim3(256), 0, st, g); RTX_HIP(hipGetLastError()); return RTX_OK; }
    if (abl == 103) { hipLaunchKernelGGL((rtx_gemm_nt<bf16_t, RTX_EPI_STORE, 2, 2, 2, 128, 3, 1>), grid, dim3(256), 0, st, g); RTX_HIP(hipGetLastError()); return RTX_OK; }
    switch (abl) {
    case 0: abl_launch<0>(g, grid, st); break;
    case 1: abl_launch<1>(g, grid, st); break;
    case 2: abl_launch<2>(g, grid, st); break;
    case 3: abl_launch<3>(g, grid, st); break;
    case 4: abl_launch<4>(g, grid, st); break;
    case 8: abl_launch<8>(g, grid, st); break;
    case 11: abl_launch<11>(g, grid, st); break;
    case 16: abl_launch<16>(g, grid, st); break;
    case 32: abl_launch<32>(g, grid, st); break;
    case 36: abl_launch<36>(g, grid, st); break;
    case 7: abl_launch<7>(g, grid, st); break;
    case 39: abl_launch<39>(g, grid, st); break;
    case 63: abl_launch<63>(g, grid, st); break;
    case 64: abl_launch<64>(g, grid, st); break;
    default: return RTX_EINVAL;
    }
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
#endif

void rtx_gemm_tile_dims(int shape, int* bm, int* bn)
{
    switch (shape) {
    case RTX_TILE_256x128: *bm = 256; *bn = 128; break;
    case RTX_TILE_128x256: *bm = 128; *bn = 256; break;
    default: *bm = 128; *bn = 128; break;
    }
}

template <typename T, int WM, int WN, int NB>
static void launch_shape(const RtxGemm& g, int epilogue, dim3 grid, hipStream_t stream)
{
    const dim3 block(WM * WN * 64);
    switch (epilogue) {
    case RTX_EPI_STORE: hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_STORE, WM, WN, NB>), grid, block, 0, stream, g); break;
    case RTX_EPI_BIAS_ROWS: hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_BIAS_ROWS, WM, WN, NB>), grid, block, 0, stream, g); break;
    default: hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_GRAD, WM, WN, NB>), grid, block, 0, stream, g); break;
    }
}

template <typename T> static void launch_type(const RtxGemm& g, int epilogue, dim3 grid, hipStream_t stream)
{
    if constexpr (sizeof(T) == 2) {
        if (g.tile_shape == RTX_TILE_128x128_K32) {   // (rtx_gemm_launch has checked: bias epilogue)
            hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_BIAS_ROWS, 2, 2, 2, 64>), grid, dim3(256), 0, stream, g);
            return;
        }
        if (g.tile_shape == RTX_TILE_128x128_D3) {    // three slices in flight (rtx_gemm_launch has checked: plain store or bias epilogue)
            if (epilogue == RTX_EPI_STORE) hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_STORE, 2, 2, 2, 128, 0, 2>), grid, dim3(256), 0, stream, g);
            else hipLaunchKernelGGL((rtx_gemm_nt<T, RTX_EPI_BIAS_ROWS, 2, 2, 2, 128, 0, 2>), grid, dim3(256), 0, stream, g);
            return;
        }
    }
    switch (g.tile_shape) {
    case RTX_TILE_256x128: launch_shape<T, 4, 2, 2>(g, epilogue, grid, stream); break;
    case RTX_TILE_128x256: launch_shape<T, 2, 4, 2>(g, epilogue, grid, stream); break;
    default: launch_shape<T, 2, 2, 2>(g, epilogue, grid, stream); break;
    }
}

int rtx_gemm_launch(const RtxGemm& g, int dtype, int epilogue, hipStream_t stream)
{
    RTX_CHECK(dtype >= RTX_DT_F32 && dtype <= RTX_DT_FP8, RTX_EINVAL, "gemm: bad operand type %d", dtype);
    RTX_CHECK(dtype != RTX_DT_FP8 || (epilogue == RTX_EPI_STORE && g.tile_shape == RTX_TILE_128x128), RTX_EINVAL,
              "gemm: fp8 operands only with the plain-store epilogue and the 128x128 tile (Gram matrix of binary data)");
    RTX_CHECK(g.m_tiles > 0 && g.n_tiles > 0 && g.k_slices > 0 && g.splits > 0, RTX_EINVAL, "gemm: empty problem");
    RTX_CHECK(epilogue == RTX_EPI_STORE || g.splits == 1, RTX_EINVAL, "gemm: split-K only with EPI_STORE");
    RTX_CHECK(epilogue >= RTX_EPI_STORE && epilogue <= RTX_EPI_GRAD, RTX_EINVAL, "gemm: bad epilogue %d", epilogue);
    if (g.C16) {
        int bm16, bn16;
        rtx_gemm_tile_dims(g.tile_shape, &bm16, &bn16);
        RTX_CHECK(epilogue == RTX_EPI_BIAS_ROWS && dtype == RTX_DT_BF16 && (g.ldc16 & 3) == 0 && (((uintptr_t)g.C16) & 7) == 0 &&
                      g.ldc16 >= (long)g.n_tiles * bn16,
                  RTX_EINVAL, "gemm: half-precision logits need the bias epilogue, bf16 operands and an 8-byte aligned [M][ldc16 >= N_pad] image");
    }
    RTX_CHECK(g.tile_shape >= RTX_TILE_128x128 && g.tile_shape <= RTX_TILE_128x128_D3, RTX_EINVAL, "gemm: bad tile shape %d", g.tile_shape);
    RTX_CHECK(g.tile_shape != RTX_TILE_128x128_D3 || (dtype == RTX_DT_BF16 && (epilogue == RTX_EPI_STORE || epilogue == RTX_EPI_BIAS_ROWS)), RTX_EINVAL,
              "gemm: the depth-3 tile exists for bf16 operands with the store / bias epilogues only");
    RTX_CHECK(g.tile_shape != RTX_TILE_128x128_K32 || (dtype == RTX_DT_BF16 && epilogue == RTX_EPI_BIAS_ROWS), RTX_EINVAL,
              "gemm: the 64-byte-slice tile exists for bf16 operands with the bias epilogue only");
    // 1-D grid laid out for the XCD-aware mapping in the kernel: 8 * ceil(groups / 8) * group_size workgroups
    const int tiles = g.m_tiles * g.n_tiles;
    int groups, gsize;
    if (g.syrk_lower) {
        RTX_CHECK(g.m_tiles == g.n_tiles && g.splits == 1 && g.tile_shape == RTX_TILE_128x128 && epilogue == RTX_EPI_STORE, RTX_EINVAL,
                  "gemm: syrk_lower needs a square 128x128-tiled problem without split-K");
        const int pr = (g.m_tiles + 7) / 8;
        groups = pr * (pr + 1) / 2;   // 8x8 patches of the lower triangle
        gsize = 64;
    } else if (g.splits > 1) { groups = g.splits; gsize = tiles; }
    else if (g.m_tiles <= g.n_tiles) { groups = g.n_tiles; gsize = g.m_tiles; }
    else { groups = g.m_tiles; gsize = g.n_tiles; }
    const dim3 grid((unsigned)(8 * ((groups + 7) / 8) * gsize));
    if (dtype == RTX_DT_FP8) hipLaunchKernelGGL((rtx_gemm_nt<fp8_t, RTX_EPI_STORE, 2, 2, 2>), grid, dim3(256), 0, stream, g);
    else if (dtype == RTX_DT_BF16) launch_type<bf16_t>(g, epilogue, grid, stream);
    else launch_type<float>(g, epilogue, grid, stream);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
