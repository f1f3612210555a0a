// dw_adam.hip -- weight gradient of one Linear layer fused with its torch.optim.Adam update (gfx950, bf16 operands).
//
//   dW[m][n] = sum_b D[b][m] * Act[b][n]          ("TN": both operands are the ROW-MAJOR batch matrices the forward and
//                                                  backward passes produce; no transposed copies exist any more)
//   column n == N_real of Act is a column of ones  -> dW[:, N_real] is the bias gradient
//
// Reference: autograd of nn.Linear (rectorch/nets.py:265-269) + optimizer.step() (rectorch/models.py:833, Adam built at
// models.py:768-770 / 657-659).
//
// Why this is its own kernel and not "a GEMM with an epilogue".  At a B = 500 step the contraction length is 512: the
// product needs 35 flop per byte of optimizer traffic (p, exp_avg, exp_avg_sq read and written: 24 B per parameter + the
// 2-B compute copy), a tenth of what the MFMA pipes can do per HBM byte.  The kernel is an HBM STREAMING kernel that
// happens to produce its gradient on the matrix cores:
//   * small tiles (64 x 128 or 32 x 128 of dW per workgroup), thousands of workgroups, 2-4 resident per CU, so some
//     workgroups are always in their load / store phases while others multiply;
//   * a workgroup issues the loads of its p / m / v tile FIRST (12 x 16 B per thread in flight), then walks K with the
//     operand slices arriving by LDS-DMA (global_load_lds, ring of three stages, counted vmcnt, raw s_barrier): the
//     optimizer state lands under the matrix work;
//   * operand fragments come out of the row-major slices through ds_read_b64_tr_b16 (transposing LDS read): the [k][m] /
//     [k][n] images need no transposition anywhere (64-byte granules XOR-swizzled on the DMA source side against bank
//     conflicts of the 4-row reads);
//   * the f32 gradient tile is parked in LDS and walked row-major: 512-byte runs of p / m / v per row, 16 B per lane;
//   * tiles are ordered with the SHORT matrix dimension fastest and every XCD gets one contiguous run of them, so the
//     128-byte lines that straddle neighbouring tiles (rows are 2400 B / 80432 B long: never line-aligned) are merged in
//     one L2, and the operand panel shared by a run stays in that L2.
// Measured alternatives that are not in the source any more (profiles/r2_dw_panel_experiment.log): a persistent
// variant that keeps the short-dimension operand resident as MFMA fragments in registers (8 B instead of 24 B of operand bytes
// per parameter delivered to the CUs, matrix / optimizer wave roles, p / m / v prefetched a whole tile ahead) reaches the same
// 72 us on the n_items x 600 matrix: the kernel is bound by HBM at the read / write mix of an Adam update (4.7 - 4.9 TB/s of
// measured traffic, what the stand-alone k_adam reaches as well), not by operand delivery.
// The gradient never reaches HBM (RTX_DW_ADAM).  RTX_DW_GRAD stores it instead (float32 and / or a bf16 image) for the
// data-parallel path and rtx_engine_loss_grads.  Rows that are not a multiple of 4 floats (n_items = 17 769 ...) take the same
// fused epilogue through dword-aligned 16-byte accesses (AL = false, round 3).
#include "rtx_gemm.h"
#include <type_traits>

#ifndef DW_SPLIT_PMV
#define DW_SPLIT_PMV 1   // (0: measurement build -- the whole optimizer state of a tile is requested before the K walk, as in rounds 2-3)
#endif

typedef __attribute__((ext_vector_type(8))) __bf16 dw_bf16x8;
typedef __attribute__((ext_vector_type(16))) float dw_f32x16;
typedef __attribute__((ext_vector_type(4))) float dw_f32x4;
typedef __attribute__((ext_vector_type(4))) unsigned dw_u32x4;
typedef __attribute__((ext_vector_type(2))) unsigned dw_u32x2;
typedef __attribute__((address_space(3))) unsigned char dw_lds_byte;

// measurement hooks (tests/native/test_gemm.cpp "dwx"): per-workgroup 100-MHz stamps of the tile's phases, and a mask that takes
// phases out of the kernel (1 = no K walk, 2 = no p / m / v loads, 4 = no p / m / v / compute-copy stores, 16 = the first DMA
// slices are issued BEFORE the optimizer-state loads).  They travel in the kernel arguments (RtxDw::dbg_*, null / 0 in the
// product: two scalar registers, no memory access of their own).
static unsigned long long* g_dw_stamps = nullptr;
static int g_dw_skip = 0;
void rtx_dw_set_stamps(unsigned long long* dev) { g_dw_stamps = dev; }
void rtx_dw_set_skip(int v) { g_dw_skip = v; }

template <int OFF> __device__ __forceinline__ void dw_rdtr(dw_u32x2& dst, unsigned addr)
{
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void dw_wait_lgkm()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);
}
template <int N> __device__ __forceinline__ void dw_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int NJ> struct DwFrag {
    dw_u32x2 alo, ahi, blo[NJ], bhi[NJ];
};

// p / exp_avg / exp_avg_sq are touched once per step: non-temporal loads and stores keep them from evicting the operand
// panels (D, activations) that every tile of a run re-reads from L2
#ifdef DW_PLAIN_LDST   // measurement build: default cache policy on the optimizer state
__device__ __forceinline__ dw_f32x4 dw_ld_nt(const float* p) { return *(const dw_f32x4*)p; }
__device__ __forceinline__ void dw_st_nt(float* p, dw_f32x4 v) { *(dw_f32x4*)p = v; }
#else
__device__ __forceinline__ dw_f32x4 dw_ld_nt(const float* p) { return __builtin_nontemporal_load((const dw_f32x4*)p); }
__device__ __forceinline__ void dw_st_nt(float* p, dw_f32x4 v) { __builtin_nontemporal_store(v, (dw_f32x4*)p); }
#endif
// Rows that are NOT a multiple of four floats (n_items = 17 769 ...: most item counts) start at any 4-byte offset.  gfx950 takes
// a global dwordx4 at any dword-aligned address (hipcc emits it for a 4-byte-aligned vector type), so the epilogue keeps its
// four-neighbours-per-thread layout and only the row's last, partial group goes element by element.  (Four elements 32 columns
// apart per thread with 4-byte accesses -- whole 128-byte runs per wave instruction -- measured 1.6x slower: 4x the instructions.)
typedef dw_f32x4 dw_f32x4_u __attribute__((aligned(4)));
__device__ __forceinline__ dw_f32x4 dw_ld_nt_u(const float* p) { return __builtin_nontemporal_load((const dw_f32x4_u*)p); }
__device__ __forceinline__ void dw_st_nt_u(float* p, dw_f32x4 v) { __builtin_nontemporal_store(v, (dw_f32x4_u*)p); }

// One float4 of a gradient tile -- elements (row, col .. col + 3) of the [M_real][N_real] tensor -- goes where the epilogue
// says: through Adam (pv / mv / vv = p, exp_avg, exp_avg_sq loaded from the same place, clamped when out of range) or to the
// gradient buffers.  Column N_real is the bias gradient (ones-column of the activations).
template <int EPI, bool AL = true>
__device__ __forceinline__ void dw_finish4(const RtxDw& p, int row, int col, const dw_f32x4& g4, const dw_f32x4& pv, const dw_f32x4& mv,
                                           const dw_f32x4& vv, float reg, float inv_bc2, bool nostore = false)
{
    if (row >= p.M_real) return;
    if (p.N_real >= col && p.N_real < col + 4) {
        const float gb = g4[p.N_real - col];
        if (p.gbias) p.gbias[row] = gb;
        if (p.gbias16) p.gbias16[row] = f32_to_bf16(gb);
        if constexpr (EPI == RTX_DW_ADAM) {
            if (p.bias_p) {   // the layer's bias takes its Adam step here too: no separate small-tensor launch
                const RtxAdamEpi& A = p.adam;
                const float bp0 = p.bias_p[row], bm0 = p.bias_m[row], bv0 = p.bias_v[row];
                float breg = 0.f;
                if (p.bias_sumsq && A.lam != 0.f) {
                    const float nrm = sqrtf(*p.bias_sumsq);
                    breg = nrm > 0.f ? A.lam / nrm : 0.f;
                }
                float gg = gb + breg * bp0;
                if (A.weight_decay != 0.f) gg += A.weight_decay * bp0;
                const float m1 = bm0 + (gg - bm0) * (1.f - A.beta1);
                const float v1 = bv0 * A.beta2 + (1.f - A.beta2) * gg * gg;
                const float denom = sqrtf(v1) / A.bc2_sqrt + A.eps;
                p.bias_p[row] = bp0 - A.step_size * (m1 / denom);
                p.bias_m[row] = m1;
                p.bias_v[row] = v1;
            }
        }
    }
    if (col >= p.N_real) return;
    const size_t off = (size_t)row * p.N_real + col;
    if constexpr (EPI == RTX_DW_ADAM) {
        const RtxAdamEpi& A = p.adam;
        dw_f32x4 pn, mn, vn;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float gg = g4[k] + reg * pv[k];
            if (A.weight_decay != 0.f) gg += A.weight_decay * pv[k];
            const float m1 = mv[k] + (gg - mv[k]) * (1.f - A.beta1);
            const float v1 = vv[k] * A.beta2 + (1.f - A.beta2) * gg * gg;
            // hardware square root and reciprocal (1 ulp each) instead of the correctly rounded library forms: the update term
            // step_size * m / denom (<= ~1e-2) is off by ~3e-7 of ITSELF -- three orders below the float32 spacing of the parameter
            // it is added to -- and the IEEE forms were a third of this kernel's instructions (14.3 M wave instructions per launch,
            // 23 us of VALU time on every SIMD: profiles/r4_dw_phase_probe.txt).  The float32 parity mode (k_adam) keeps them.
            const float denom = __builtin_amdgcn_sqrtf(v1) * inv_bc2 + A.eps;
            pn[k] = pv[k] - (A.step_size * m1) * __builtin_amdgcn_rcpf(denom);
            mn[k] = m1;
            vn[k] = v1;
        }
        if constexpr (!AL) {
            if (col + 4 > p.N_real) {   // the row's last group: N_real - col of its elements exist
                for (int k = 0; k < p.N_real - col; ++k) {
                    A.p[off + k] = pn[k]; A.m[off + k] = mn[k]; A.v[off + k] = vn[k];
                    if (A.gkeep) A.gkeep[off + k] = g4[k];
                    if (A.sh) ((bf16_t*)A.sh)[(size_t)row * A.ld_sh + col + k] = f32_to_bf16(pn[k]);
                    if (A.shT) ((bf16_t*)A.shT)[(size_t)(col + k) * A.ld_shT + row] = f32_to_bf16(pn[k]);
                }
                return;
            }
            dw_st_nt_u(A.p + off, pn);
            dw_st_nt_u(A.m + off, mn);
            dw_st_nt_u(A.v + off, vn);
            if (A.gkeep) *(dw_f32x4_u*)(A.gkeep + off) = g4;
        } else {
            if (nostore) {   // measurement: the arithmetic stays alive, nothing is written unless the result is a NaN
                if (!(pn[0] + mn[1] + vn[2] + pn[3] + mn[0] + vn[1] == pn[0] + mn[1] + vn[2] + pn[3] + mn[0] + vn[1])) A.p[off] = pn[0];
                return;
            }
            dw_st_nt(A.p + off, pn);
            dw_st_nt(A.m + off, mn);
            dw_st_nt(A.v + off, vn);
            if (A.gkeep) *(dw_f32x4*)(A.gkeep + off) = g4;
        }
        if (A.sh) store4<bf16_t>((bf16_t*)A.sh + (size_t)row * A.ld_sh + col, pn[0], pn[1], pn[2], pn[3]);
        if (A.shT) {   // hidden layers only (a few hundred thousand elements): the transposed compute copy of the backward chain
#pragma unroll
            for (int k = 0; k < 4; ++k) ((bf16_t*)A.shT)[(size_t)(col + k) * A.ld_shT + row] = f32_to_bf16(pn[k]);
        }
    } else {
        const bool vec = (p.N_real & 3) == 0;
        const int nv = min(4, p.N_real - col);
        if (p.gW) {
            if (vec) *(dw_f32x4*)(p.gW + off) = g4;
            else
                for (int k = 0; k < nv; ++k) p.gW[off + k] = g4[k];
        }
        if (p.g16) {
            if (vec) store4<bf16_t>(p.g16 + off, g4[0], g4[1], g4[2], g4[3]);
            else
                for (int k = 0; k < nv; ++k) p.g16[off + k] = f32_to_bf16(g4[k]);
        }
    }
}

__device__ __forceinline__ float dw_dae_reg(const RtxDw& p)
{
    if (p.adam.sumsq && p.adam.lam != 0.f) {
        const float nrm = sqrtf(*p.adam.sumsq);
        return nrm > 0.f ? p.adam.lam / nrm : 0.f;
    }
    return 0.f;
}

// WM x WN waves; a wave owns 32 x (TNC / WN) of the tile (NJ = TNC / 32 / WN accumulators): tile = (32 WM) x TNC
// TNC = 128, or 256 (round 6: the optimizer state of a tile row is then ONE KB per array and wave instruction instead of two
// 512-byte pieces of two rows -- tests/native/test_gemm.cpp "streams": the access pattern alone 57 vs 64 us on the encoder matrix)
// AL = false (fused Adam only): rows of N_real % 4 != 0 floats (dw_f32x4_u above)
// KS: batch rows per K slice (64; 32 for the 256-column tile, whose 64-row slice would be 36 KB: four 18-KB stages keep two
// workgroups on a CU with three slices ahead)
template <int WM, int WN, int NS, int EPI, bool AL = true, int TNC = 128, int KS = 64>
__device__ __forceinline__ void dw_tile(const RtxDw& p, const unsigned bid)   // bid: workgroup number within this problem's grid
{
    constexpr int NW = WM * WN, NTH = NW * 64, TM = WM * 32, NJ = TNC / 32 / WN;
    constexpr int SA = TM * 2, SBB = TNC * 2;                   // bytes of one k-row of the A / B slice images
    constexpr int ABYTES = KS * SA, STAGE = ABYTES + KS * SBB;  // 4 / 8 / 16 KB + 16 / 32 KB (KS = 64)
    constexpr int PCA = ABYTES / 1024;                          // 1-KB DMA pieces of an A slice
    // (eight waves on a 32-row tile: four pieces for eight waves -- waves 4 .. 7 fetch pieces 0 .. 3 a second time, to the same place, so that
    //  every wave has the same number of operations in flight and the counted vmcnt waits below hold for all of them)
    constexpr int QA = (PCA + NW - 1) / NW, QB = (KS * SBB / 1024) / NW, LPS = QA + QB;
    constexpr int TPR = TNC / 4;                                // epilogue: threads per tile row (16 B each)
    constexpr int RPP = NTH / TPR, NP = TM / RPP;               // tile rows per epilogue pass, passes
    static_assert(WN * NJ * 32 == TNC && (PCA % NW == 0 || NW % PCA == 0) && PCA >= 1 && (KS * SBB / 1024) % NW == 0 && (KS == 64 || KS == 32), "tile shape: whole DMA instructions per wave");
    static_assert(TM * TNC * 4 <= NS * STAGE, "the parked gradient tile must fit in the stages");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, g = lane >> 5;

    // tile order: short dimension fastest, one contiguous run per XCD (workgroup b runs on XCD b % 8: speed only)
    int tm, tn;
    {
        const int total = p.m_tiles * p.n_tiles;
        const int per_xcd = (total + 7) / 8;
        const int id = (int)(bid & 7) * per_xcd + (int)(bid >> 3);
        if (id >= total) return;
        if (p.n_tiles <= p.m_tiles) { tm = id / p.n_tiles; tn = id % p.n_tiles; }
        else { tn = id / p.m_tiles; tm = id % p.m_tiles; }
    }

    const int skip = p.dbg_skip;
    unsigned long long* stamps = p.dbg_stamps;
    if (stamps) stamps += (size_t)bid * 8;
    if (stamps && tid == 0) {
        stamps[0] = __builtin_amdgcn_s_memrealtime();
        stamps[7] = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11)) | ((unsigned long long)__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) << 32);   // HW_ID, XCC_ID
    }
    const size_t rowA = (size_t)p.lda * 2, rowB = (size_t)p.ldb * 2;
    const unsigned char* gA[QA];
#pragma unroll
    for (int q = 0; q < QA; ++q) {
        const int o = ((wave + q * NW) % PCA) * 1024 + lane * 16;   // physical byte of this lane's piece in the A slice image
        const int rr = o / SA, ww = o % SA;
        const int cc = (SA >= 256) ? ((ww >> 6) ^ (rr & 3)) : (SA == 128) ? ((ww >> 6) ^ ((rr >> 1) & 1)) : 0;
        gA[q] = (const unsigned char*)p.A + (size_t)rr * rowA + (size_t)tm * TM * 2 + (cc << 6) + (ww & 63);
    }
    const unsigned char* gB[QB];
#pragma unroll
    for (int q = 0; q < QB; ++q) {
        const int o = (wave + q * NW) * 1024 + lane * 16;
        const int rr = o / SBB, ww = o % SBB;
        const int cc = (ww >> 6) ^ (rr & 3);
        // (TNC = 256 on a row of an odd number of 128-column blocks: the last tile's upper half lies beyond the row -- its lanes fetch
        //  the row's last 16 bytes instead, values that only reach accumulator columns >= N_pad, which the epilogue drops)
        const size_t cb = (size_t)tn * SBB + (cc << 6) + (ww & 63);
        gB[q] = (const unsigned char*)p.B + (size_t)rr * rowB + (TNC > 128 ? (cb < rowB ? cb : rowB - 16) : cb);
    }
    dw_lds_byte* lbase = (dw_lds_byte*)smem;
    auto load_slice = [&](int stage, int t) __attribute__((always_inline)) {
        dw_lds_byte* sb = lbase + stage * STAGE + wave * 1024;
#pragma unroll
        for (int q = 0; q < QA; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(gA[q] + (size_t)t * KS * rowA),
                                             (void __attribute__((address_space(3)))*)(lbase + stage * STAGE + ((wave + q * NW) % PCA) * 1024), 16, 0, 0);
#pragma unroll
        for (int q = 0; q < QB; ++q)
            __builtin_amdgcn_global_load_lds((const void*)(gB[q] + (size_t)t * KS * rowB),
                                             (void __attribute__((address_space(3)))*)(sb + ABYTES + q * NW * 1024), 16, 0, 0);
    };
    const bool dma_first = (skip & 16) != 0;
    if (dma_first && !(skip & 1)) {
#pragma unroll
        for (int i = 0; i < (NS >= 3 ? NS - 1 : 1); ++i) load_slice(i, i);
    }

    // ---- optimizer state of this tile: issued before anything else, consumed after the K walk -------------------------
    const int col4 = (tid % TPR) * 4, rowl = tid / TPR;
    const int col = tn * TNC + col4;
    dw_f32x4 pv[NP], mv[NP], vv[NP];
    if (skip & 2) {
#pragma unroll
        for (int q = 0; q < NP; ++q) { pv[q] = dw_f32x4{0.1f, 0.2f, 0.3f, 0.4f}; mv[q] = dw_f32x4{0.f, 0.f, 0.f, 0.f}; vv[q] = dw_f32x4{1e-4f, 1e-4f, 1e-4f, 1e-4f}; }
    }
    // out-of-range threads load a clamped (valid) address instead of branching around the load: a branch per load makes hipcc
    // wait vmcnt(0) behind each one.  AL = false: the row's last group is partial: its thread loads the row's LAST four elements
    // (a valid address); its own are taken from them, shifted, where they are consumed (below).
    auto load_state = [&](auto q0c, auto q1c) __attribute__((always_inline)) {
        constexpr int q0 = decltype(q0c)::value, q1 = decltype(q1c)::value;
        const int colc = min(col, p.N_real - 4);
#pragma unroll
        for (int q = q0; q < q1; ++q) {
            const int rowc = min(tm * TM + q * RPP + rowl, p.M_real - 1);
            const size_t off = (size_t)rowc * p.N_real + colc;
            if constexpr (AL) {
                pv[q] = dw_ld_nt(p.adam.p + off);
                mv[q] = dw_ld_nt(p.adam.m + off);
                vv[q] = dw_ld_nt(p.adam.v + off);
            } else {
                pv[q] = dw_ld_nt_u(p.adam.p + off);
                mv[q] = dw_ld_nt_u(p.adam.m + off);
                vv[q] = dw_ld_nt_u(p.adam.v + off);
            }
        }
    };
    // Round 4: the tile's optimizer state arrives in TWO halves.  The first (passes 0 .. NP/2) leaves before the K walk as before;
    // the second leaves right behind the LAST operand slice's DMA, three slices before the walk ends, and is consumed after the
    // first half has gone through Adam and its stores: half the bytes queue in front of the first operand slice (the walk starts
    // earlier), and a workgroup's reads overlap its own writes instead of coming in one burst each (the kernel needs reads and
    // writes in flight together: DESIGN.md 4.5).  Needs the three-stage ring and at least three slices; otherwise all up front.
#ifndef DW_EARLY_PASSES
#define DW_EARLY_PASSES (NP / 2)   // (measurement builds: 0 .. NP)
#endif
    constexpr int NPH = (DW_SPLIT_PMV && EPI == RTX_DW_ADAM && NS >= 3 && NP >= 2) ? (DW_EARLY_PASSES < NP ? DW_EARLY_PASSES : NP) : NP;   // passes loaded up front
    constexpr int HL = 3 * (NP - NPH);                                                               // late load instructions per thread
    const bool late = HL > 0 && !(skip & 3) && p.k_slices * (64 / KS) >= NS;
    if (!(skip & 2)) {
        if constexpr (EPI == RTX_DW_ADAM) {
            load_state(std::integral_constant<int, 0>(), std::integral_constant<int, NPH>());
            if (!late && NPH < NP) load_state(std::integral_constant<int, NPH>(), std::integral_constant<int, NP>());
        }
    }

    // ---- fragment addressing: a transposing read hands the 16 lanes of a group a [4 k][16 x] block (lane p fetches 8 bytes
    //      at k-row p >> 2, column (p & 3) * 4; lane i receives column i of the 4 rows).  Group g4 = lane >> 4: column half
    //      g4 & 1, k-group g4 >> 1 -- the MFMA 32x32x16 operand layout (lane = (row & 31, k-group)). ---------------------------
    unsigned offA, offB[NJ];
    {
        const int p16 = lane & 15, g4 = lane >> 4, s = p16 >> 2;
        const int ca = (SA >= 256) ? (wm ^ s) : (SA == 128) ? (wm ^ (s >> 1)) : 0;   // a wave's 32 rows = one 64-byte granule
        offA = (unsigned)(((g4 >> 1) * 8 + s) * SA + (ca << 6) + (g4 & 1) * 32 + (p16 & 3) * 8);
#pragma unroll
        for (int j = 0; j < NJ; ++j)
            offB[j] = (unsigned)(ABYTES + ((g4 >> 1) * 8 + s) * SBB + (((wn * NJ + j) ^ s) << 6) + (g4 & 1) * 32 + (p16 & 3) * 8);
    }
    dw_f32x16 acc[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    constexpr int NR = 2 + 2 * NJ;   // LDS reads per fragment set

#define DW_FRAG(F, kk)                                                   \
    dw_rdtr<((kk)*16) * SA>(F.alo, sbase + offA);                         \
    dw_rdtr<((kk)*16 + 4) * SA>(F.ahi, sbase + offA);                     \
    _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                      \
        dw_rdtr<((kk)*16) * SBB>(F.blo[j], sbase + offB[j]);              \
        dw_rdtr<((kk)*16 + 4) * SBB>(F.bhi[j], sbase + offB[j]);          \
    }
#define DW_MMA(F)                                                                                                        \
    {                                                                                                                    \
        dw_u32x4 a_;                                                                                                     \
        a_[0] = F.alo[0]; a_[1] = F.alo[1]; a_[2] = F.ahi[0]; a_[3] = F.ahi[1];                                          \
        _Pragma("unroll") for (int j = 0; j < NJ; ++j) {                                                                 \
            dw_u32x4 b_;                                                                                                 \
            b_[0] = F.blo[j][0]; b_[1] = F.blo[j][1]; b_[2] = F.bhi[j][0]; b_[3] = F.bhi[j][1];                          \
            acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(dw_bf16x8, a_), __builtin_bit_cast(dw_bf16x8, b_), acc[j], 0, 0, 0); \
        }                                                                                                                \
    }

    const int nk = (skip & 1) ? 0 : p.k_slices * (64 / KS);   // >= 2 (KS = 32: >= 4)
    if (!dma_first && nk) {
#pragma unroll
        for (int i = 0; i < (NS >= 3 ? NS - 1 : 1); ++i) load_slice(i, i);
    }
    if (stamps && tid == 0) stamps[1] = __builtin_amdgcn_s_memrealtime();
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        // my pieces of slice t have landed.  Younger operations that may still fly: slice t + 1's pieces (LPS), and -- in the last two
        // iterations -- the late half of the optimizer state (HL), issued behind the last slice's DMA; p / m / v's first half is older
        // (NS - 2 younger slices in the steady state; rem - 1 of them in the last NS - 1 iterations, where the late loads fly as well)
        const int rem = nk - t;
        if (NS >= 3 && rem > NS - 1) dw_wait_vm<(NS >= 3 ? NS - 2 : 0) * LPS>();
        else if (NS >= 4 && rem == 3) { if (late) dw_wait_vm<2 * LPS + HL>(); else dw_wait_vm<2 * LPS>(); }
        else if (NS >= 3 && rem == 2) { if (late) dw_wait_vm<LPS + HL>(); else dw_wait_vm<LPS>(); }
        else { if (late && NS >= 3) dw_wait_vm<HL>(); else dw_wait_vm<0>(); }
        __builtin_amdgcn_s_barrier();         // everybody's have; everybody is done reading slice t-1
        if (stamps && tid == 0 && t == 0) stamps[2] = __builtin_amdgcn_s_memrealtime();
        if (t + NS - 1 < nk) {                // refill the stage slice t-1 just released
            int nst = stage + NS - 1;
            if (nst >= NS) nst -= NS;
            load_slice(nst, t + NS - 1);
            if constexpr (HL > 0)
                if (late && t + NS == nk) load_state(std::integral_constant<int, NPH>(), std::integral_constant<int, NP>());   // behind the LAST slice's DMA
        }
        {
            const unsigned sbase = (unsigned)(size_t)(lbase + stage * STAGE);
            DwFrag<NJ> x, y;
            DW_FRAG(x, 0)
            DW_FRAG(y, 1)
            dw_wait_lgkm<NR>();
            DW_MMA(x)
            if constexpr (KS == 64) {
                DW_FRAG(x, 2)
                dw_wait_lgkm<NR>();
                DW_MMA(y)
                DW_FRAG(y, 3)
                dw_wait_lgkm<NR>();
                DW_MMA(x)
            }
            dw_wait_lgkm<0>();
            DW_MMA(y)
        }
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
#undef DW_FRAG
#undef DW_MMA
    if (stamps && tid == 0) stamps[3] = __builtin_amdgcn_s_memrealtime();
    if (late && nk) dw_wait_vm<HL>();   // (every DMA piece has landed: the last iteration waited for it; the late state loads may still fly)
    else dw_wait_vm<0>();
    __builtin_amdgcn_s_barrier();   // every fragment read is done: the stages become the gradient tile's parking space

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    float* tile = (float*)smem;
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) tile[(wm * 32 + (e & 3) + 8 * (e >> 2) + 4 * g) * TNC + (wn * NJ + j) * 32 + r] = acc[j][e];
    __syncthreads();
    if (stamps && tid == 0) stamps[4] = __builtin_amdgcn_s_memrealtime();
    const bool nostore = (skip & 4) != 0;

    float reg = 0.f, inv_bc2 = 1.f;
    if constexpr (EPI == RTX_DW_ADAM) { reg = dw_dae_reg(p); inv_bc2 = 1.f / p.adam.bc2_sqrt; }
    int sh = 0;   // AL = false, the thread of the row's last (partial) group: its elements sit sh places up in what it loaded
    if constexpr (EPI == RTX_DW_ADAM && !AL) sh = (col < p.N_real && col > p.N_real - 4) ? col - (p.N_real - 4) : 0;
#pragma unroll
    for (int q = 0; q < NP; ++q) {
        const int lr = q * RPP + rowl;
        const dw_f32x4 g4 = *(const dw_f32x4*)(tile + lr * TNC + col4);
        if constexpr (EPI == RTX_DW_ADAM && !AL) {
            dw_f32x4 a, b, c;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int kk = min(k + sh, 3);
                a[k] = kk == 0 ? pv[q][0] : kk == 1 ? pv[q][1] : kk == 2 ? pv[q][2] : pv[q][3];
                b[k] = kk == 0 ? mv[q][0] : kk == 1 ? mv[q][1] : kk == 2 ? mv[q][2] : mv[q][3];
                c[k] = kk == 0 ? vv[q][0] : kk == 1 ? vv[q][1] : kk == 2 ? vv[q][2] : vv[q][3];
            }
            dw_finish4<EPI, AL>(p, tm * TM + lr, col, g4, a, b, c, reg, inv_bc2, nostore);
        } else {
            dw_finish4<EPI, AL>(p, tm * TM + lr, col, g4, pv[q], mv[q], vv[q], reg, inv_bc2, nostore);
        }
    }
    if (stamps) {
        if (tid == 0) stamps[5] = __builtin_amdgcn_s_memrealtime();
        dw_wait_vm<0>();
        if (tid == 0) stamps[6] = __builtin_amdgcn_s_memrealtime();
    }
}

template <int WM, int WN, int NS, int EPI, bool AL = true, int TNC = 128, int KS = 64>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64 * (WM >= 4 ? 1 : 2)) / 256) void rtx_dw_tn(const RtxDw p)
{
    dw_tile<WM, WN, NS, EPI, AL, TNC, KS>(p, blockIdx.x);
}

// Several matrices in ONE launch (same K, same tile configuration, same epilogue): problem k owns the workgroups
// [first[k], first[k + 1]) -- every count a multiple of 8, so a workgroup's XCD is the same as in a launch of its own.
// The engine folds the small layers' weight kernels into the encoder matrix's launch: beside a 1580-workgroup streaming kernel
// a 35-workgroup launch of its own waits for slots and crawls (53 + 33 us measured for two kernels that take 12 us each alone).
struct RtxDwGroup {
    int n;
    unsigned first[RTX_DW_GROUP_MAX + 1];
    RtxDw p[RTX_DW_GROUP_MAX];
};
template <int WM, int WN, int NS, int EPI, bool AL = true, int TNC = 128, int KS = 64>
__global__ __launch_bounds__(WM * WN * 64, (WM * WN * 64 * (WM >= 4 ? 1 : 2)) / 256) void rtx_dw_tn_group(const RtxDwGroup g)
{
    int k = 0;
#pragma unroll
    for (int q = 1; q < RTX_DW_GROUP_MAX; ++q)
        if (q < g.n && blockIdx.x >= g.first[q]) k = q;
    dw_tile<WM, WN, NS, EPI, AL, TNC, KS>(g.p[k], blockIdx.x - g.first[k]);   // (AL = false serves the aligned problems of the group as well)
}

// (Round 4 ran this launch as a PERSISTENT grid -- 512 resident workgroups pulling tiles from per-XCD counters, to close the ~5 us a
// slot stays empty between two workgroups.  It was 1.5x SLOWER (104-115 vs 70-76 us): resident workgroups fall into lock step -- all
// load, all multiply, all store -- and HBM serves pure read / pure write bursts at 4.6 / 4.3 TB/s against 6.3 for the mix that the
// per-tile launch's dispatch jitter maintains; in the step it also held the LDS the chain's kernels need.  profiles/r4_dw_persistent.txt.)
template <int WM, int WN, int NS, int EPI, bool AL = true, int TNC = 128, int KS = 64> static int dw_launch_group(const RtxDw* d, int n, hipStream_t stream)
{
    constexpr int TM = WM * 32, LDS = NS * (KS * TM * 2 + KS * TNC * 2);
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_dw_tn_group<WM, WN, NS, EPI, AL, TNC, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    RtxDwGroup g = {};
    g.n = n;
    unsigned total = 0;
    for (int k = 0; k < n; ++k) {
        g.first[k] = total;
        g.p[k] = d[k];
        g.p[k].dbg_skip = g_dw_skip; g.p[k].dbg_stamps = g_dw_stamps;
        total += (unsigned)(8 * ((d[k].m_tiles * d[k].n_tiles + 7) / 8));
    }
    g.first[n] = total;
    hipLaunchKernelGGL((rtx_dw_tn_group<WM, WN, NS, EPI, AL, TNC, KS>), dim3(total), dim3(WM * WN * 64), LDS, stream, g);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

template <int WM, int WN, int NS, int EPI, bool AL = true, int TNC = 128, int KS = 64> static int dw_launch(const RtxDw& d, hipStream_t stream)
{
    constexpr int TM = WM * 32, LDS = NS * (KS * TM * 2 + KS * TNC * 2);
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_dw_tn<WM, WN, NS, EPI, AL, TNC, KS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
        configured = true;
    }
    const int total = d.m_tiles * d.n_tiles;
    const dim3 grid((unsigned)(8 * ((total + 7) / 8)));
    RtxDw dd = d;
    dd.dbg_skip = g_dw_skip; dd.dbg_stamps = g_dw_stamps;
    const int lds = std::min(LDS + (d.lds_pad > 0 ? d.lds_pad : 0), 160 * 1024);   // (RtxDw::lds_pad: room for a kernel running beside this one)
    hipLaunchKernelGGL((rtx_dw_tn<WM, WN, NS, EPI, AL, TNC, KS>), grid, dim3(WM * WN * 64), lds, stream, dd);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

int rtx_dw_tile_rows(int cfg) { return cfg == RTX_DW_64x128 ? 64 : (cfg == RTX_DW_128x128 || cfg == RTX_DW_128x128_W4 || cfg == RTX_DW_128x128_K32) ? 128 : 32; }
int rtx_dw_tile_cols(int cfg) { return (cfg >= RTX_DW_32x256 && cfg <= RTX_DW_32x256_K32) ? 256 : 128; }

template <int EPI, bool AL = true> static int dw_launch_cfg(const RtxDw& d, int cfg, hipStream_t stream)
{
    switch (cfg) {
    case RTX_DW_32x128: return dw_launch<1, 4, 3, EPI, AL>(d, stream);      // 4 waves, 3 stages (60 KB): 2 workgroups per CU
    case RTX_DW_32x128_S2: return dw_launch<1, 4, 2, EPI, AL>(d, stream);   // 4 waves, 2 stages (40 KB): 4 workgroups per CU
    case RTX_DW_128x128: return dw_launch<4, 2, 2, EPI, AL>(d, stream);     // 8 waves, 2 stages (64 KB), 32 x 64 per wave: half the
                                                                             //   operand bytes per parameter of the 64-row tile
    case RTX_DW_128x128_W4: return dw_launch<4, 1, 2, EPI, AL>(d, stream);  // 4 waves, 2 stages (64 KB), 32 x 128 per wave
    case RTX_DW_32x256: return dw_launch<1, 8, 2, EPI, AL, 256>(d, stream);     // 8 waves (32 x 32 each), 2 stages (72 KB): 2 workgroups per CU
    case RTX_DW_32x256_S3: return dw_launch<1, 8, 3, EPI, AL, 256>(d, stream);  // 3 stages (108 KB): 1 workgroup per CU
    case RTX_DW_32x256_K32: return dw_launch<1, 8, 4, EPI, AL, 256, 32>(d, stream);  // 32-row slices, 4 stages (72 KB): 2 workgroups per CU, three slices ahead
    case RTX_DW_128x128_K32: return dw_launch<4, 2, 4, EPI, AL, 128, 32>(d, stream); // half the operand bytes per parameter of 64 x 128, four 16-KB stages (64 KB): 2 workgroups per CU
    default: return dw_launch<2, 4, 3, EPI, AL>(d, stream);                 // 8 waves, 3 stages (72 KB): 2 workgroups per CU
    }
}

// d.m_tiles = M_pad / rtx_dw_tile_rows(cfg), d.n_tiles = ceil(N_pad / rtx_dw_tile_cols(cfg)), d.k_slices = K_pad / 64 (K_pad a multiple of 128)
int rtx_dw_launch(const RtxDw& d, int epilogue, int cfg, hipStream_t stream)
{
    RTX_CHECK(d.A && d.B && d.m_tiles > 0 && d.n_tiles > 0 && d.k_slices >= 2, RTX_EINVAL, "dw: bad problem (%d x %d tiles, %d K slices)", d.m_tiles, d.n_tiles,
              d.k_slices);
    RTX_CHECK(epilogue == RTX_DW_GRAD || epilogue == RTX_DW_ADAM, RTX_EINVAL, "dw: bad epilogue %d", epilogue);
    RTX_CHECK(cfg >= RTX_DW_64x128 && cfg <= RTX_DW_128x128_K32, RTX_EINVAL, "dw: bad tile configuration %d", cfg);
    RTX_CHECK(d.M_real >= 1 && d.N_real >= 1, RTX_EINVAL, "dw: empty tensor");
    if (epilogue == RTX_DW_ADAM) {
        RTX_CHECK(d.N_real >= 4, RTX_EINVAL, "dw: the fused Adam epilogue needs rows of at least 4 floats (got %d)", d.N_real);
        RTX_CHECK(d.adam.p && d.adam.m && d.adam.v, RTX_EINVAL, "dw: Adam state is NULL");
        if (d.N_real & 3) return dw_launch_cfg<RTX_DW_ADAM, false>(d, cfg, stream);   // rows at any 4-byte offset: dword-aligned 16-byte accesses
        RTX_CHECK((((uintptr_t)d.adam.p | (uintptr_t)d.adam.m | (uintptr_t)d.adam.v | (uintptr_t)d.adam.gkeep) & 15) == 0, RTX_EINVAL,
                  "dw: Adam buffers must be 16-byte aligned");
        return dw_launch_cfg<RTX_DW_ADAM>(d, cfg, stream);
    }
    RTX_CHECK((d.N_real & 3) != 0 || ((((uintptr_t)d.gW) & 15) == 0 && (((uintptr_t)d.g16) & 7) == 0), RTX_EINVAL, "dw: gradient buffers must be 16-byte aligned");
    return dw_launch_cfg<RTX_DW_GRAD>(d, cfg, stream);
}

// d[0..n): same d.k_slices; small problems first keeps the big one's tail free of stragglers.  RTX_DW_GRAD groups serve the
// data-parallel step (gradients leave as the float32 / bf16 images the exchange sends).
template <int EPI, bool AL = true> static int dw_launch_group_cfg(const RtxDw* d, int n, int cfg, hipStream_t stream)
{
    switch (cfg) {
    case RTX_DW_32x128: return dw_launch_group<1, 4, 3, EPI, AL>(d, n, stream);
    case RTX_DW_32x128_S2: return dw_launch_group<1, 4, 2, EPI, AL>(d, n, stream);
    case RTX_DW_128x128: return dw_launch_group<4, 2, 2, EPI, AL>(d, n, stream);
    case RTX_DW_128x128_W4: return dw_launch_group<4, 1, 2, EPI, AL>(d, n, stream);
    case RTX_DW_32x256: return dw_launch_group<1, 8, 2, EPI, AL, 256>(d, n, stream);
    case RTX_DW_32x256_S3: return dw_launch_group<1, 8, 3, EPI, AL, 256>(d, n, stream);
    case RTX_DW_32x256_K32: return dw_launch_group<1, 8, 4, EPI, AL, 256, 32>(d, n, stream);
    case RTX_DW_128x128_K32: return dw_launch_group<4, 2, 4, EPI, AL, 128, 32>(d, n, stream);
    default: return dw_launch_group<2, 4, 3, EPI, AL>(d, n, stream);
    }
}

int rtx_dw_launch_group(const RtxDw* d, int n, int epilogue, int cfg, hipStream_t stream)
{
    RTX_CHECK(d && n >= 1 && n <= RTX_DW_GROUP_MAX, RTX_EINVAL, "dw group: 1..%d problems (got %d)", RTX_DW_GROUP_MAX, n);
    if (n == 1) return rtx_dw_launch(d[0], epilogue, cfg, stream);
    RTX_CHECK(epilogue == RTX_DW_ADAM || epilogue == RTX_DW_GRAD, RTX_EINVAL, "dw group: bad epilogue %d", epilogue);
    RTX_CHECK(cfg >= RTX_DW_64x128 && cfg <= RTX_DW_128x128_K32, RTX_EINVAL, "dw: bad tile configuration %d", cfg);
    bool any_unaligned = false;   // one matrix with rows of N_real % 4 != 0 floats: the whole launch takes the unaligned epilogue
    for (int k = 0; k < n; ++k) {
        const RtxDw& q = d[k];
        RTX_CHECK(q.A && q.B && q.m_tiles > 0 && q.n_tiles > 0 && q.k_slices >= 2 && q.k_slices == d[0].k_slices, RTX_EINVAL, "dw group: bad problem %d", k);
        RTX_CHECK(q.M_real >= 1 && q.N_real >= 1, RTX_EINVAL, "dw group: problem %d is empty", k);
        if (epilogue == RTX_DW_ADAM) {
            RTX_CHECK(q.N_real >= 4 && q.adam.p && q.adam.m && q.adam.v, RTX_EINVAL, "dw group: problem %d is not fusable", k);
            if (q.N_real & 3) any_unaligned = true;
            else
                RTX_CHECK((((uintptr_t)q.adam.p | (uintptr_t)q.adam.m | (uintptr_t)q.adam.v | (uintptr_t)q.adam.gkeep) & 15) == 0, RTX_EINVAL,
                          "dw: Adam buffers must be 16-byte aligned");
        } else {
            RTX_CHECK((q.N_real & 3) != 0 || ((((uintptr_t)q.gW) & 15) == 0 && (((uintptr_t)q.g16) & 7) == 0), RTX_EINVAL,
                      "dw: gradient buffers must be 16-byte aligned");
        }
    }
    if (epilogue == RTX_DW_ADAM) return any_unaligned ? dw_launch_group_cfg<RTX_DW_ADAM, false>(d, n, cfg, stream) : dw_launch_group_cfg<RTX_DW_ADAM>(d, n, cfg, stream);
    return dw_launch_group_cfg<RTX_DW_GRAD>(d, n, cfg, stream);
}
