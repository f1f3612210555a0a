// gemm_dma.hip -- bf16 MFMA GEMM with LDS-DMA operand staging (gfx950), for the step's big contractions.
//
//   NT : C[m][n] = sum_k A[m][k] * B[n][k]     both operands K-contiguous          (forward: Act x W^T)
//   NN : C[m][n] = sum_k A[m][k] * B[k][n]     B stored K-major, N-contiguous      (backward-data: dOut x W)
//
// The NN form reads W itself ([out][in] row-major = K-major for the data gradient), so the engine keeps no transposed
// weight copy: the B fragments come out of LDS through ds_read_b64_tr_b16 (gfx950's transposing LDS read).
//
// Structure (same pipeline as the EASE Gram kernel, syrk.hip): a workgroup of WM x WN waves, every wave owning
// (MI*32) x (NJ*32) of C as MI x NJ MFMA 32x32x16 accumulators.  Operand K-slices of 64 bf16 go global -> LDS by DMA
// (global_load_lds_dwordx4: no staging registers, no ds_write pass) into a ring of NS stages; one raw s_barrier per
// slice, counted vmcnt so the next slice's DMA stays in flight across the barrier; fragment reads are inline asm with
// counted lgkmcnt (a compiler-visible LDS read behind a DMA in flight would get a vmcnt(0) that drains the ring).
//
// LDS images (the DMA writes lane-linear, so every swizzle is applied to the per-lane GLOBAL address and to the read):
//   K-contiguous operand : row R of the tile at R * 128 B, 16-byte chunk c at slot c ^ (R & 7)   -> ds_read_b128
//   K-major operand (NN B): k-row r at r * (BN*2) B, 64-byte granule c at slot c ^ (r & 3)        -> ds_read_b64_tr_b16
//     (a transposing read takes 4 consecutive k-rows x 64 B per half-wave; the XOR puts them on 4 different bank groups)
//
// The big configuration (4x2 waves x 128x64 per wave = a 512 x 128 tile) covers ALL batch rows of a B = 500 step, so
// every byte of the HBM-resident weight matrix is read by exactly one workgroup; 2 x 80 KB stages fill the CU's LDS.
//
// Epilogues go through LDS (the stages are dead by then): each wave parks 32 x 64 of C at a time in a private region and
// re-reads it row-wise -- 16-byte stores of 128-byte row pieces instead of 4-byte stores from the MFMA layout, and, for the
// logits, the online-softmax partial (max, sum exp) of each row over the wave's 64-column strip with NO cross-lane
// reduction beyond one shuffle (round 1's shuffle version cost +12 us; reference models.py:813 log_softmax).
#include "rtx_gemm.h"

#include <utility>

#ifndef GD_SCHED
#define GD_SCHED 0   // where in a slice's compute step the next slice's DMA is issued (see compute()): 0 = a quarter behind each MFMA block
#endif
// the 256x256 tiles take schedule 3 (thirds behind the first three MFMA blocks: the last piece gets a block of MFMAs to land, 4096^3:
// 885-906 vs 811-849 TF/s on four waves, 831-864 vs 784-808 on eight); the step's tiles keep 0 (its data-gradient GEMM shares the
// device with the weight kernel, where an earlier DMA burst cost more than it gained -- DESIGN 4.4)
#define GD_SCHED_BIG (GD_SCHED == 0 ? 3 : GD_SCHED)
#ifndef GD_SCHED_S2
#define GD_SCHED_S2 GD_SCHED   // the step's data-gradient tile (128x128, two stages); -DGD_SCHED_S2=3 builds the A/B variant
#endif
// measurement: shader-clock stamps of K slices 8..15 taken by wave 0 of workgroup 0 (tests/native/test_gemm.cpp "stamps"):
// [slice][0] top of the iteration, [1] my DMA pieces have landed (vmcnt), [2] past the barrier, [3] compute done; entries 32..63 the same
// for a loader wave; [64] kernel entry, [65] the 100-MHz clock there, [66] main loop done, [67] epilogue stored, [68] the 100-MHz clock there
__device__ unsigned long long* g_gd_stamps = nullptr;
__device__ int g_gd_skip = 0;   // measurement (loader-wave variant): 1 = compute waves skip the MFMAs, 2 = also the fragment reads
void rtx_gemm_dma_set_skip(int v) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gd_skip), &v, sizeof(v)); }
void rtx_gemm_dma_set_stamps(unsigned long long* dev) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_gd_stamps), &dev, sizeof(dev)); }

typedef __attribute__((ext_vector_type(8))) __bf16 gd_bf16x8;
typedef __attribute__((ext_vector_type(16))) float gd_f32x16;
typedef __attribute__((ext_vector_type(4))) float gd_f32x4;   // native vectors: arrays of them stay in registers (HIP's uint2 / float4
typedef __attribute__((ext_vector_type(4))) unsigned gd_u32x4; // structs in an array were spilled to scratch right behind the asm reads)
typedef __attribute__((ext_vector_type(2))) unsigned gd_u32x2;
typedef __attribute__((address_space(3))) unsigned char gd_lds_byte;

template <int... Is, typename F> __device__ __forceinline__ void gd_static_for_impl(std::integer_sequence<int, Is...>, F&& f)
{
    (f(std::integral_constant<int, Is>{}), ...);
}
template <int N, typename F> __device__ __forceinline__ void gd_static_for(F&& f)
{
    gd_static_for_impl(std::make_integer_sequence<int, N>{}, static_cast<F&&>(f));
}

template <int OFF> __device__ __forceinline__ void gd_rd128(gd_u32x4& dst, unsigned addr)
{
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int OFF> __device__ __forceinline__ void gd_rdtr(gd_u32x2& dst, unsigned addr)
{
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(OFF));
}
template <int N> __device__ __forceinline__ void gd_wait_lgkm()
{
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
    __builtin_amdgcn_sched_barrier(0);   // keep the MFMAs below the wait (they are register-only: "memory" does not order them)
}
template <int N> __device__ __forceinline__ void gd_wait_vm()
{
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// one k16-step's operand fragments.  The transposing reads deliver 8 bytes each: their halves are only put together at
// the MFMA, AFTER the lgkmcnt wait (a copy the compiler places right behind the asm read would move stale registers).
template <int FORM, int MI, int NJ> struct GdFrag {
    gd_u32x4 a[MI];
    gd_u32x4 b[NJ];
    gd_u32x2 blo[NJ], bhi[NJ];
};

// LW > 0: LW extra LOADER waves issue every DMA piece and the WM x WN compute waves only read fragments and issue MFMAs (a
// DMA instruction holds its wave's issue port for 60-180 cycles, during which that wave feeds the matrix pipe nothing).
template <int FORM, int EPI, int WM, int WN, int MI, int NJ, int NS, int LW = 0, int SCHED = GD_SCHED>
__global__ __launch_bounds__((WM* WN + LW) * 64, (WM * WN + LW + 3) / 4) void rtx_gemm_dma(const RtxGemm p)
{
    constexpr int NW = WM * WN, BM = WM * MI * 32, BN = WN * NJ * 32, STAGE = (BM + BN) * 128;
    constexpr int NLD = LW ? LW : NW;                     // waves that issue DMA
    constexpr int QA = BM / 8 / NLD, QB = BN / 8 / NLD;  // 1-KB DMA instructions per loading wave per slice
    constexpr int SB = BN * 2;                            // NN: bytes of one k-row of the B slice image
    constexpr int LPS = QA + QB;                          // loads per slice per wave
    constexpr int NR = MI + (FORM == RTX_FORM_NN ? 2 * NJ : NJ);   // LDS reads per fragment set
    static_assert(BM % (8 * NLD) == 0 && BN % (8 * NLD) == 0, "tile rows must split evenly over the waves' DMA blocks");
    static_assert(LW == 0 || NS == 2, "loader waves: two stages");
    static_assert(FORM == RTX_FORM_NT || SB >= 256, "NN needs at least 128 columns per tile (64-byte granule swizzle)");
    extern __shared__ __attribute__((aligned(1024))) unsigned char smem[];

    const unsigned long long t_entry = __builtin_readcyclecounter(), r_entry = __builtin_amdgcn_s_memrealtime();
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (p.hop_word && blockIdx.x == 0 && tid == 0) __hip_atomic_store(p.hop_word, p.hop_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    if (p.wait_word) rtx_fold_wait(p.wait_word, p.wait_seq);   // a cross-stream dependency folded into this kernel (rtx_gemm.h)

    const int lw = LW ? wave - NW : wave;   // index among the DMA-issuing waves
    const int wm = wave / WN, wn = wave % WN;
    const int r = lane & 31, g = lane >> 5;

    // XCD-aware work mapping (workgroup b runs on XCD b % 8, used for speed only -- see gemm.hip)
    int tm, tn, split;
    {
        const int L = blockIdx.x, xcd = L & 7, j = L >> 3;
        if (p.splits > 1) {
            const int tiles = p.m_tiles * p.n_tiles;
            split = xcd + 8 * (j / tiles);
            if (split >= p.splits) return;
            const int t = j % tiles;
            tm = t % p.m_tiles;
            tn = t / p.m_tiles;
        } else if (p.xcd_block) {
            split = 0;
            const int sbm = (p.m_tiles + 7) >> 3, sb = xcd + 8 * (j >> 5), in = j & 31;
            tm = (sb % sbm) * 8 + (in & 7);
            tn = (sb / sbm) * 4 + (in >> 3);
            if (tm >= p.m_tiles || tn >= p.n_tiles) return;
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
    const int per = (p.k_slices + p.splits - 1) / p.splits;
    const int ks0 = split * per;
    const int nk = min(ks0 + per, p.k_slices) - ks0;

    // ---- DMA source addresses (per lane, fixed for the whole K walk) ------------------------------------------------
    const size_t rowA = (size_t)p.lda * 2, rowB = (size_t)p.ldb * 2;
    const int brow = lane >> 3, chunk = (lane & 7) ^ brow;
    const unsigned char* gA = (const unsigned char*)p.A + ((size_t)tm * BM + lw * 8 + brow) * rowA + chunk * 16 + (size_t)ks0 * 128;
    const unsigned char* gB[FORM == RTX_FORM_NN ? QB : 1];
    if constexpr (FORM == RTX_FORM_NT) {
            gB[0] = (const unsigned char*)p.B + ((size_t)tn * BN + lw * 8 + brow) * rowB + chunk * 16 + (size_t)ks0 * 128;
    } else {
#pragma unroll
            for (int q = 0; q < QB; ++q) {
            const int o = (lw + q * NLD) * 1024 + lane * 16;   // physical byte of this lane's 16-byte piece in the slice image
            const int rr = o / SB, ww = o % SB;
            const int cc = (ww >> 6) ^ (rr & 3);                 // logical 64-byte granule that lives at this slot
            gB[q] = (const unsigned char*)p.B + ((size_t)ks0 * 64 + rr) * rowB + (size_t)tn * BN * 2 + (cc << 6) + (ww & 63);
            }
    }
    gd_lds_byte* lbase = (gd_lds_byte*)smem;

    // One K slice = LPS one-KB DMA instructions per wave (QA of A, QB of B).  Issuing them costs the wave 60-180 cycles
    // EACH (MI355X_MICROARCH.md): in a burst in front of the MFMAs that is 1-2 thousand cycles per slice during which the
    // matrix pipe idles (measured: 2.5 us per slice against 0.85 us of MFMA work).  So a slice's DMA is cut into four
    // parts, each issued right behind one of the four MFMA blocks of the previous slice's compute step: the matrix pipe is
    // busy with the block just queued while the wave issues the next part.
    auto load_one = [&](int stage, int t, auto qc) __attribute__((always_inline)) {
            constexpr int q = decltype(qc)::value;
            gd_lds_byte* sb = lbase + stage * STAGE + lw * 1024;
            if constexpr (q < QA) {
            __builtin_amdgcn_global_load_lds((const void*)(gA + (size_t)t * 128 + (size_t)q * NLD * 8 * rowA),
                                             (void __attribute__((address_space(3)))*)(sb + q * NLD * 1024), 16, 0, 0);
            } else if constexpr (FORM == RTX_FORM_NT) {
            constexpr int qb = q - QA;
            __builtin_amdgcn_global_load_lds((const void*)(gB[0] + (size_t)t * 128 + (size_t)qb * NLD * 8 * rowB),
                                             (void __attribute__((address_space(3)))*)(sb + BM * 128 + qb * NLD * 1024), 16, 0, 0);
            } else {
            constexpr int qb = q - QA;
            __builtin_amdgcn_global_load_lds((const void*)(gB[qb] + (size_t)t * 64 * rowB),
                                             (void __attribute__((address_space(3)))*)(sb + BM * 128 + qb * NLD * 1024), 16, 0, 0);
        }
    };
    auto load_part = [&](int stage, int t, auto partc) __attribute__((always_inline)) {
        constexpr int part = decltype(partc)::value;
        constexpr int lo = part * LPS / 4, hi = (part + 1) * LPS / 4;
        gd_static_for<hi - lo>([&](auto ic) __attribute__((always_inline)) { load_one(stage, t, std::integral_constant<int, lo + decltype(ic)::value>{}); });
        __builtin_amdgcn_sched_barrier(0);
    };
    auto load_slice = [&](int stage, int t) __attribute__((always_inline)) {
        gd_static_for<LPS>([&](auto ic) __attribute__((always_inline)) { load_one(stage, t, ic); });
    };

    unsigned long long* stamps = (blockIdx.x == 0 && (tid == 0 || tid == NW * 64)) ? g_gd_stamps : nullptr;
    if constexpr (LW > 0) {
        if (wave >= NW) {   // ---- loader waves: slice t+1 is requested right behind barrier t, which frees its stage
            if (stamps) stamps += 32;
            load_slice(0, 0);
            for (int t = 0; t < nk; ++t) {
                const bool stamp = stamps && t >= 8 && t < 16;
                if (stamp) stamps[(t - 8) * 4 + 0] = __builtin_readcyclecounter();
                gd_wait_vm<0>();
                if (stamp) stamps[(t - 8) * 4 + 1] = __builtin_readcyclecounter();
                __builtin_amdgcn_s_barrier();
                if (stamp) stamps[(t - 8) * 4 + 2] = __builtin_readcyclecounter();
                if (t + 1 < nk) load_slice((t + 1) & 1, t + 1);
                if (stamp) stamps[(t - 8) * 4 + 3] = __builtin_readcyclecounter();
            }
            __builtin_amdgcn_s_barrier();
            return;
        }
    }
    const int skip = LW ? __builtin_amdgcn_readfirstlane(g_gd_skip) : 0;
    gd_f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    // ---- fragment addressing -----------------------------------------------------------------------------------------
    // K-contiguous operands: chunk c = g + 2 kk of row R sits at slot c ^ (R & 7); R & 7 == r & 7 for every fragment row
    unsigned slk[4];
#pragma unroll
    for (int kk = 0; kk < 4; ++kk) slk[kk] = (unsigned)(((g + 2 * kk) ^ (r & 7)) * 16);
    const unsigned offA = (unsigned)((wm * MI * 32 + r) * 128);
    const unsigned offB_nt = (unsigned)((BM + wn * NJ * 32 + r) * 128);
    // K-major B (transposing reads): the 16 lanes of a group fetch a [4 k][16 n] block, lane p at k-row p >> 2,
    // 8 bytes at column (p & 3) * 4; group g4 = lane >> 4: n-half g4 & 1, k-group g4 >> 1 (MFMA B operand: lane (n, kgroup))
    unsigned offB_nn[NJ];
    {
        const int p16 = lane & 15, g4 = lane >> 4, s = p16 >> 2;
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            const int c = wn * NJ + j;   // logical 64-byte granule of this 32-column block
            offB_nn[j] = (unsigned)(BM * 128 + ((g4 >> 1) * 8 + s) * SB + ((c ^ s) << 6) + (g4 & 1) * 32 + (p16 & 3) * 8);
        }
    }

    // compute slice `stage`; when pf_t >= 0 also request slice pf_t into stage pf_stage, part by part behind the MFMA blocks
    auto compute = [&](int stage, int pf_stage, int pf_t) __attribute__((always_inline)) {
        const unsigned sbase = (unsigned)(size_t)(lbase + stage * STAGE);
        GdFrag<FORM, MI, NJ> x, y;
        auto frag = [&](GdFrag<FORM, MI, NJ>& F, auto kkc) __attribute__((always_inline)) {
            constexpr int kk = decltype(kkc)::value;
            if constexpr (FORM == RTX_FORM_NT) {
                const unsigned ab = sbase + offB_nt + slk[kk];
                gd_static_for<NJ>([&](auto jc) __attribute__((always_inline)) { gd_rd128<decltype(jc)::value * 4096>(F.b[decltype(jc)::value], ab); });
            } else {
                gd_static_for<NJ>([&](auto jc) __attribute__((always_inline)) {
                    constexpr int j = decltype(jc)::value;
                    gd_rdtr<(kk * 16) * SB>(F.blo[j], sbase + offB_nn[j]);
                    gd_rdtr<(kk * 16 + 4) * SB>(F.bhi[j], sbase + offB_nn[j]);
                });
            }
            const unsigned aa = sbase + offA + slk[kk];
            gd_static_for<MI>([&](auto ic) __attribute__((always_inline)) { gd_rd128<decltype(ic)::value * 4096>(F.a[decltype(ic)::value], aa); });
        };
        auto mma = [&](const GdFrag<FORM, MI, NJ>& F) __attribute__((always_inline)) {
#pragma unroll
            for (int j = 0; j < NJ; ++j) {
                gd_u32x4 bq;
                if constexpr (FORM == RTX_FORM_NN) { bq[0] = F.blo[j][0]; bq[1] = F.blo[j][1]; bq[2] = F.bhi[j][0]; bq[3] = F.bhi[j][1]; }
                else bq = F.b[j];
#pragma unroll
                for (int i = 0; i < MI; ++i)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(gd_bf16x8, F.a[i]), __builtin_bit_cast(gd_bf16x8, bq),
                                                                        acc[i][j], 0, 0, 0);
            }
        };
        const bool pf = pf_t >= 0;
        if constexpr (LW > 0) {
            // loader-wave variant: three waves per SIMD leave 168 registers, 128 of them accumulators -- one fragment set; the
            // SIMD's other compute wave feeds the matrix pipe while this one waits for its reads
            gd_static_for<4>([&](auto kkc) __attribute__((always_inline)) {
                if (skip < 2) {
                    frag(x, kkc);
                    gd_wait_lgkm<0>();
                }
                if (skip == 0) mma(x);
            });
            return;
        }
        if constexpr (SCHED == 0) {
            frag(x, std::integral_constant<int, 0>{});
            frag(y, std::integral_constant<int, 1>{});
            gd_wait_lgkm<NR>();
            mma(x);
            if (pf) load_part(pf_stage, pf_t, std::integral_constant<int, 0>{});
            frag(x, std::integral_constant<int, 2>{});
            gd_wait_lgkm<NR>();
            mma(y);
            if (pf) load_part(pf_stage, pf_t, std::integral_constant<int, 1>{});
            frag(y, std::integral_constant<int, 3>{});
            gd_wait_lgkm<NR>();
            mma(x);
            if (pf) load_part(pf_stage, pf_t, std::integral_constant<int, 2>{});
            gd_wait_lgkm<0>();
            mma(y);
            if (pf) load_part(pf_stage, pf_t, std::integral_constant<int, 3>{});
        } else if constexpr (SCHED == 3) {
            // three parts behind the first three MFMA blocks: the last piece has a whole block of MFMAs to land before the next wait
            auto third = [&](auto partc) __attribute__((always_inline)) {
                constexpr int part = decltype(partc)::value;
                constexpr int lo = part * LPS / 3, hi = (part + 1) * LPS / 3;
                gd_static_for<hi - lo>([&](auto ic) __attribute__((always_inline)) { load_one(pf_stage, pf_t, std::integral_constant<int, lo + decltype(ic)::value>{}); });
                __builtin_amdgcn_sched_barrier(0);
            };
            frag(x, std::integral_constant<int, 0>{});
            frag(y, std::integral_constant<int, 1>{});
            gd_wait_lgkm<NR>();
            mma(x);
            if (pf) third(std::integral_constant<int, 0>{});
            frag(x, std::integral_constant<int, 2>{});
            gd_wait_lgkm<NR>();
            mma(y);
            if (pf) third(std::integral_constant<int, 1>{});
            frag(y, std::integral_constant<int, 3>{});
            gd_wait_lgkm<NR>();
            mma(x);
            if (pf) third(std::integral_constant<int, 2>{});
            gd_wait_lgkm<0>();
            mma(y);
        } else if constexpr (SCHED == 1) {
            // front-loaded: the whole next slice is requested in the first half of this slice's compute, so every piece has at
            // least half a slice of MFMA time to land before the wait at the top of the next iteration
            frag(x, std::integral_constant<int, 0>{});
            frag(y, std::integral_constant<int, 1>{});
            if (pf) { load_part(pf_stage, pf_t, std::integral_constant<int, 0>{}); load_part(pf_stage, pf_t, std::integral_constant<int, 1>{}); }
            gd_wait_lgkm<NR>();
            mma(x);
            if (pf) { load_part(pf_stage, pf_t, std::integral_constant<int, 2>{}); load_part(pf_stage, pf_t, std::integral_constant<int, 3>{}); }
            frag(x, std::integral_constant<int, 2>{});
            gd_wait_lgkm<NR>();
            mma(y);
            frag(y, std::integral_constant<int, 3>{});
            gd_wait_lgkm<NR>();
            mma(x);
            gd_wait_lgkm<0>();
            mma(y);
        } else {
            // everything right behind the barrier (the Gram kernel's order)
            if (pf) {
                load_part(pf_stage, pf_t, std::integral_constant<int, 0>{}); load_part(pf_stage, pf_t, std::integral_constant<int, 1>{});
                load_part(pf_stage, pf_t, std::integral_constant<int, 2>{}); load_part(pf_stage, pf_t, std::integral_constant<int, 3>{});
            }
            frag(x, std::integral_constant<int, 0>{});
            frag(y, std::integral_constant<int, 1>{});
            gd_wait_lgkm<NR>();
            mma(x);
            frag(x, std::integral_constant<int, 2>{});
            gd_wait_lgkm<NR>();
            mma(y);
            frag(y, std::integral_constant<int, 3>{});
            gd_wait_lgkm<NR>();
            mma(x);
            gd_wait_lgkm<0>();
            mma(y);
        }
    };

    // ---- main loop: ring of NS stages, slices t .. t+NS-2 in flight at the top of iteration t ------------------------
    if constexpr (LW == 0) {
#pragma unroll
        for (int s = 0; s < NS - 1; ++s)
            if (s < nk) load_slice(s, s);
    }
    int stage = 0;
    for (int t = 0; t < nk; ++t) {
        const bool stamp = stamps && t >= 8 && t < 16;
        if (stamp) stamps[(t - 8) * 4 + 0] = __builtin_readcyclecounter();
        if constexpr (LW == 0) {
            if (NS == 2 || nk - t < 2) gd_wait_vm<0>();   // my loads of slice t have landed
            else gd_wait_vm<LPS>();                        //   (NS = 3: slice t+1 stays in flight)
        }
        if (stamp) stamps[(t - 8) * 4 + 1] = __builtin_readcyclecounter();
        __builtin_amdgcn_s_barrier();                  // everybody's have; everybody is done reading slice t-1
        if (stamp) stamps[(t - 8) * 4 + 2] = __builtin_readcyclecounter();
        int nst = stage + NS - 1;                      // the stage slice t-1 just released takes slice t+NS-1
        if (nst >= NS) nst -= NS;
        compute(stage, nst, (LW == 0 && t + NS - 1 < nk) ? t + NS - 1 : -1);
        if (stamp) stamps[(t - 8) * 4 + 3] = __builtin_readcyclecounter();
        stage = stage + 1 == NS ? 0 : stage + 1;
    }
    if constexpr (LW == 0) gd_wait_vm<0>();
    if (stamps && tid == 0) { stamps[64] = t_entry; stamps[65] = r_entry; stamps[66] = __builtin_readcyclecounter(); }
    __builtin_amdgcn_s_barrier();   // all fragment reads are done: the stages become the epilogue's scratch

    // ---- epilogue through LDS ------------------------------------------------------------------------------------------
    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
    constexpr int WCOLS = NJ * 32, WLD = WCOLS + 4;   // floats; +4 keeps the row-wise b128 reads conflict-free
    static_assert(NW * 32 * WLD * 4 <= NS * STAGE, "epilogue scratch must fit in the stages");
    float* wreg = (float*)smem + wave * (32 * WLD);
    const int erow = lane >> 1, ehalf = lane & 1;      // row-wise pass: two lanes per row, WCOLS / 2 columns each
    constexpr int HC = WCOLS / 2, NQ = HC / 4;
    constexpr int LPR = WCOLS / 4, RPI = 64 / LPR;    // store pass: lanes per row, rows per store instruction
    const int srow = lane / LPR, scol = (lane % LPR) * 4;
    const int col0 = tn * BN + wn * WCOLS;             // first column of this wave
    float bj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        bj[j] = 0.f;
        if (EPI == RTX_EPI_BIAS_ROWS && p.bias) {
            const int col = col0 + j * 32 + r;
            bj[j] = p.bias[col < p.N_real ? col : 0];
        }
    }
    const long ld = p.ldc;
    float* cbase = p.C + (EPI == RTX_EPI_STORE ? (size_t)split * p.slab_stride : (size_t)0);
    const bool vec_ok = ((ld & 3) == 0) && (((uintptr_t)cbase & 15) == 0);
#pragma unroll
    for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) wreg[((e & 3) + 8 * (e >> 2) + 4 * g) * WLD + j * 32 + r] = acc[i][j][e] + bj[j];
        __builtin_amdgcn_wave_barrier();
        gd_f32x4 v[NQ];
#pragma unroll
        for (int q = 0; q < NQ; ++q) v[q] = *(const gd_f32x4*)(wreg + erow * WLD + ehalf * HC + q * 4);
        __builtin_amdgcn_wave_barrier();
        const int row = tm * BM + wm * MI * 32 + i * 32 + erow;
        const int colh = col0 + ehalf * HC;
        if constexpr (EPI == RTX_EPI_BIAS_ROWS) {
            if (p.lse_part) {
                // online-softmax partial of this row over the wave's strip: the lane's HC values, then one shuffle
                float m = -INFINITY;
#pragma unroll
                for (int q = 0; q < NQ; ++q) {
                    const int c = colh + q * 4;
                    if (c + 0 < p.N_real) m = fmaxf(m, v[q].x);
                    if (c + 1 < p.N_real) m = fmaxf(m, v[q].y);
                    if (c + 2 < p.N_real) m = fmaxf(m, v[q].z);
                    if (c + 3 < p.N_real) m = fmaxf(m, v[q].w);
                }
                const float mo = __shfl_xor(m, 1, 64);
                const float mm = fmaxf(m, mo);
                float s = 0.f;
                if (mm != -INFINITY) {
#pragma unroll
                    for (int q = 0; q < NQ; ++q) {
                        const int c = colh + q * 4;
                        if (c + 0 < p.N_real) s += __expf(v[q].x - mm);
                        if (c + 1 < p.N_real) s += __expf(v[q].y - mm);
                        if (c + 2 < p.N_real) s += __expf(v[q].z - mm);
                        if (c + 3 < p.N_real) s += __expf(v[q].w - mm);
                    }
                }
                s += __shfl_xor(s, 1, 64);
                if (ehalf == 0 && row < p.M_real) p.lse_part[(size_t)row * p.lse_ld + (tn * WN + wn)] = make_float2(mm, s);
            }
        }
        // the stores take a second read of the parked block with LPR lanes side by side on one row: a store instruction writes RPI
        // whole row segments (full 128-byte lines); from the row-per-lane-pair registers above every lane's 16 bytes would
        // land on a line of their own (64 partial-line requests per instruction)
        gd_f32x4 w[32 / RPI];
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) w[it] = *(const gd_f32x4*)(wreg + (it * RPI + srow) * WLD + scol);
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int it = 0; it < 32 / RPI; ++it) {
            const int orow = tm * BM + wm * MI * 32 + i * 32 + it * RPI + srow;
            const int c = col0 + scol;
            if ((EPI == RTX_EPI_STORE) || orow < p.M_real) {
                float* dst = cbase + (size_t)orow * ld + c;
                if (EPI == RTX_EPI_STORE || (vec_ok && c + 3 < p.N_real)) {
                    *(gd_f32x4*)dst = w[it];
                } else {
                    if (c + 0 < p.N_real) dst[0] = w[it].x;
                    if (c + 1 < p.N_real) dst[1] = w[it].y;
                    if (c + 2 < p.N_real) dst[2] = w[it].z;
                    if (c + 3 < p.N_real) dst[3] = w[it].w;
                }
            }
        }
    }
    if (stamps && tid == 0) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        stamps[67] = __builtin_readcyclecounter();
        stamps[68] = __builtin_amdgcn_s_memrealtime();
    }
}

// ------------------------------------------------------------------------------------------------------------------------
void rtx_gemm_dma_tile_dims(int cfg, int* bm, int* bn)
{
    switch (cfg) {
    case RTX_DMA_512x128: *bm = 512; *bn = 128; break;
    case RTX_DMA_256x256: *bm = 256; *bn = 256; break;
    case RTX_DMA_256x256_W4: *bm = 256; *bn = 256; break;
    case RTX_DMA_256x256_LW: *bm = 256; *bn = 256; break;
    default: *bm = 128; *bn = 128; break;
    }
}

template <int FORM, int EPI, int WM, int WN, int MI, int NJ, int NS, int LW = 0, int SCHED = GD_SCHED>
static int gd_launch(const RtxGemm& g, dim3 grid, hipStream_t stream)
{
    constexpr int BM = WM * MI * 32, BN = WN * NJ * 32, LDS = NS * (BM + BN) * 128;
    static bool configured = false;
    if (!configured) {
        RTX_HIP(hipFuncSetAttribute((const void*)rtx_gemm_dma<FORM, EPI, WM, WN, MI, NJ, NS, LW, SCHED>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS));
        configured = true;
    }
    hipLaunchKernelGGL((rtx_gemm_dma<FORM, EPI, WM, WN, MI, NJ, NS, LW, SCHED>), grid, dim3((WM * WN + LW) * 64), LDS, stream, g);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

template <int FORM, int EPI> static int gd_launch_cfg(const RtxGemm& g, dim3 grid, hipStream_t stream)
{
    switch (g.tile_shape) {
    case RTX_DMA_512x128: return gd_launch<FORM, EPI, 4, 2, 4, 2, 2>(g, grid, stream);
    case RTX_DMA_256x256: return gd_launch<FORM, EPI, 2, 4, 4, 2, 2, 0, GD_SCHED_BIG>(g, grid, stream);
    case RTX_DMA_128x128_S2: return gd_launch<FORM, EPI, 2, 2, 2, 2, 2, 0, GD_SCHED_S2>(g, grid, stream);
    case RTX_DMA_256x256_W4: return gd_launch<FORM, EPI, 2, 2, 4, 4, 2, 0, GD_SCHED_BIG>(g, grid, stream);
    case RTX_DMA_256x256_LW: return gd_launch<FORM, EPI, 2, 4, 4, 2, 2, 4>(g, grid, stream);
    default: return gd_launch<FORM, EPI, 2, 2, 2, 2, 3>(g, grid, stream);
    }
}

// bf16 operands only.  g.tile_shape is an RtxDmaCfg; g.form RTX_FORM_NT / RTX_FORM_NN; epilogue RTX_EPI_STORE / RTX_EPI_BIAS_ROWS.
int rtx_gemm_dma_launch(const RtxGemm& g, int epilogue, hipStream_t stream)
{
    RTX_CHECK(g.form == RTX_FORM_NT || g.form == RTX_FORM_NN, RTX_EINVAL, "gemm_dma: form %d not supported", g.form);
    RTX_CHECK(epilogue == RTX_EPI_STORE || epilogue == RTX_EPI_BIAS_ROWS, RTX_EINVAL, "gemm_dma: bad epilogue %d", epilogue);
    RTX_CHECK(g.m_tiles > 0 && g.n_tiles > 0 && g.k_slices > 0 && g.splits > 0, RTX_EINVAL, "gemm_dma: empty problem");
    RTX_CHECK(epilogue == RTX_EPI_STORE || g.splits == 1, RTX_EINVAL, "gemm_dma: split-K only with EPI_STORE");
    RTX_CHECK(g.tile_shape >= RTX_DMA_128x128 && g.tile_shape <= RTX_DMA_256x256_LW, RTX_EINVAL, "gemm_dma: bad tile configuration %d", g.tile_shape);
    RTX_CHECK((g.splits - 1) * ((g.k_slices + g.splits - 1) / g.splits) < g.k_slices, RTX_EINVAL, "gemm_dma: %d splits leave an empty split of %d slices",
              g.splits, g.k_slices);
    const int tiles = g.m_tiles * g.n_tiles;
    int groups, gsize;
    if (g.splits > 1) { groups = g.splits; gsize = tiles; }
    else if (g.m_tiles <= g.n_tiles) { groups = g.n_tiles; gsize = g.m_tiles; }
    else { groups = g.m_tiles; gsize = g.n_tiles; }
    dim3 grid((unsigned)(8 * ((groups + 7) / 8) * gsize));
    RtxGemm gb = g;
    gb.xcd_block = g.xcd_block && g.splits == 1 && g.m_tiles >= 8 && g.n_tiles >= 4;   // thin grids keep the strip order (a 2-row grid would leave 3/4 of every block empty)
    if (gb.xcd_block) {
        const int sbs = ((g.m_tiles + 7) / 8) * ((g.n_tiles + 3) / 4);
        grid = dim3((unsigned)(8 * 32 * ((sbs + 7) / 8)));
    }
    if (g.form == RTX_FORM_NT) {
        if (epilogue == RTX_EPI_STORE) return gd_launch_cfg<RTX_FORM_NT, RTX_EPI_STORE>(gb, grid, stream);
        return gd_launch_cfg<RTX_FORM_NT, RTX_EPI_BIAS_ROWS>(gb, grid, stream);
    }
    if (epilogue == RTX_EPI_STORE) return gd_launch_cfg<RTX_FORM_NN, RTX_EPI_STORE>(gb, grid, stream);
    return gd_launch_cfg<RTX_FORM_NN, RTX_EPI_BIAS_ROWS>(gb, grid, stream);
}
