// rtx_gemm.h -- the one MFMA GEMM of the Mult-VAE/DAE path (gfx950).
//
//   C[m][n] = sum_k A[m][k] * B[n][k]            ("NT": both operands K-contiguous)
//
// Every contraction of the training step is expressed in this form by keeping the operands in the
// layouts it wants (engine.hip): forward  Act[B,in]  x W[out,in];  backward-data  dOut[B,out] x W^T[in,out];
// weight gradient  dOut^T[out,B] x Act^T[in,B].  Operand buffers are zero-padded to multiples of the
// 128x128 tile (rtx_pad), so the main loop carries no bounds checks.
//
// Tile: 128x128 (4 waves), 256x128 / 128x256 (8 waves); 64x64 or 64x128 of C per wave as 32x32 MFMA blocks.
// K is consumed in 128-BYTE slices per row (64 bf16 / 32 f32): global -> registers (16 B per lane,
// 8 lanes cover one 128-B row segment) -> LDS (144-B row stride: 16-B pad makes the ds_read_b128
// fragment reads conflict-free) -> MFMA.  Double-buffered LDS, one barrier per slice.
//   bf16 : v_mfma_f32_32x32x16_bf16  (8 bf16 per lane per operand = one ds_read_b128)
//   f32  : v_mfma_f32_32x32x2_f32    (one ds_read_b128 feeds 4 MFMAs; exact f32, parity mode)
//   fp8  : v_mfma_f32_32x32x16_fp8_fp8 (OCP e4m3; one ds_read_b128 feeds 2 MFMAs) -- only the EASE Gram matrix of
//          small-integer data, where fp8 operands and f32 accumulation are exact
#pragma once
#include "rtx_common.h"

enum RtxEpilogue {
    RTX_EPI_STORE = 0,   // C (fp32) [M_pad][ldc] (+ split * slab_stride): raw accumulators, unguarded
    RTX_EPI_BIAS_ROWS = 1,  // C[m][n] = acc + bias[n] for m < M_real, n < N_real (ldc arbitrary): logits
    RTX_EPI_GRAD = 2,    // gW[m * N_real + n] = acc (m < M_real, n < N_real); gb[m] = acc at n == N_real
};

// Adam state of the tensor a fused weight-gradient launch (RTX_DW_ADAM, dw_adam.hip) updates (all [M_real][N_real] row-major float32)
struct RtxAdamEpi {
    float* p;
    float* m;
    float* v;
    float* gkeep;        // also store the gradient here (nullable): callers that want p.grad
    void* sh;            // T [M_pad][ld_sh]   compute copy (nullable)
    void* shT;           // T [N_pad][ld_shT]  transposed compute copy (nullable)
    int ld_sh, ld_shT;
    float step_size, bc2_sqrt, beta1, beta2, eps, weight_decay;
    float lam;           // DAE: g += lam * p / ||p||, ||p||^2 read from *sumsq
    const float* sumsq;
};

enum RtxTileShape {     // workgroup tile of C; 128x128 runs 4 waves, the others 8 waves (64x64 or 64x128 per wave)
    RTX_TILE_128x128 = 0,
    RTX_TILE_256x128 = 1,
    RTX_TILE_128x256 = 2,
    RTX_TILE_128x128_D3 = 4,    // bf16, RTX_EPI_STORE / RTX_EPI_BIAS_ROWS: the 128 x 128 tile with THREE K slices in flight (three register sets)
    RTX_TILE_128x128_K32 = 3,   // bf16 + RTX_EPI_BIAS_ROWS only: 64-byte K slices, 41 KB of LDS -> three workgroups per CU (the logits product)
};
void rtx_gemm_tile_dims(int shape, int* bm, int* bn);

// operand forms.  NT: A[m][k], B[n][k] (both K-contiguous).  NN: B is [k][n] (K-major: the weight matrix itself in the data
// gradient).  TN: A is [k][m] and B is [k][n] (the weight gradient straight from the row-major activations / deltas).
enum RtxForm { RTX_FORM_NT = 0, RTX_FORM_NN = 1, RTX_FORM_TN = 2 };
// tile configurations of the LDS-DMA GEMM (gemm_dma.hip): 4-wave 128x128, 8-wave 512x128 (all rows of a B = 500 step), 8-wave 256x256
enum RtxDmaCfg { RTX_DMA_128x128 = 0, RTX_DMA_512x128 = 1, RTX_DMA_256x256 = 2, RTX_DMA_128x128_S2 = 3, RTX_DMA_256x256_W4 = 4, RTX_DMA_256x256_LW = 5 };   // _S2: two stages (64 KB of LDS: co-resident with other kernels); _W4: 256x256 tile on FOUR waves, 128x128 of C per wave (256 accumulator registers): half the LDS bytes per MFMA of the 8-wave tile
void rtx_gemm_dma_tile_dims(int cfg, int* bm, int* bn);

struct RtxGemm {
    const void* A;       // [M_pad][lda] elements of T   (TN: [K_pad][lda])
    const void* B;       // [N_pad][ldb]                 (NN / TN: [K_pad][ldb])
    int form;            // RtxForm (0 = NT)
    long lda, ldb;       // leading dimensions in elements
    long a_slice_stride, b_slice_stride;   // rtx_gemm_launch (NT, register-staged): bytes between two 128-byte K slices of a row; 0 = row-major (128)
    int tile_shape;      // RtxTileShape; M_pad / N_pad must be multiples of the tile
    int m_tiles, n_tiles;
    int k_slices;        // total 128-byte K slices  (= K_pad * sizeof(T) / 128)
    int splits;          // split-K factor (grid.y); only with RTX_EPI_STORE
    int syrk_lower;      // 1: A == B, only tiles on or below the diagonal are computed (8x8-patch workgroup order)
    float* C;
    long ldc;
    long slab_stride;    // elements between split-K slabs
    const float* bias;   // RTX_EPI_BIAS_ROWS
    float* gbias;        // RTX_EPI_GRAD (nullable)
    int M_real, N_real;
    float2* lse_part;    // RTX_EPI_BIAS_ROWS (nullable): per row, per 64-column strip (running max, sum exp) of the
    int lse_ld;          //   biased logits -> the row log-sum-exp needs no second pass over the [B, n_items] logits
    void* C16;           // RTX_EPI_BIAS_ROWS, bf16 operands, rtx_gemm_launch (nullable): the biased logits leave as IEEE HALF [M][ldc16]
    long ldc16;          //   (clamped to +-65504) INSTEAD of the float32 C -- the training step's logits, whose only reader is the
                         //   loss kernel (it overwrites them in place with the bf16 d loss / d logits); the log-sum-exp partials are
                         //   still taken from the float32 accumulators
    int xcd_block;       // gemm_dma, splits == 1: 1 = every XCD works on 8 x 4 blocks of tiles (12 operand panels per 32 tiles in its
                         //   L2 instead of the strip order's 18 or 33)
    // gemm_f32_km, splits == 1, RTX_EPI_GRAD (round 5): the LAST partial wave of tiles.  790 tiles on 256 CUs are 3 full rounds and a
    // fourth one that is 9 % full -- a quarter of the launch's time for 3 % of its work.  Tiles from index `tail_t0` on along the
    // LONG tile dimension are therefore not computed by the main grid but by `tail_splits` extra workgroups each (appended to the
    // grid), every one over 1 / tail_splits of K; their partial sums go to tail_C ([split][rows][tail_ldc], tile-local coordinates)
    // and a small reduction (rtx_launch_tail_reduce, fixed order) writes the gradient.  tail_splits <= 1: off.
    // gemm_dma (round 5): when set, thread 0 of workgroup 0 stores hop_seq to *hop_word (agent scope) before anything else -- the
    // "producer side" of a cross-stream dependency folded into the first kernel that FOLLOWS the producers on their stream: by the
    // time any workgroup of this kernel runs, the kernels before it have completed and released (in-order queue), so a consumer that
    // sees the number on another stream (k_hop_wait) may start kernels that read their output.  Costs the stream nothing, where a
    // stream memory operation or an event costs it 6-9 us (engine.hip: stream_dependency)
    unsigned* hop_word;
    unsigned hop_seq;
    // the "consumer side" folded into a kernel (rtx_gemm_nt, rtx_gemm_dma): when set, every workgroup first waits until *wait_word
    // has reached wait_seq (one lane polls at agent scope; an acquire fence only if it really had to wait -- normally the number
    // was stored long before this kernel was dispatched and the dispatch's own acquire covers the producers' output)
    const unsigned* wait_word;
    unsigned wait_seq;
    int tail_t0, tail_splits, tail_block0;
    float* tail_C;
    long tail_ldc, tail_slab_stride;
};

// operand element type.  RTX_DT_F32 = 0 and RTX_DT_BF16 = 1 keep the meaning of the former `is_bf16` flag.
enum RtxDtype { RTX_DT_F32 = 0, RTX_DT_BF16 = 1, RTX_DT_FP8 = 2 };

int rtx_gemm_launch(const RtxGemm& g, int dtype, int epilogue, hipStream_t stream);
// bf16, LDS-DMA staging, NT / NN, RTX_EPI_STORE (split-K slabs) / RTX_EPI_BIAS_ROWS (+ log-sum-exp partials); g.tile_shape = RtxDmaCfg
int rtx_gemm_dma_launch(const RtxGemm& g, int epilogue, hipStream_t stream);

// float32 operands with a K-major matrix (gemm_f32.hip): g.form RTX_FORM_NN / RTX_FORM_TN, 128x128 tiles, g.k_slices = K_pad / 32;
// RTX_EPI_STORE (split-K slabs) / RTX_EPI_GRAD
int rtx_gemm_f32_km_launch(const RtxGemm& g, int epilogue, hipStream_t stream);

// ---- weight gradient in TN form, optionally fused with the Adam update (dw_adam.hip; bf16 operands) ------------------
enum RtxDwEpilogue { RTX_DW_GRAD = 0, RTX_DW_ADAM = 1 };
enum RtxDwCfg { RTX_DW_64x128 = 0, RTX_DW_32x128 = 1, RTX_DW_32x128_S2 = 2, RTX_DW_128x128 = 3,
                RTX_DW_128x128_W4 = 4,     // round 5: 128 x 128 on FOUR waves (32 x 128 of dW per wave: 4 MFMAs per 5 fragment reads instead of
                                           //   1 per 2): the long-K tile (a batch of thousands of rows: configs[3] on one GPU)
                RTX_DW_32x256 = 5,         // round 6: 32 x 256 on eight waves, two stages (72 KB): a tile row of the optimizer state is 1 KB
                RTX_DW_32x256_S3 = 6,      //   the same with three stages (108 KB: one workgroup per CU)
                RTX_DW_32x256_K32 = 7,     //   32-row K slices, four stages (72 KB): two workgroups per CU, three slices ahead
                RTX_DW_128x128_K32 = 8 };  // round 6: 128 x 128 on eight waves, 32-row slices, four stages (64 KB, two workgroups per CU): half the operand bytes of 64 x 128 through a CU's memory queue
struct RtxDw {
    const void* A;       // delta      bf16 [K_pad][lda]: k = batch row, m = output feature (contiguous)
    const void* B;       // activation bf16 [K_pad][ldb]: n = input feature (contiguous); column N_real holds ones
    long lda, ldb;
    int m_tiles, n_tiles;   // M_pad / rtx_dw_tile_rows(cfg), ceil(N_pad / rtx_dw_tile_cols(cfg))
    int k_slices;           // K_pad / 64 (>= 2)
    int M_real, N_real;
    float* gW;           // RTX_DW_GRAD: float32 [M_real][N_real] (nullable)
    bf16_t* g16;         // RTX_DW_GRAD: bf16 image of the same (nullable): staged for a bf16 gradient exchange
    float* gbias;        // [M_real] = column N_real of the product (nullable)
    bf16_t* gbias16;     // RTX_DW_GRAD: the same as bf16 (nullable)
    RtxAdamEpi adam;     // RTX_DW_ADAM
    float* bias_p;       // RTX_DW_ADAM (nullable): the layer's bias [M_real] and its Adam moments, updated in the same launch
    float* bias_m;       //   with the scalars of `adam`
    float* bias_v;
    const float* bias_sumsq;   // DAE: squared norm of the bias tensor (nullable)
    // unused LDS added to this launch's workgroups (single-matrix launch only): with > 8 KB ONE 64 x 128 workgroup (72 KB) fits a CU instead of
    // two.  Measurement knob of the decoder matrix's launch, which runs beside the data-gradient chain: halving what it keeps in flight
    // takes the chain from 91 to 74 us and stretches the launch itself from 87 to 126 (engine option "dw_side_pad", default 0)
    int lds_pad;
    int dbg_skip;                    // measurement (rtx_dw_set_skip / rtx_dw_set_stamps fill them at launch; 0 / null otherwise)
    unsigned long long* dbg_stamps;
};
void rtx_gemm_dma_set_skip(int v);   // measurement only (gemm_dma.hip g_gd_skip)
void rtx_gemm_dma_set_stamps(unsigned long long* dev);   // measurement hook: 32 device entries (gemm_dma.hip)
void rtx_dw_set_stamps(unsigned long long* dev);   // measurement hooks (dw_adam.hip g_dw_stamps / g_dw_skip): 8 entries per workgroup
void rtx_dw_set_skip(int mask);
int rtx_dw_tile_rows(int cfg);
int rtx_dw_tile_cols(int cfg);
int rtx_dw_launch(const RtxDw& d, int epilogue, int cfg, hipStream_t stream);
#define RTX_DW_GROUP_MAX 6
// the same for up to RTX_DW_GROUP_MAX matrices in ONE launch (equal k_slices, one epilogue)
int rtx_dw_launch_group(const RtxDw* d, int n, int epilogue, int cfg, hipStream_t stream);

// Gram matrix of the EASE solver (syrk.hip): C[m][n] = sum_k A[m][k] A[n][k] for the 128-column tiles on or below the
// diagonal; A has 256 * rows256 zero-padded rows of k_slices * 128 bytes (fp8 e4m3 or bf16; element (row, slice ks) at
// A + row * row_bytes + ks * slice_bytes), C is [256 * rows256][ldc]
// with cols128 * 128 <= ldc valid columns.
int rtx_syrk_lower_launch(const void* A, long row_bytes, long slice_bytes, int rows256, int cols128, int k_slices, int fp8, float* C, long ldc,
                          hipStream_t stream);
