// rtx_kernels.h -- launchers of the non-GEMM kernels of the Mult-VAE/DAE step (kernels.hip).
// All pointers are device pointers; every launcher enqueues on `stream` and returns RTX_OK / RTX_E*.
#pragma once
#include "rtx_common.h"

// A batch of users as rows of a CSR matrix resident in HBM.  Batch row b is CSR row
// (row_ids ? row_ids[b] : b).  values == NULL means every stored entry is 1.0 (implicit feedback).
struct RtxCsrView {
    const int64_t* indptr;
    const int32_t* indices;
    const float* values;
    const int32_t* row_ids;
    int32_t max_row_len;   // longest row of the matrix the view comes from (0 = unknown: a densified batch)
    int32_t avg_row_len;   // its mean row length, rounded up
};

// the opaque rtx_csr of include/rectorch_hip.h: a scipy-style CSR matrix resident in HBM
struct rtx_csr {
    int64_t* indptr = nullptr;
    int32_t* indices = nullptr;
    float* values = nullptr;  // nullptr -> all ones
    int64_t n_rows = 0, nnz = 0;
    int32_t n_cols = 0;
    int32_t max_row_len = 0;
};

// ---- K1: sparse user rows -> dense normalised (+dropout) input ---------------------------------------
//   X  [Bp][ldx]  row-major: forward A operand and (read K-major) weight-gradient B operand; rows >= B are
//                 written as zeros; column Iin holds ONES for b < B, so the weight-gradient product emits the
//                 bias gradient as one extra output column (the forward weight copy has a zero column there)
//   tsum[b] = sum of the TARGET row's values (s_b of the multinomial likelihood)
struct RtxGatherArgs {
    RtxCsrView in, target;
    int B, Bp, I, ldx;
    int Iin;              // input columns (= I, or I + cond_dim: trailing condition columns stay raw)
    void* X;
    float* tsum;
    int training;
    float dropout_p;
    const uint8_t* mask;  // [B][I] injected keep-mask (nullable -> Philox)
    uint64_t seed, offset;
    // Scatter form (round 4, nullable): the image is ALL ZERO except the columns the previous launch wrote, which it listed per row
    // slot (written[b * written_cap ..], n_written[b]).  The launch clears those, writes this batch's stored entries and the ones
    // column (2- / 4-byte scattered stores) and lists them again: ~150 stores per user instead of a 40-KB row of zeros.  The
    // caller guarantees the invariant (engine.hip: any other writer of the image, or a row longer than written_cap - 1, makes the
    // next launch a full rewrite) and in.max_row_len < written_cap.
    int32_t* written;
    int32_t* n_written;
    int written_cap;
};
int rtx_launch_gather(const RtxGatherArgs& a, int is_bf16, hipStream_t stream);

// ---- first encoder layer as a sparse product (spmm_in.hip; bf16 numerics) -----------------------------------------
// k_in_chunks: the batch's stored entries, normalised / dropped out exactly as k_gather does, as ONE stream of 64-entry
// chunks (item | bf16 value << 16), users back to back; desc[chunk] = user (desc[n_chunks] = -1); wsplit[0..16] = a split
// of the stream at user boundaries.  Capacity: sum_b max(1, ceil(len_b / 64)) chunks + 64 chunks of read-ahead slack.
#define RTX_SPMM_WAVES 16
struct RtxInChunksArgs {
    RtxCsrView in;
    int B, I, Iin;
    int training;
    float dropout_p;
    const uint8_t* mask;
    uint64_t seed, offset;
    uint32_t* ent;
    int32_t* desc;
    int32_t* wsplit;
    int64_t cap_chunks;  // capacity of ent / desc in chunks: a user whose chunks would not fit writes nothing (0 = unchecked)
    // optional: what k_gather<bf16> writes, from the same pass over the entries (training step)
    RtxCsrView target;   // rows whose sums go to tsum (read only if tsum != NULL)
    float* tsum;         // [Bp] nullable
    bf16_t* X;           // [Bp][ldx] nullable: dense image, rows >= B zero, column Iin = 1 for b < B
    int ldx, Bp;
};
int rtx_launch_in_chunks(const RtxInChunksArgs& a, hipStream_t stream);
// k_spmm_in: O32 / R [Bp][Np] = act(chunks x W^T + bias) with the padding and ones-column conventions of k_post (forward)
struct RtxSpmmInArgs {
    const uint32_t* ent;
    const int32_t* desc;
    const int32_t* wsplit;
    int B, Bp;
    const bf16_t* W;   // [>= N_real][ldw] compute copy of the first layer's weight matrix
    int ldw, Kin;      // Kin = input columns an entry may name (n_items + cond_dim)
    const float* bias;
    int N_real, Np, tanh_act;
    float* O32;        // nullable
    bf16_t* R;         // nullable
    int ones_col;
    uint64_t* stamps;  // nullable measurement hook: [workgroups][2 waves][4] shader-clock stamps (start, staged, summed, end)
};
size_t rtx_spmm_in_lds_bytes(int Kin);   // <= 160 KB or the launch is refused
int rtx_launch_spmm_in(const RtxSpmmInArgs& a, hipStream_t stream);

// ---- hidden layers of the forward pass in one launch each (small_layers.hip; bf16 numerics, padded input width <= 1024) ----
// Z == 0: R / O32 [Bp][Np] = act(A W^T + bias) with the conventions of k_post (forward).  Z > 0: the VAE head with the
// conventions of k_vae_fwd (W rows [0, Z) = mu, [Z, 2Z) = logvar; R = the next operand [Bp][Np = Zp]).
struct RtxSmallFwdArgs {
    const bf16_t* A;   // [Bp][lda]
    const bf16_t* W;   // [w_rows][ldw]
    int lda, ldw, w_rows;
    int B, Bp, N_real, Np, tanh_act;
    const float* bias;
    float* O32;        // hidden: nullable
    bf16_t* R;
    // VAE head
    int Z, training;
    float *mu32, *lv32, *eps32, *mu_out, *lv_out;
    const float* eps_in;
    uint64_t seed, offset;
};
bool rtx_small_fwd_ok(int K);
void rtx_small_set_kw(int kw);      // measurement knob: waves that share the K range of a 16-row block (1 or 2)
void rtx_small_set_waves(int w);   // measurement knob: 16-row waves per workgroup of the one-launch hidden layers (1, 2 or 4)
int rtx_launch_small_fwd(const RtxSmallFwdArgs& a, hipStream_t stream);
// backward through a hidden layer (Z == 0: conventions of k_post backward) or the VAE head (Z > 0: of k_vae_bwd):
// Dout [Bp][Np] from D [Bp][ld] (the layer's output gradient) and W^T [wt_rows][ld] (row n = input feature n)
struct RtxSmallBwdArgs {
    const bf16_t* D;
    const bf16_t* WT;
    int ld, wt_rows;
    int B, Bp, N_real, Np, tanh_act;
    const float* O32;  // hidden: the saved activation of the layer below [Bp][Np] (read if tanh_act)
    bf16_t* Dout;
    // VAE head
    int Z, training;
    const float *mu32, *lv32, *eps32;
    float beta, inv_batch;
};
int rtx_launch_small_bwd(const RtxSmallBwdArgs& a, hipStream_t stream);

// DataSampler densify: rows -> float32 [B][I] (ld = I), optional second matrix
int rtx_launch_csr_to_dense(const RtxCsrView& v, int B, int I, float* out, hipStream_t stream);

// ---- post kernels: fp32 GEMM output (possibly split-K slabs) -> the next GEMM's operand (row-major) -----
enum { RTX_POST_FWD = 0, RTX_POST_BWD = 1 };
struct RtxPostArgs {
    const float* C;     // [splits][Bp][ldc]
    int splits;
    long slab_stride;
    int ldc;
    int B, Bp;          // valid / padded batch rows processed by this call
    int N_real, Np;     // valid / padded feature columns
    int tanh_act;
    const float* bias;  // FWD
    float* O32;         // FWD: post-activation fp32 [Bp][Np] (nullable); BWD: input (the saved activation)
    void* R;            // row-major output  T [Bp][Np]   (nullable)
    int ones_col;       // FWD: R[b][N_real] = 1 for b < B (bias-gradient column of the next layer's weight gradient)
};
int rtx_launch_post(const RtxPostArgs& a, int mode, int is_bf16, hipStream_t stream);

// ---- VAE head -------------------------------------------------------------------------------------
struct RtxVaeFwdArgs {
    const float* C;  // [splits][Bp][ldc], columns [0,Z) = mu, [Z,2Z) = logvar (pre-bias)
    int splits;
    long slab_stride;
    int ldc;
    int B, Bp, Z, Zp;
    const float* bias;   // [2Z]
    float* mu32;         // [Bp][Z] engine copies for the backward
    float* lv32;
    float* eps32;
    float* mu_out;       // [B][Z] user outputs (nullable)
    float* lv_out;
    void* Zr;            // T [Bp][Zp], column Z = ones (b < B)
    int training;
    const float* eps_in; // [B][Z] injected (nullable -> Philox)
    uint64_t seed, offset;
};
int rtx_launch_vae_fwd(const RtxVaeFwdArgs& a, int is_bf16, hipStream_t stream);

struct RtxVaeBwdArgs {
    const float* C;  // dz [splits][Bp][ldc]
    int splits;
    long slab_stride;
    int ldc;
    int B, Bp, Z, Np;  // Np = P(2Z)
    const float* mu32;
    const float* lv32;
    const float* eps32;
    int training;
    float beta, inv_batch;
    void* D;   // T [Bp][Np]
};
int rtx_launch_vae_bwd(const RtxVaeBwdArgs& a, int is_bf16, hipStream_t stream);

// ---- loss ------------------------------------------------------------------------------------------
struct RtxLossArgs {
    const float* Y;  // logits [Bp][ldy]
    int ldy, B, I;
    const float2* part;  // per-row, per-strip (max, sumexp) partials from the logits GEMM epilogue (nullable)
    int n_strips, part_ld;
    RtxCsrView target;
    const float* tsum;
    float* lse;       // [Bp] out
    float* row_loss;  // [Bp] out
    float inv_batch;
    // VAE KL term
    const float* mu32;
    const float* lv32;
    int Z;
    float beta;
};
// loss_out[0] = sum(row_loss[0..B)) + lam * sum_t sqrt(sumsq[t]);  loss_accum[0] += the same (nullable)
int rtx_launch_reduce_loss(const float* row_loss, int B, float lam, const float* sumsq, int n_tensors,
                           float* loss_out, float* loss_accum, hipStream_t stream, uint32_t* mailbox = nullptr, uint32_t seq = 0, uint32_t tag = 0);
// Loss AND its gradient w.r.t. the logits in one pass over Y (reference models.py:813-815 + autograd of log_softmax):
//   lse_b from the strip partials the logits GEMM left (or, without them, a first pass over the row),
//   row-loss partials (their fixed-order sum is the loss),  D[b][i] = (s_b * softmax(Y_b)_i - t_bi) * inv_batch  (T = bf16 / f32, zero padding)
struct RtxDlogitsArgs {
    RtxLossArgs loss;   // Y, target, tsum, part..., outputs lse / row_loss, KL inputs
    int Bp;             // rows [B, Bp) of D are written as zeros
    void* D;            // T [Bp][ldd]
    int ldd;            // >= I, multiple of 8; columns [I, ldd) are written as zeros
    const void* Y16;    // T = bf16 only (nullable): the logits as IEEE half [Bp][ldd] INSTEAD of loss.Y (rtx_gemm_launch's C16);
                        //   may be the same buffer as D -- every thread reads its 16 bytes before it writes them
};
int rtx_launch_dlogits(const RtxDlogitsArgs& a, int is_bf16, hipStream_t stream);
// one workgroup per (user, 4096-column chunk): loss.row_loss receives [B][rtx_dlogits_chunks(ldd)] partial sums
int rtx_dlogits_chunks(int ldd);
// predict(): logits[b][i] = -inf where the input has a stored non-zero
int rtx_launch_neg_inf(const RtxCsrView& in, int B, float* logits, long ld, int n_items, hipStream_t stream);
// public loss_function on dense tensors: row_loss[b] = s*lse - <x,y>  (+ beta * KL_b)
int rtx_launch_dense_loss(const float* Y, const float* X, int B, int I, const float* mu, const float* lv, int Z,
                          float beta, float inv_batch, float* row_loss, hipStream_t stream);

// ---- dense batch -> CSR (drop-in train_batch(tr_batch, te_batch) / predict(x) with dense tensors) --
int rtx_launch_dense_count(const float* X, int B, int I, int32_t* counts, hipStream_t stream);
int rtx_launch_scan_counts(const int32_t* counts, int B, int64_t* indptr, hipStream_t stream);
int rtx_launch_dense_fill(const float* X, int B, int I, const int64_t* indptr, int32_t* indices, float* values,
                          hipStream_t stream);

// ---- fused multi-tensor Adam (+ shadow refresh) ------------------------------------------------------
#define RTX_MAX_TENSORS 32
struct RtxAdamTensor {
    float* p;
    const float* g;
    const bf16_t* g16;   // nullable: read the gradient from this bf16 image instead of g
    float* m;
    float* v;
    void* sh;    // T [rows_p][ld_sh]   compute copy, same orientation   (nullable)
    void* shT;   // T [cols_p][ld_shT]  compute copy, transposed          (nullable; the engine keeps none any more)
    const float* sumsq;  // DAE: this tensor's squared norm (g += lam * p / ||p||), nullable
    int rows, cols, ld_sh, ld_shT;
    int tile_start;  // first tile of this tensor in the launch
    int flat;        // filled by rtx_launch_adam: walked as one contiguous array in chunks of 4096 elements (see k_adam)
};
struct RtxAdamArgs {
    RtxAdamTensor t[RTX_MAX_TENSORS];
    int n;
    int total_tiles;   // filled by rtx_launch_adam
    int update;        // 0: only refresh the shadows from the master parameters
    float step_size;   // lr / (1 - beta1^t)
    float bc2_sqrt;    // sqrt(1 - beta2^t)
    float beta1, beta2, eps, weight_decay;
    float lam;         // DAE: g += lam * p / ||p||  (per-tensor norms in t[k].sumsq)
    float grad_scale;  // multiplies g before use (1.0; data-parallel averaging hooks)
};
int rtx_launch_adam(RtxAdamArgs& a, int is_bf16, hipStream_t stream);
int rtx_launch_cast_f32_bf16(const float* src, bf16_t* dst, long n, hipStream_t stream);
// float32 parity mode: split-K slabs [splits][M_pad][ldc] of a small weight-gradient product -> gW [M_real][N_real], gbias [M_real] (column N_real)
int rtx_launch_tail_reduce(const float* C, int splits, long slab_stride, long ldc, int rows, int cols, int row0, int col0, int M_real, int N_real,
                           float* gW, float* gbias, hipStream_t stream);
int rtx_launch_dw_slab_reduce(const float* C, int splits, long slab_stride, long ldc, int M_real, int N_real, float* gW, float* gbias, hipStream_t stream);
int rtx_launch_sumsq(const float* const* params_host, const long* sizes, int n, float* sumsq, hipStream_t stream);

// evaluate() on the device: exact top-kmax per score row + nDCG@k / Recall@k for each cut-off in ks (host array)
int rtx_launch_topk_metrics(const float* scores, long ld, int B, int n_items, const RtxCsrView& held, const int* ks, int n_k,
                            int kmax, double* ndcg, double* recall, int32_t* topk, hipStream_t stream, long out_ld = 0, const RtxCsrView* excl = nullptr);
