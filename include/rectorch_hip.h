/*
 * rectorch_hip.h -- C ABI of librectorch_hip.so: the MI355X (gfx950) implementation of rectorch's
 * Mult-VAE / Mult-DAE training and scoring path.
 *
 * The reference (makgyver/rectorch) is pure Python and has no FFI; its boundary for this path is the
 * Python class API.  Each entry point below names the reference code it replaces.  The Python mirror
 * of that API (rectorch_amd/{nets,models,samplers}.py) binds these symbols with ctypes and exposes
 * them as torch.library ops; INTEGRATION.md shows the binding a rectorch maintainer would add.
 *
 * Conventions
 *   - plain C, no torch types.  Pointers are HIP DEVICE pointers unless the name ends in _host.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls only enqueue work.
 *   - return 0 on success, <0 (RTX_E*) on failure; rtx_last_error() gives the message (thread local).
 *   - the caller owns every buffer it passes.  An rtx_engine owns its compute-precision weight copies
 *     ("shadows"), activations and workspaces.  One engine per device, externally serialised.
 *   - parameter tensors are float32, [out,in] row-major, in the order of net.parameters():
 *     enc W0,b0,W1,b1,... then dec W0,b0,...   (reference rectorch/nets.py:260-270, 208-217)
 */
#ifndef RECTORCH_HIP_H
#define RECTORCH_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RTX_MAX_LAYERS 8

#define RTX_VAE 0 /* MultiVAE_net: last encoder layer emits mu|logvar, reparameterised z        */
#define RTX_DAE 1 /* MultiDAE_net: tanh on every encoder layer                                  */

#define RTX_FP32 0 /* parity mode: v_mfma_f32_32x32x2_f32, exact f32 products and sums           */
#define RTX_BF16 1 /* throughput mode: v_mfma_f32_32x32x16_bf16, f32 accumulate, f32 master params */

typedef struct rtx_engine rtx_engine;
typedef struct rtx_csr rtx_csr;

typedef struct {
    int32_t n_enc;                        /* Linear layers in the encoder                          */
    int32_t n_dec;                        /* Linear layers in the decoder                          */
    int32_t enc_dims[RTX_MAX_LAYERS + 1]; /* enc_dims[0] = n_items ... enc_dims[n_enc] = latent    */
    int32_t dec_dims[RTX_MAX_LAYERS + 1]; /* dec_dims[0] = latent  ... dec_dims[n_dec] = n_items   */
    int32_t variant;                      /* RTX_VAE | RTX_DAE                                     */
    int32_t numerics;                     /* RTX_FP32 | RTX_BF16                                   */
    float dropout_p;                      /* nn.Dropout(p) on the normalised input (nets.py:392)   */
    int32_t max_batch;                    /* largest batch this engine will be given               */
    int32_t splitk;                       /* split-K factor for the two K = n_items GEMMs, 0 = auto */
    int32_t cond_dim;                     /* CMultiVAE_net (nets.py:455-480): the input rows carry cond_dim extra
                                           * columns after the n_items item columns; they enter the first encoder
                                           * layer raw (no normalisation, no dropout).  0 = MultiVAE / MultiDAE   */
} rtx_cfg;

/* A batch of users.  Either rows of a resident CSR matrix (fast path: nothing dense crosses the API)
 * or dense [batch][n_items] float32 device tensors.  With cfg.cond_dim > 0 the INPUT (csr / x_dense) has
 * n_items + cond_dim columns, the target keeps n_items and must be given (the conditioned samplers always do) (drop-in for train_batch(tr_batch, te_batch) /
 * predict(x), reference models.py:817-822, 619-624). */
typedef struct {
    const rtx_csr* csr;          /* input rows (NULL -> use x_dense)                                 */
    const int32_t* row_ids;      /* device int32 [batch] row numbers (NULL -> rows 0..batch-1)       */
    const rtx_csr* target_csr;   /* loss target rows, same row ids (NULL -> target = input)          */
    const float* x_dense;        /* device [batch][n_items]                                          */
    const float* target_dense;   /* device [batch][n_items] (NULL -> target = input)                 */
    int32_t batch;
} rtx_batch;

/* Per-step scalars: what MultiVAE.train_batch / MultiDAE + torch.optim.Adam hold (models.py:817-835,
 * 768-770, 657-659). */
typedef struct {
    float beta;                  /* VAE: KL weight after annealing (models.py:824-827)               */
    float lam;                   /* DAE: weight of sum_W ||W||_2   (models.py:702-706)               */
    float inv_batch;             /* scale of the batch mean: 1/B, or 1/B_global under data parallel  */
    float lr, beta1, beta2, eps, weight_decay; /* Adam hyper-parameters                              */
    int32_t step;                /* Adam step count t of THIS update (1-based)                       */
    int32_t flags;               /* RTX_STEP_* bits                                                   */
    uint64_t seed, offset;       /* Philox stream for dropout and the reparameterisation noise       */
    const uint8_t* dropout_mask; /* injected keep-mask [batch][n_items] (NULL -> Philox)             */
    const float* eps_noise;      /* injected N(0,1) draws [batch][latent]  (NULL -> Philox)          */
} rtx_step;

/* In bf16 numerics rtx_engine_train_step runs the Adam update of every weight matrix inside its weight-gradient
 * kernel (dw_adam.hip), so those gradients are never written to HBM; set this bit to ALSO store them in the bound
 * gradient buffers (p.grad in the Python mirror).  float32 numerics and rtx_engine_loss_grads always store them. */
#define RTX_STEP_KEEP_GRADS 1
/* Mult-DAE under data parallelism: the regulariser lam * sum ||W||_2 is the same on every rank, so only ONE rank may add
 * it to the loss it reports before the ranks' losses are summed; the others set this bit (the update itself is unchanged) */
#define RTX_STEP_NO_REG_IN_LOSS 2
/* rtx_engine_loss_grads in bf16 numerics with rtx_engine_bind_grads16 done: every gradient is written ONLY as the bf16 image the
 * data-parallel exchange sends (the float32 gradient buffers are not touched, no cast pass follows); the optimizer then reads
 * the reduced bf16 gradients (rtx_engine_apply_adam_layers / _rows with their grads_bf16 arguments) */
#define RTX_STEP_GRADS_BF16 8
/* ABI 7 -- rtx_engine_train_step (bf16, single GPU): do NOT make the caller's stream wait for the engine's side stream at the end
 * of the step.  The caller promises not to touch anything the step produces outside the engine (parameters, optimizer state, the
 * loss buffers) on any stream before its next call into this engine or rtx_engine_join(e, stream) -- every engine entry point
 * that takes a stream resolves the open join first; a following training step that starts from a prefetched batch
 * (rtx_engine_set_next_batch) resolves it INSIDE its first kernel, so two steps follow each other on the caller's stream without
 * a wait packet between them.  What an epoch loop wants (rectorch/models.py:409-419); train_batch's float (rtx_engine_wait_loss)
 * needs no join. */
#define RTX_STEP_DEFER_JOIN 16

/* called on the host right after the kernels producing the gradients of layer `layer` (its W and b)
 * have been enqueued; layers complete in reverse order (last decoder layer first).  A data-parallel
 * caller records an event here and starts the RCCL all-reduce of that bucket on a side stream. */
typedef void (*rtx_layer_cb)(int32_t layer, void* user);

const char* rtx_last_error(void);
int32_t rtx_abi_version(void);

/* ---- resident CSR matrices: replaces DataSampler's per-batch scipy row gather + toarray()
 *      (rectorch/samplers.py:97-105).  Column ids must be unique within a row. -------------------- */
int rtx_csr_upload(const int64_t* indptr_host, const int32_t* indices_host, const float* values_host,
                   int64_t n_rows, int32_t n_cols, rtx_csr** out); /* values_host NULL -> all 1.0 */
int rtx_csr_destroy(rtx_csr* m);
int rtx_csr_shape(const rtx_csr* m, int64_t* n_rows, int32_t* n_cols, int64_t* nnz);
/* dense float32 [batch][n_cols] image of the given rows: what DataSampler.__iter__ yields */
int rtx_csr_gather_dense(const rtx_csr* m, const int32_t* row_ids, int32_t batch, float* out, void* stream);

/* ---- engine ------------------------------------------------------------------------------------ */
int rtx_engine_create(const rtx_cfg* cfg, rtx_engine** out);
int rtx_engine_destroy(rtx_engine* e);
int32_t rtx_engine_n_tensors(const rtx_engine* e);
/* shape of parameter tensor t: rows = out features, cols = in features (1 for a bias) */
int rtx_engine_tensor_shape(const rtx_engine* e, int32_t t, int32_t* rows, int32_t* cols);
/* bind the float32 master parameters, gradient buffers and Adam moments (exp_avg / exp_avg_sq of
 * torch.optim.Adam's state) -- one device pointer per tensor.  grads/moments may be NULL for an
 * inference-only engine. */
int rtx_engine_bind(rtx_engine* e, float* const* params, float* const* grads, float* const* exp_avg,
                    float* const* exp_avg_sq);
/* refresh the compute-precision copies after the master parameters were changed from outside
 * (load_state_dict, reference models.py:513) */
/* bf16 gradient images, one per tensor in parameter order, same shapes as the float32 gradients (weights [out][in] row-major);
 * NULL unbinds.  Used by steps flagged RTX_STEP_GRADS_BF16. */
int rtx_engine_bind_grads16(rtx_engine* e, uint16_t* const* grads_bf16);
int rtx_engine_sync_shadows(rtx_engine* e, void* stream);

/* VAE_net.forward / AE_net.forward (nets.py:322-339, 82-91) and VAE.predict / AETrainer.predict
 * (models.py:619-625, 467-473).  logits [batch][n_items]; mu/logvar [batch][latent] (VAE, nullable).
 * training != 0 applies dropout and samples z (uses step->seed/offset or the injected draws). */
int rtx_engine_forward(rtx_engine* e, const rtx_batch* batch, int32_t training, const rtx_step* step,
                       int32_t remove_train, float* logits, float* mu, float* logvar, void* stream);
/* MultiVAE_net.encode / MultiDAE_net.encode (nets.py:394-405, 219-225):
 * VAE: out0 = mu, out1 = logvar; DAE: out0 = h, out1 unused.  [batch][latent] */
int rtx_engine_encode(rtx_engine* e, const rtx_batch* batch, int32_t training, const rtx_step* step, float* out0,
                      float* out1, void* stream);
/* MultiVAE_net.decode / MultiDAE_net.decode (nets.py:413-417, 227-233): z [batch][latent] -> logits */
int rtx_engine_decode(rtx_engine* e, const float* z, int32_t batch, float* logits, void* stream);

/* forward + loss_function + backward of train_batch (models.py:829-832): gradients land in the bound
 * grad buffers; loss_out[0] = loss (device float, nullable); loss_accum[0] += loss (nullable). */
int rtx_engine_loss_grads(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out,
                          float* loss_accum, rtx_layer_cb cb, void* user, void* stream);
/* optimizer.step() (models.py:833): fused multi-tensor Adam over the bound tensors + shadow refresh */
int rtx_engine_apply_adam(rtx_engine* e, const rtx_step* step, void* stream);
/* the same update restricted to layers [layer_lo, layer_hi) (network order, encoder first).  A data-parallel caller
 * applies it bucket by bucket, each as soon as that bucket's gradient all-reduce has landed, on its own stream, so the
 * optimizer pass of the decoder matrix hides under the exchange of the encoder matrix.  grads_bf16 (nullable): one
 * pointer per bound tensor (2 per layer, ALL layers, W then b) to a bf16 image of the reduced gradient, read instead
 * of the float32 gradient buffers (bf16 gradient exchange halves the bytes on xGMI and the Adam read). */
int rtx_engine_apply_adam_layers(rtx_engine* e, const rtx_step* step, int32_t layer_lo, int32_t layer_hi,
                                 const uint16_t* const* grads_bf16, void* stream);
/* Sharded optimizer for data parallelism (reduce-scatter -> Adam on the local rows -> all-gather of the compute copy):
 * the update of rows [row_lo, row_hi) of layer `layer`'s weight matrix (row_hi may reach into the row padding) and,
 * with with_bias, of its whole (replicated) bias.  w_grad_bf16 / b_grad_bf16 (nullable): bf16 images of the reduced
 * gradients of the weight matrix (element 0 = row 0) and of the bias, read instead of the bound float32 buffers. */
int rtx_engine_apply_adam_rows(rtx_engine* e, const rtx_step* step, int32_t layer, int32_t row_lo, int32_t row_hi,
                               int32_t with_bias, const uint16_t* w_grad_bf16, const uint16_t* b_grad_bf16, void* stream);
/* the compute copy of layer `layer`'s weight matrix in HBM: [padded_rows][ld] elements of elem_bytes bytes (bf16 or
 * float32), padded_rows a multiple of 128; valid until the next rtx_engine_train_step (which may swap buffers) */
int rtx_engine_shadow_region(rtx_engine* e, int32_t layer, void** base, int32_t* padded_rows, int32_t* ld, int32_t* elem_bytes);
/* ---- RCCL hooks: what a host that is not Python needs for the data-parallel step.  One communicator
 * per process / GPU; RCCL (librccl.so) is bound at run time on first use.  Rank 0 creates the id, the host distributes
 * its RTX_COMM_ID_BYTES bytes out of band, every rank calls rtx_comm_init.  All collectives are IN PLACE and enqueue on
 * `stream`: a step is  rtx_engine_loss_grads -> rtx_comm_allreduce[_many] of the gradient buffers (or
 * rtx_comm_reduce_scatter of a weight matrix's padded region) -> rtx_engine_apply_adam[_layers|_rows] ->
 * rtx_comm_allgather of rtx_engine_shadow_region when the optimizer is sharded. */
#define RTX_COMM_ID_BYTES 128
typedef struct rtx_comm rtx_comm;
int rtx_comm_unique_id(uint8_t* id_out /* [RTX_COMM_ID_BYTES], host */);
int rtx_comm_init(const uint8_t* id /* host */, int32_t rank, int32_t world, rtx_comm** out);
int rtx_comm_destroy(rtx_comm* c);
int rtx_comm_rank(const rtx_comm* c, int32_t* rank, int32_t* world);
/* sum over the ranks of n elements (dtype RTX_FP32 or RTX_BF16) */
int rtx_comm_allreduce(rtx_comm* c, void* buf, int64_t n, int32_t dtype, void* stream);
/* the same for several buffers in one RCCL group (the per-tensor gradient buffers of an engine); bufs / counts: HOST arrays */
int rtx_comm_allreduce_many(rtx_comm* c, void* const* bufs, const int64_t* counts, int32_t n_bufs, int32_t dtype, void* stream);
/* n_total elements = world equal blocks; afterwards block `rank` holds the sum of that block over the ranks */
int rtx_comm_reduce_scatter(rtx_comm* c, void* buf, int64_t n_total, int32_t dtype, void* stream);
/* bytes_total = world equal blocks; block r is replaced by rank r's block on every rank */
int rtx_comm_allgather(rtx_comm* c, void* buf, int64_t bytes_total, void* stream);

/* several collectives issued between start and end go to RCCL as one group (ncclGroupStart / ncclGroupEnd) */
int rtx_comm_group_start(rtx_comm* c);
int rtx_comm_group_end(rtx_comm* c);

/* ---- the data-parallel step scheduled by the ENGINE (round 3; the reference has no counterpart) ----------------
 * rtx_engine_train_step_dp is train_batch for one rank of a data-parallel job: forward + loss + backward on this rank's
 * users (step->inv_batch = 1 / GLOBAL batch), the gradient exchange and the optimizer, all enqueued by ONE call -- no host
 * callback, no per-bucket event created per step.  The gradients leave the weight-gradient kernels as the images the
 * collectives send (engine-owned exchange buffer, comm_dtype), in two buckets: the decoder matrix (+ its bias) on the engine's
 * side stream beside the data-gradient chain, everything else behind the chain on the caller's stream.
 *   sharded = 0: all-reduce, then Adam on every rank (replicated weights stay bit-identical);
 *   sharded = 1: a big weight matrix's region (rows padded to a multiple of 128: equal blocks) is reduce-scattered, Adam runs
 *                on this rank's rows only -- the optimizer's 28 B/param of HBM traffic shrink by the number of ranks -- and the
 *                compute copy is all-gathered in place; biases and small layers are all-reduced and replicated.  The float32
 *                master rows / Adam moments of the OTHER ranks' rows go stale on this rank (rtx_engine_dp_owned_rows says which
 *                rows are current); gather them before reading parameters from the host side.
 * Collectives: an RCCL communicator (rtx_comm_init), or caller-supplied functions (any transport: the tests use
 * torch.distributed/gloo), or -- emulate = 1 -- none at all: every collective is replaced by device copies of the bytes one
 * rank of `world` would move and Adam runs on 1/world of the rows: the HBM cost of rank 0's step on ONE GPU (timing only: the
 * other ranks' rows are never updated). */
typedef struct {
    /* all IN PLACE on `buf`, enqueued on `stream`; dtype RTX_FP32 | RTX_BF16; return 0 on success.
     * reduce_scatter: n_total = world equal blocks, block `rank` receives the sum; all_gather: bytes_total = world equal
     * blocks, block r is replaced by rank r's */
    int (*all_reduce)(void* ctx, void* buf, int64_t n, int32_t dtype, void* stream);
    int (*reduce_scatter)(void* ctx, void* buf, int64_t n_total, int32_t dtype, void* stream);
    int (*all_gather)(void* ctx, void* buf, int64_t bytes_total, void* stream);
    int (*group_start)(void* ctx); /* nullable */
    int (*group_end)(void* ctx);   /* nullable */
    void* ctx;
} rtx_dp_ops;
typedef struct {
    int32_t rank, world;
    int32_t sharded;       /* 0 | 1 (above) */
    int32_t comm_dtype;    /* RTX_FP32 (exact sums: parity) | RTX_BF16 (half the bytes on xGMI) */
    int32_t emulate;       /* 1: no communicator, device copies of the same size (timing of rank 0's step on one GPU) */
    rtx_comm* comm;        /* RCCL communicator, or NULL when ops / emulate is given */
    const rtx_dp_ops* ops; /* caller-supplied collectives (copied), or NULL */
    /* ABI 7: the collectives of bucket A (the decoder matrix, issued on the engine's SIDE stream beside the data-gradient
     * chain) go through a communicator / function table of their own, so that RCCL's per-communicator serialisation cannot
     * couple them with bucket B's collectives on the caller's stream (with ONE communicator used from two streams RCCL runs
     * the operations in issue order: bucket B's reduce would wait for bucket A's all-gather).  NULL = bucket A shares
     * comm / ops (the single-communicator schedule of ABI 5-6).  Every rank must make the same choice. */
    rtx_comm* comm_side;        /* second RCCL communicator of the same ranks (rtx_comm_init with a second id), or NULL */
    const rtx_dp_ops* ops_side; /* caller-supplied collectives for the side stream (copied), or NULL */
    int32_t shard_min_elems;    /* sharded = 1: weight matrices of at least this many elements are sharded; 0 = the engine's
                                 * default (2^20, or what the option "dp_shard_min_elems" says) */
} rtx_dp_cfg;
int rtx_engine_dp_attach(rtx_engine* e, const rtx_dp_cfg* cfg /* NULL detaches */);
int rtx_engine_train_step_dp(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out, float* loss_accum,
                             void* stream);
/* rows [row_lo, row_hi) of layer `layer`'s weight matrix whose float32 master / Adam state this rank keeps current
 * (everything when the layer is not sharded); sharded_out: 1 if the layer is sharded */
int rtx_engine_dp_owned_rows(const rtx_engine* e, int32_t layer, int32_t* row_lo, int32_t* row_hi, int32_t* sharded_out);

/* float32 -> bfloat16 (round to nearest even) of n contiguous elements: stages a gradient bucket for a bf16 all-reduce */
int rtx_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream);
/* ABI 7 -- THIS step's loss without draining the stream (the reference's train_batch ends in `return loss.item()`,
 * models.py:835).  With the mailbox enabled the loss reduction of every training step ALSO stores {loss, ticket, step->step} into
 * coherent host memory owned by the engine (the ticket is the engine's own monotonic count of reductions: a step count that restarts
 * or repeats cannot match a stale entry); rtx_engine_wait_loss(step) spins on the host until the mailbox carries the ticket of the
 * LAST step enqueued, checks that this is step `step` (RTX_ESTATE otherwise) and returns its loss: the kernels behind the loss
 * (weight gradients, Adam) keep running and the host can enqueue the next step under them.  timeout_s <= 0: 60 s.  Steps are
 * waited for in order, each before the next is enqueued (the mailbox holds the LAST step's loss). */
int rtx_engine_loss_mailbox(rtx_engine* e, int32_t enable);
int rtx_engine_wait_loss(rtx_engine* e, int32_t step, float* loss_host, double timeout_s);
/* ABI 7 -- announce the batch of the training step AFTER the next rtx_engine_train_step call (and its dropout stream: seed /
 * offset / dropout_mask of next_step; the other fields are ignored).  That call then also gathers the announced batch, on the
 * engine's side stream under its last weight-gradient + Adam launch, into a second batch image; the step that is then given
 * exactly this batch (same csr / target_csr / row_ids pointers, batch, seed, offset, mask) starts with the first-layer product.
 * A hint only: any other batch is gathered by its own step.  The announced row ids must stay unchanged until that step.  NULL
 * cancels.  bf16 numerics, resident CSR batches; the single-GPU fused step gathers under its last weight kernel, the data-parallel
 * step (rtx_engine_train_step_dp, since round 6) behind bucket A on its side stream.  The reference densifies every batch on the
 * host (samplers.py:99-100). */
int rtx_engine_set_next_batch(rtx_engine* e, const rtx_batch* next, const rtx_step* next_step);
/* resolves a join left open by a step flagged RTX_STEP_DEFER_JOIN: `stream` continues only after everything that step put on
 * the engine's side stream (no-op when nothing is open) */
int rtx_engine_join(rtx_engine* e, void* stream);
/* both of the above: one full train_batch */
int rtx_engine_train_step(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out,
                          float* loss_accum, void* stream);

/* MultiVAE.loss_function / MultiDAE's likelihood term on dense tensors (models.py:813-815):
 * loss_out[0] = mean_b(s_b*LSE_b - <x_b, y_b>) + beta * KLD   (mu/logvar NULL -> no KL term) */
int rtx_multinomial_loss(const float* recon, const float* x, int32_t batch, int32_t n_items, const float* mu,
                         const float* logvar, int32_t latent, float beta, float* loss_out, void* stream);

/* the regulariser of MultiDAE.loss_function (models.py:702-706): out[0] = sum_t ||tensor_t||_2.
 * `tensors` is a HOST array of n device pointers, `sizes` a HOST array of element counts. */
int rtx_sum_l2_norms(const float* const* tensors, const int64_t* sizes, int32_t n, float* out, void* stream);

/* Device-side consumer of predict() for evaluation.evaluate (rectorch/evaluation.py:100-106 + rectorch/metrics.py:
 * 136-147, 187-196): per user the exact top-max(ks) of the score row, then nDCG@k and Recall@k for every cut-off
 * against that user's held-out CSR row (column ids sorted).  ndcg / recall: device double [n_k][batch] (nullable);
 * topk_idx: device int32 [batch][max(ks, kmax)] sorted by descending score (nullable).  ks_host is a HOST array. */
/* evaluation.evaluate's whole loop (rectorch/evaluation.py:100-106) in ONE call: for every batch i, the users
 * row_ids[batch_offsets[i] .. batch_offsets[i + 1]) (device int32; batch_offsets is a HOST array of n_batches + 1 entries) are scored
 * in eval mode from their rows of `train`, their train items set to -inf (predict(remove_train=True)), and reduced to nDCG@k /
 * Recall@k against their rows of `heldout`.  scores_scratch: device float32 [max batch][n_items], overwritten batch after batch.
 * ndcg / recall: device double [n_k][total users] (nullable), user j of the loader in column j.  ks_host is a HOST array. */
int rtx_engine_evaluate_topk(rtx_engine* e, const rtx_csr* train, const rtx_csr* heldout, const int32_t* row_ids,
                             const int64_t* batch_offsets, int32_t n_batches, const int32_t* ks_host, int32_t n_k,
                             float* scores_scratch, double* ndcg, double* recall, void* stream);

int rtx_topk_metrics(const float* scores, int64_t ld, int32_t batch, int32_t n_items, const rtx_csr* heldout,
                     const int32_t* row_ids, const int32_t* ks_host, int32_t n_k, double* ndcg, double* recall,
                     int32_t* topk_idx, int32_t kmax, void* stream);

/* ---- EASE closed-form model (rectorch/models.py:1003-1069) ------------------------------------------
 * rtx_ease_fit replaces EASE.train (models.py:1015-1025: G = X^T X; G[diag] += lam; P = inv(G); B = P / (-diag P);
 * B[diag] = 0) with the Gram matrix, a blocked f64 Cholesky, the triangular inverse and P = W^T W all on f64 MFMA
 * (Gram matrix on fp8 / bf16 MFMA when that is exact: data that are integers after a power-of-two scale <= 8, with
 * max|s x|^2 * n_users < 2^24; float64 MFMA otherwise).  The item-item
 * matrix B stays in HBM as double [n_items][n_items]; rtx_ease_scores replaces `model = X.dot(B)` + the look-up of
 * EASE.predict (models.py:1025, 1054-1057): rows `row_ids` (nullable = 0..batch-1) of X times B into out (device
 * double [batch][n_items]), with -inf at the non-zero entries of row b (or mask_row_ids[b]) of `mask` when given.
 * Returns RTX_EINVAL when X^T X + lam I is not positive definite (lam <= 0 on rank-deficient data). */
typedef struct rtx_ease rtx_ease;
int rtx_ease_fit(const rtx_csr* X, double lam, rtx_ease** out, void* stream);
int rtx_ease_destroy(rtx_ease* h);
int rtx_ease_weights(const rtx_ease* h, double** B_dev, int32_t* n_items);
/* device-to-device copy of B into a caller buffer of n_items * n_items doubles */
int rtx_ease_copy_weights(const rtx_ease* h, double* dst_dev, void* stream);
int rtx_ease_scores(const rtx_ease* h, const rtx_csr* X, const int32_t* row_ids, int32_t batch, const rtx_csr* mask,
                    const int32_t* mask_row_ids, double* out, void* stream);
/* HIP-event durations of the last fit (ms; any pointer nullable): whole fit, Gram matrix, chol_ms = Cholesky +
 * inverse of the factor, inv_ms = P = W^T W */
int rtx_ease_timings(const rtx_ease* h, double* fit_ms, double* gram_ms, double* chol_ms, double* inv_ms);

/* ---- SVAE: sequential VAE, one user sequence per optimizer step (rectorch/nets.py:624-693,
 * rectorch/models.py:1581-1635) ---------------------------------------------------------------------------------------
 * Parameter tensors in the order of SVAE_net.parameters(): enc W,b ... dec W,b ..., item_embed.weight [n_items][embed],
 * gru.weight_ih_l0 [3R][embed], gru.weight_hh_l0 [3R][R], gru.bias_ih_l0 [3R], gru.bias_hh_l0 [3R] (gate order r|z|n). */
typedef struct rtx_svae rtx_svae;
typedef struct {
    int32_t n_items, embed_size, rnn_size;
    int32_t n_enc, n_dec;
    int32_t enc_dims[RTX_MAX_LAYERS + 1]; /* enc_dims[0] = rnn_size ... enc_dims[n_enc] = latent */
    int32_t dec_dims[RTX_MAX_LAYERS + 1]; /* dec_dims[0] = latent   ... dec_dims[n_dec] = n_items */
    int32_t max_len;                      /* longest input sequence (time steps) */
} rtx_svae_cfg;
int rtx_svae_create(const rtx_svae_cfg* cfg, rtx_svae** out);
int rtx_svae_destroy(rtx_svae* s);
/* "gemm_bf16" (0 / 1, default 0): every matrix product of the model (input projection, encoder / decoder layers, their data and
 * weight gradients) takes its operands rounded to bf16 and accumulates in float32 on the bf16 MFMA -- the dtype BASELINE.json
 * configs[4] names; the GRU recurrences, the losses, the master weights and Adam stay float32.  Default: float32 products (the
 * reference's arithmetic, ~1e-6 parity). */
/* ABI 8 -- THIS step's loss without draining the stream (rtx_engine_loss_mailbox / rtx_engine_wait_loss for the sequence model):
 * once enabled, every rtx_svae_train_step / rtx_svae_train_pack also stores its loss into coherent host memory as soon as it is
 * final -- before the backward recurrence -- and rtx_svae_wait_loss returns the loss of the LAST step enqueued (steps are waited
 * for in order).  Reference: `return loss.item()` at the end of train_batch (rectorch/models.py:835). */
int rtx_svae_loss_mailbox(rtx_svae* s, int32_t enable);
int rtx_svae_wait_loss(rtx_svae* s, float* loss_host, double timeout_s);
int rtx_svae_set_option(rtx_svae* s, const char* key, int32_t value);
int32_t rtx_svae_n_tensors(const rtx_svae* s);
int rtx_svae_tensor_shape(const rtx_svae* s, int32_t t, int32_t* rows, int32_t* cols);
int rtx_svae_bind(rtx_svae* s, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq);
/* SVAE_net.forward (nets.py:666-676) on one sequence `items` (device int32 [T]); z is always sampled (eps_noise: injected
 * N(0,1) draws [T][latent], NULL -> Philox(seed, offset)).  Any output may be NULL: logits_all [T][n_items], logits_last
 * [n_items] = recon_x[:, -1, :] with -inf at the items of the sequence when remove_train (SVAE.predict,
 * models.py:1628-1635), mu / logvar [T][latent]. */
int rtx_svae_forward(rtx_svae* s, const int32_t* items, int32_t T, const float* eps_noise, uint64_t seed, uint64_t offset,
                     int32_t remove_train, float* logits_all, float* logits_last, float* mu, float* logvar, void* stream);
/* MultiVAE.train_batch with SVAE.loss_function and the SVAE optimizer (models.py:817-835, 1622-1626, 1618-1620):
 * forward, loss = sum_t NLL_t * step->inv_batch + step->beta * mean_t KL_t (inv_batch = 1 / number of ones in the
 * target), backward (decoder, encoder head, GRU through time, embedding), Adam with step->weight_decay.  The target is
 * either a CSR over the T steps (device int64 indptr [T+1], int32 indices, implicit ones) or a dense device [T][n_items]. */
int rtx_svae_train_step(rtx_svae* s, const int32_t* items, int32_t T, const int64_t* target_indptr, const int32_t* target_indices,
                        const float* target_dense, const rtx_step* step, float* loss_out, float* loss_accum, void* stream);
/* NOT in the reference (which takes one Adam step per user; samplers.py:517-571, models.py:1609-1635): `n_seq` user sequences
 * in ONE optimizer step.  The sequences are concatenated -- items [total_steps], rows [seq_ptr[u], seq_ptr[u+1]) belong to
 * sequence u (device int32 [n_seq+1]; max_len of the handle bounds total_steps) -- and the step minimises
 * sum_t nll_scale[t] * NLL_t + sum_t kl_scale[t] * KL_t (device float [total_steps] each): with nll_scale = 1 / (d_u * n_seq) and
 * kl_scale = beta / (T_u * n_seq) that is the mean over the pack of the reference's per-user loss -- gradient accumulation
 * over the pack, then one Adam step.  The products become [total_steps, .] GEMMs, the recurrences run one workgroup per
 * sequence side by side.  With n_seq = 1 it computes what rtx_svae_train_step computes.  step->inv_batch / beta are ignored. */
int rtx_svae_train_pack(rtx_svae* s, const int32_t* items, int32_t total_steps, const int32_t* seq_ptr, int32_t n_seq, const float* nll_scale,
                        const float* kl_scale, const int64_t* target_indptr, const int32_t* target_indices, const rtx_step* step, float* loss_out,
                        float* loss_accum, void* stream);

/* measurement knobs of one engine (the defaults are the shipped configuration): key "fuse_adam" (0/1, bf16 step:
 * Adam inside the weight-gradient kernels), "two_stream" (0/1: the two big ones on a second stream beside the
 * data-gradient chain), "side_low_prio" (0/1: that stream at the lowest priority; before the first step), "lse_fuse" (0/1:
 * log-sum-exp partials from the logits GEMM epilogue), "nt_regstage" (0/1: the big NT contractions on the register-staged GEMM
 * instead of the LDS-DMA one), "dw_cfg" (0..7: tile configuration of the weight-gradient kernel, RtxDwCfg in csrc/rtx_gemm.h; 0 = 64 x 128, the default), "splitk" (split factor of the K = n_items GEMMs, 0 = automatic), "in_on_main" (0/1: the encoder matrix's weight kernel on
 * the caller's stream behind the chain), "sparse_in" (0/1, default 0 since round 4, bf16: the first encoder layer as a sparse VALU
 * product over the batch's stored entries -- spmm_in.hip -- instead of the dense MFMA split-K GEMM; needs a CSR batch,
 * n_items + cond_dim <= 20 480 and at most ~4000 expected 64-entry chunks per batch), "small_fwd" / "small_bwd" (0/1, bf16: hidden
 * layers and the VAE head of the forward pass / of the data-gradient chain as one register-resident launch each --
 * small_layers.hip -- for padded widths <= 1024), "logits16" (0/1, bf16 training step: the logits leave their product as IEEE half
 * in the buffer of d loss / d logits and the loss kernel converts them in place), "gather_scatter" (0/1: the dense batch image stays
 * all-zero between batches and only the stored entries are cleared / written -- resident CSR batches), "dp_shard_min_elems" (>= 1, before
 * rtx_engine_dp_attach: smallest weight matrix the sharded optimizer shards; default 2^20 elements), "hop_values" (0/1: the step's
 * two cross-stream dependencies as hipStreamWriteValue32 / hipStreamWaitValue32 pairs on signal memory instead of events; falls
 * back to events where the device cannot wait on a value), "hop_wrap" (>= 2: the sequence number at which both streams drain and
 * the signal words restart from zero; default 2^31 - 16, tests lower it).
 * Replaces round 1's RTX_* environment switches. */
int rtx_engine_set_option(rtx_engine* e, const char* key, int32_t value);
/* the current value of a knob; also "last_sparse_in": 1 when the last forward pass ran the first layer as the sparse product;
 * "side_concurrent": 1 when the step's second stream was seen to run beside the caller's; data parallel, the LAST step's exchange
 * per rank: "dp_bytes_all_reduce", "dp_bytes_reduce_scatter", "dp_bytes_all_gather" (buffer bytes handed to the collectives,
 * saturating at INT32_MAX) and "dp_collectives" (their number) */
int rtx_engine_get_option(const rtx_engine* e, const char* key, int32_t* value);

/* ---- instrumentation: per-kernel HIP-event timing on the engine's stream ------------------------- */
/* enable: 0 = off, 1 = every launch of the site, N > 1 = every N-th launch (two event records cost a few microseconds of the
 * stream they are recorded on: a sampled site perturbs a timed loop N times less) */
int rtx_engine_set_timing(rtx_engine* e, const char* site /* NULL = every launch site */, int32_t enable);
/* synchronises, returns up to `cap` entries (name, total ms, launches) and clears the counters */
int rtx_engine_get_timings(rtx_engine* e, int32_t cap, char (*names)[48], float* total_ms, int32_t* launches,
                           int32_t* n_out);
/* algorithmic HBM bytes / MFMA flops of one train step at batch B (DESIGN.md derivation) */
int rtx_engine_step_cost(const rtx_engine* e, int32_t batch, double* hbm_bytes, double* flops);

#ifdef __cplusplus
}
#endif
#endif
