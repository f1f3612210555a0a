// engine.hip -- host orchestration of the Mult-VAE / Mult-DAE step on MI355X and the C ABI
// (include/rectorch_hip.h).  One stream-ordered sequence of hand-written kernels per call; no host
// synchronisation on the CSR fast path.
//
// Data layout in HBM (T = bf16 or f32 by cfg.numerics; P(d) = roundup(d+1,128), Bp = roundup(B,128)):
//   layer l (in_l -> out_l), l = 0 .. NL-1 (encoder layers then decoder layers)
//     Wsh[l]   T [P(out)][P(in)]    compute copy of W_l, row-major like the master: the forward B operand (K = in
//                                   contiguous) AND, read K-major, the backward-data B operand
//     A[l]     T [Bp][P(in)]        input activation of layer l; column `in` = ones for b < B: the forward A operand and,
//                                   read K-major, the weight-gradient B operand (the ones column yields the bias gradient)
//     O32[l]   f32 [Bp][P(out)]     post-activation output (tanh layers)         (backward derivative)
//     D[l]     T [Bp][P(out)]       d loss / d pre-activation: backward-data A operand and, read K-major, the
//                                   weight-gradient A operand
//   Y  f32 [Bp][P(I)] logits;  Cacc f32 split-K slabs / small GEMM outputs.
//   No transposed copy of anything exists: the kernels that contract over the batch or over the output features read
//   the row-major matrices K-major (ds_read_b64_tr_b16 for bf16, plain 4-byte reads for f32).
//   All pads are zero (memset at creation; producers rewrite the batch padding every call), so no GEMM
//   needs a bounds check in its main loop.
#include "engine_internal.h"


// ------------------------------------------------------------------------------------------------
int dev_alloc(rtx_engine* e, void** p, size_t bytes, bool zero)
{
    if (bytes == 0) bytes = 16;
    hipError_t rc = hipMalloc(p, bytes);
    if (rc != hipSuccess) {
        rtx_set_error("hipMalloc(%zu bytes) failed: %s", bytes, hipGetErrorString(rc));
        return RTX_ENOMEM;
    }
    if (e) e->allocs.push_back(*p);
    if (zero) {
        // hipMemset runs on the NULL stream, and neither the engine's side stream nor a caller's non-blocking stream (every
        // torch.cuda.Stream) waits for that one: without the synchronise below the zeroing can land AFTER the first kernels that
        // write the buffer.  Found by the stream-ordered multi-rank test and the batch prefetch (round 5): a data-parallel step on
        // a non-blocking stream lost its first gradient images to the late memset of the exchange buffer, a prefetched batch image
        // was zeroed after the gather had filled it.  Allocation is set-up time; the wait costs nothing per step.
        RTX_HIP(hipMemset(*p, 0, bytes));
        RTX_HIP(hipStreamSynchronize(nullptr));
    }
    return RTX_OK;
}

static int build_layers(const rtx_cfg& c, std::vector<Layer>& L)
{
    RTX_CHECK(c.n_enc >= 1 && c.n_dec >= 1 && c.n_enc <= RTX_MAX_LAYERS && c.n_dec <= RTX_MAX_LAYERS, RTX_EINVAL,
              "bad layer counts %d/%d", c.n_enc, c.n_dec);
    RTX_CHECK(c.enc_dims[c.n_enc] == c.dec_dims[0], RTX_EINVAL, "latent size mismatch: enc %d vs dec %d",
              c.enc_dims[c.n_enc], c.dec_dims[0]);
    RTX_CHECK(c.enc_dims[0] == c.dec_dims[c.n_dec], RTX_EINVAL, "n_items mismatch: enc %d vs dec %d", c.enc_dims[0],
              c.dec_dims[c.n_dec]);
    RTX_CHECK(c.cond_dim >= 0, RTX_EINVAL, "negative cond_dim %d", c.cond_dim);
    L.clear();
    for (int i = 0; i < c.n_enc; ++i) {
        Layer l;
        l.in = c.enc_dims[i] + (i == 0 ? c.cond_dim : 0);   // CMultiVAE_net: temp_dims[0] += cond_dim (nets.py:459-460)
        l.out = c.enc_dims[i + 1];
        l.tanh_act = true;
        if (i == c.n_enc - 1 && c.variant == RTX_VAE) {  // mu | logvar, linear (reference nets.py:262-265, 398-404)
            l.out = 2 * c.enc_dims[i + 1];
            l.tanh_act = false;
        }
        L.push_back(l);
    }
    for (int i = 0; i < c.n_dec; ++i) {
        Layer l;
        l.in = c.dec_dims[i];
        l.out = c.dec_dims[i + 1];
        l.tanh_act = (i != c.n_dec - 1);  // reference nets.py:413-417 / 227-233
        L.push_back(l);
    }
    for (auto& l : L) {
        RTX_CHECK(l.in > 0 && l.out > 0, RTX_EINVAL, "non-positive layer size");
        l.inp = rtx_pad(l.in);
        l.outp = rtx_pad(l.out);
    }
    return RTX_OK;
}

// ---- GEMM helpers -------------------------------------------------------------------------------------
// A contraction whose output goes to the fp32 scratch Cacc (possibly as split-K slabs):
//   form NT: C[Mp][Np] = A[Mp][Kp] x B[Np][Kp]^T      (forward)
//   form NN: C[Mp][Np] = A[Mp][Kp] x B[Kp][Np]        (backward-data: B = the weight copy itself)
struct GemmPlan {
    int cfg;        // LDS-DMA kernel: RtxDmaCfg; register-staged kernels (f32, optionally bf16 NT): 128x128 tiles
    int regstage;
    int bm, bn;
    int m_tiles, n_tiles, k_slices, splits;
};

static GemmPlan plan_gemm(const rtx_engine* e, int Mp, int Np, int Kp, int form = RTX_FORM_NT)
{
    GemmPlan pl = {};
    pl.k_slices = (int)((size_t)Kp * e->esz / 128);     // 64 bf16 or 32 f32 per slice
    pl.regstage = !e->bf16 || (form == RTX_FORM_NT && e->opt_nt_regstage && (long)Np * Kp >= (1L << 21));
    if (!pl.regstage) {
        // the 512-row tile covers every batch row of a B <= 512 step: each weight byte is read by one workgroup.  Small
        // problems (hidden layers) are latency-bound: 128x128 tiles give 4x the workgroups.
        const bool big = (Mp % 512 == 0) && ((long)Np * Kp >= (1L << 21));
        pl.cfg = big ? RTX_DMA_512x128 : RTX_DMA_128x128;
        // data-gradient chain beside the weight-gradient kernels (two streams): a 64-KB-LDS configuration, so that its
        // workgroups fit on a CU next to one of theirs (the 512-row tile takes the whole LDS of a CU)
        if (form == RTX_FORM_NN && e->opt_fuse_adam && e->opt_two_stream) pl.cfg = RTX_DMA_128x128_S2;
        // a batch of thousands of rows (configs[3] on one GPU: B = 4096): the product is no longer a skinny one hiding beside a
        // streaming kernel but 90 GFLOP of its own -- the 512-row tile needs a third of the operand bytes per flop of the 128 x 128 one
        if (big && e->opt_big_batch_tiles && Mp >= 1024) pl.cfg = RTX_DMA_512x128;
        rtx_gemm_dma_tile_dims(pl.cfg, &pl.bm, &pl.bn);
    } else {
        pl.cfg = RTX_TILE_128x128;
        pl.bm = pl.bn = 128;
    }
    pl.m_tiles = Mp / pl.bm;
    pl.n_tiles = Np / pl.bn;
    const int tiles = pl.m_tiles * pl.n_tiles;
    // split K so that one wave of workgroups fills the chip: 1 per CU for the LDS-DMA kernels (their stages fill the
    // LDS), 2 per CU for the register-staged f32 kernel; at least two K slices per workgroup
    const int resident = (pl.regstage || pl.cfg == RTX_DMA_128x128_S2) ? 512 : 256;
    int s = 1;
    if (tiles * 2 <= resident) {
        s = resident / tiles;
        if (pl.regstage && s >= 8) s &= ~7;   // register-staged kernel: every XCD owns whole splits
        // data-gradient products: the slabs are summed by a post kernel beside the streaming weight kernel, where every slab
        // costs: 16 slabs instead of 25 at the ml-20m shape is 5 us per step (286 vs 291; 12: 288, 8: 292)
        if (form == RTX_FORM_NN && s > 16 && e->bf16) s = 16;   // (float32: one stream, nothing beside the post kernel -- fill the chip)
        if (pl.k_slices >= 64 && e->cfg.splitk > 0) s = e->cfg.splitk;
        if (pl.k_slices >= 64 && form == RTX_FORM_NT && e->opt_splitk_fwd > 0) s = e->opt_splitk_fwd;
        if (pl.k_slices >= 64 && form == RTX_FORM_NN && e->opt_splitk_bwd > 0) s = e->opt_splitk_bwd;
        const int max_s = pl.k_slices / 2 > 0 ? pl.k_slices / 2 : 1;
        if (s > max_s) s = max_s;
        if (s < 1) s = 1;
    }
    // every split must own at least one slice: per = ceil(k / s) slices each -> ceil(k / per) non-empty splits
    const int per = (pl.k_slices + s - 1) / s;
    pl.splits = (pl.k_slices + per - 1) / per;
    return pl;
}

// A deferred join folded into the first-layer product (RtxGemm::wait_word) makes EVERY workgroup of that grid spin until the side
// stream has stored its number.  The side stream's remaining kernels (weight gradient + Adam, loss sum, the next batch's gather, the
// store itself) must therefore be able to make progress beside a grid that is entirely resident and spinning.  Occupancy argument: the
// register-staged 128 x 128 product takes 73 728 B of LDS per workgroup, i.e. at most TWO workgroups per CU whatever else limits it; a
// grid of G workgroups leaves at least 2 * n_cus - G of those slots empty, and a CU with an empty slot has >= 86 KB of LDS, >= 28 wave
// slots and >= 328 registers per lane and SIMD free -- room for a workgroup of any kernel the side stream runs (the largest, the 64 x 128
// weight-gradient tile: 72 KB, 8 waves, <= 128 registers).  With fewer than 16 empty slots, or any other product kernel, the join is
// the one-wave k_hop_wait in front of the step instead (resolve_join): a spinning wave that holds nothing.
static bool fold_has_room(const rtx_engine* e, int Mp, int Np, int Kp)
{
    const GemmPlan pl = plan_gemm(e, Mp, Np, Kp, RTX_FORM_NT);
    if (!pl.regstage || pl.cfg != RTX_TILE_128x128) return false;
    const long groups = pl.splits > 1 ? pl.splits : (pl.m_tiles <= pl.n_tiles ? pl.n_tiles : pl.m_tiles);
    const long gsize = pl.splits > 1 ? (long)pl.m_tiles * pl.n_tiles : (pl.m_tiles <= pl.n_tiles ? pl.m_tiles : pl.n_tiles);
    const long grid = 8 * ((groups + 7) / 8) * gsize;   // (rtx_gemm_launch's grid: idle workgroups of the XCD padding exit at once, counted anyway)
    return grid + 16 <= 2L * e->n_cus;
}

size_t plan_cacc_elems(rtx_engine* e, int Np, int Kp)
{
    size_t mx = 0;
    const int keep = e->opt_nt_regstage, keep2 = e->opt_two_stream;
    for (int rs = 0; rs < 4; ++rs) {     // whichever kernels the knobs select later
        e->opt_nt_regstage = rs & 1;
        e->opt_two_stream = rs >> 1;
        for (int form : {RTX_FORM_NT, RTX_FORM_NN})
            for (int Mp = 128; Mp <= e->Bp_alloc; Mp += 128) {
                const GemmPlan pl = plan_gemm(e, Mp, Np, Kp, form);
                mx = std::max(mx, (size_t)pl.splits * Mp * Np);
            }
    }
    e->opt_nt_regstage = keep;
    e->opt_two_stream = keep2;
    return mx;
}

static int gemm_to_cacc(rtx_engine* e, int form, const void* A, long lda, const void* B, long ldb, int Mp, int Np, int Kp, int* splits_out,
                        hipStream_t st, uint32_t* hop_word = nullptr, uint32_t hop_seq = 0, const uint32_t* wait_word = nullptr, uint32_t wait_seq = 0)
{
    const GemmPlan pl = plan_gemm(e, Mp, Np, Kp, form);
    RtxGemm g = {};
    g.form = form;
    g.A = A; g.B = B; g.lda = lda; g.ldb = ldb;
    g.k_slices = pl.k_slices; g.tile_shape = pl.cfg; g.m_tiles = pl.m_tiles; g.n_tiles = pl.n_tiles; g.splits = pl.splits;
    g.C = e->Cacc; g.ldc = Np; g.slab_stride = (long)Mp * Np;
    RTX_CHECK((size_t)g.splits * Mp * Np <= e->cacc_elems, RTX_ESTATE, "internal: Cacc too small (%d x %d x %d)", g.splits, Mp, Np);
    *splits_out = g.splits;
    g.xcd_block = 1;   // (the launcher keeps the strip order for split-K and for grids under 8 x 4 tiles)
    RTX_CHECK(!hop_word || !pl.regstage, RTX_ESTATE, "internal: a folded stream hop needs the LDS-DMA product");
    g.hop_word = hop_word; g.hop_seq = hop_seq;
    RTX_CHECK(!wait_word || e->bf16, RTX_ESTATE, "internal: a folded join needs a bf16 product");
    g.wait_word = wait_word; g.wait_seq = wait_seq;
    if (!pl.regstage) return rtx_gemm_dma_launch(g, RTX_EPI_STORE, st);
    if (form == RTX_FORM_NT) return rtx_gemm_launch(g, e->bf16 ? RTX_DT_BF16 : RTX_DT_F32, RTX_EPI_STORE, st);
    return rtx_gemm_f32_km_launch(g, RTX_EPI_STORE, st);
}

// ---- batch resolution --------------------------------------------------------------------------------
static int ensure_tmp(rtx_engine* e, TempCsr& t, int64_t nnz)
{
    if (!t.indptr) {
        RTX_TRY(dev_alloc(e, (void**)&t.indptr, sizeof(int64_t) * (e->cfg.max_batch + 1)));
        RTX_TRY(dev_alloc(e, (void**)&t.counts, sizeof(int32_t) * (e->cfg.max_batch + 1)));
    }
    if (nnz > t.cap) {
        int64_t cap = nnz + nnz / 2 + 1024;
        // old buffers stay in e->allocs and are freed with the engine (growth is rare)
        RTX_TRY(dev_alloc(e, (void**)&t.indices, sizeof(int32_t) * cap, false));
        RTX_TRY(dev_alloc(e, (void**)&t.values, sizeof(float) * cap, false));
        t.cap = cap;
    }
    return RTX_OK;
}

static int dense_to_view(rtx_engine* e, TempCsr& t, const float* x, int B, int width, RtxCsrView* v, hipStream_t st)
{
    RTX_TRY(ensure_tmp(e, t, 0));
    RTX_TRY(rtx_launch_dense_count(x, B, width, t.counts, st));
    RTX_TRY(rtx_launch_scan_counts(t.counts, B, t.indptr, st));
    int64_t nnz = 0;
    RTX_HIP(hipMemcpyAsync(&nnz, t.indptr + B, sizeof(int64_t), hipMemcpyDeviceToHost, st));
    RTX_HIP(hipStreamSynchronize(st));  // dense drop-in path only; the CSR path never synchronises
    RTX_TRY(ensure_tmp(e, t, nnz));
    RTX_TRY(rtx_launch_dense_fill(x, B, width, t.indptr, t.indices, t.values, st));
    v->indptr = t.indptr; v->indices = t.indices; v->values = t.values; v->row_ids = nullptr;
    v->max_row_len = 0; v->avg_row_len = 0;   // a densified batch: row lengths unknown -> the dense first layer (sparse_in_ok)
    return RTX_OK;
}

static int resolve_batch(rtx_engine* e, const rtx_batch* b, RtxCsrView* in, RtxCsrView* tg, hipStream_t st, int need_target = 1)
{
    RTX_CHECK(b, RTX_EINVAL, "batch is NULL");
    RTX_CHECK(b->batch >= 1 && b->batch <= e->cfg.max_batch, RTX_EINVAL, "batch %d outside [1, max_batch=%d]", b->batch,
              e->cfg.max_batch);
    if (b->csr) {
        RTX_CHECK(b->csr->n_cols == e->Iin, RTX_EINVAL, "CSR has %d columns, network expects %d (items + conditions)", b->csr->n_cols,
                  e->Iin);
        RTX_CHECK(b->row_ids || b->batch <= b->csr->n_rows, RTX_EINVAL, "batch larger than the matrix");
        in->indptr = b->csr->indptr; in->indices = b->csr->indices; in->values = b->csr->values; in->row_ids = b->row_ids;
        in->max_row_len = b->csr->max_row_len;
        in->avg_row_len = (int32_t)std::min<int64_t>((b->csr->nnz + std::max<int64_t>(b->csr->n_rows, 1) - 1) / std::max<int64_t>(b->csr->n_rows, 1), INT32_MAX);
    } else {
        RTX_CHECK(b->x_dense, RTX_EINVAL, "batch has neither csr nor x_dense");
        RTX_TRY(dense_to_view(e, e->tmp_in, b->x_dense, b->batch, e->Iin, in, st));
    }
    if (b->target_csr) {
        RTX_CHECK(b->target_csr->n_cols == e->I, RTX_EINVAL, "target CSR has %d columns, expected %d", b->target_csr->n_cols, e->I);
        tg->indptr = b->target_csr->indptr; tg->indices = b->target_csr->indices; tg->values = b->target_csr->values;
        tg->row_ids = b->row_ids;
        tg->max_row_len = b->target_csr->max_row_len;
    } else if (b->target_dense) {
        RTX_TRY(dense_to_view(e, e->tmp_tg, b->target_dense, b->batch, e->I, tg, st));
    } else {
        // a conditioned input has cond_dim extra columns: it cannot be its own target.  Scoring calls (need_target = 0)
        // never read the target beyond its row sums, which k_gather restricts to the item columns.
        RTX_CHECK(e->Iin == e->I || !need_target, RTX_EINVAL, "a conditioned network (cond_dim = %d) needs an explicit target", e->Iin - e->I);
        *tg = *in;
    }
    return RTX_OK;
}

// ---- forward ---------------------------------------------------------------------------------------
// Runs layers [l0, l1).  If l0 == 0 the gather kernel builds A[0] from `in`.  The last network layer
// writes logits to `logits` (ld = ldlog); with want_lse it also leaves the log-sum-exp partials (training).
// The first layer as a sparse product (spmm_in.hip): bf16 numerics, a batch that names rows of a resident CSR matrix (its
// longest row bounds the chunk stream), a first layer followed by an ordinary activation, weight rows that fit the LDS.
// the training step's logits as half precision in the delta buffer (opt_logits16): needs the epilogue's log-sum-exp partials and
// the register-staged product that writes them
static bool logits16_on(const rtx_engine* e) { return e->bf16 && e->opt_logits16 && e->opt_lse_fuse && e->opt_nt_regstage; }

static bool sparse_in_ok(const rtx_engine* e, const RtxCsrView* in, int Bp, int64_t* chunks)
{
    if (!e->bf16 || !e->opt_sparse_in || e->NL < 2 || (e->vae && e->cfg.n_enc == 1)) return false;
    if (in->max_row_len <= 0 || e->Iin > 65536 || rtx_spmm_in_lds_bytes(e->Iin) > 160 * 1024) return false;
    *chunks = (int64_t)Bp * std::max(1, (in->max_row_len + 63) / 64) + 64;   // + the read-ahead of the last wave
    // every workgroup of k_spmm_in walks the whole chunk stream (~6 ns per chunk), the dense product re-reads the weights and
    // the dense batch image (~25 us + 8 us per 1000 rows): beyond ~4000 expected chunks (ml-20m at B = 500: ~1500; Netflix-
    // shaped rows at B = 4096: ~18 000) the dense product wins
    const int64_t expected = (int64_t)Bp * ((in->avg_row_len + 63) / 64 + 1);
    return *chunks * 256 <= ((int64_t)512 << 20) && expected <= 4096;
}
static int ensure_in_chunks(rtx_engine* e, int64_t chunks, hipStream_t st)
{
    if (chunks <= e->in_cap_chunks) return RTX_OK;
    if (e->in_ent) {   // a matrix with longer rows than the last one: rare, so simply wait and regrow
        RTX_HIP(hipStreamSynchronize(st));
        for (void* p : {(void*)e->in_ent, (void*)e->in_desc}) {
            e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), p));
            (void)hipFree(p);
        }
        e->in_ent = nullptr; e->in_desc = nullptr;
    }
    RTX_TRY(dev_alloc(e, (void**)&e->in_ent, (size_t)chunks * 256));
    RTX_TRY(dev_alloc(e, (void**)&e->in_desc, (size_t)(chunks + 128) * sizeof(int32_t)));
    if (!e->in_wsplit) RTX_TRY(dev_alloc(e, (void**)&e->in_wsplit, (RTX_SPMM_WAVES + 1) * sizeof(int32_t)));
    e->in_cap_chunks = chunks;
    return RTX_OK;
}

// Third form (round 5, option "hop_kernels"): the dependency as two ONE-WAVE KERNELS -- k_hop_set on the producing stream stores a
// sequence number (agent-scope release) behind the kernels it follows, k_hop_wait on the consuming stream spins on it (acquire,
// s_sleep between polls, bounded) in front of the kernels that need the data.  A kernel boundary on each side: the producers'
// end-of-kernel release has completed before k_hop_set runs (in-order queue), the consumers' start-of-kernel acquire comes after
// k_hop_wait has seen the number.  No stream memory operation, no event: those are packets that make the command processor release
// to SYSTEM scope (the signal word is host-visible memory) and cost the stream 6-9 us each (profiles/r4_step_timeline.txt: the
// gaps behind k_dlogits and between two steps).
__global__ void k_hop_set(uint32_t* word, uint32_t v)
{
    if (threadIdx.x == 0) __hip_atomic_store(word, v, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
}
__global__ void k_hop_wait(const uint32_t* word, uint32_t v, uint32_t* stuck)
{
    if (threadIdx.x != 0) return;
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // 100 MHz
    while ((int32_t)(__hip_atomic_load(word, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) - v) < 0) {
        __builtin_amdgcn_s_sleep(8);
        if (__builtin_amdgcn_s_memrealtime() - t0 > 2000000000ull) {   // 20 s: the producer is gone; do not hang the device
            if (stuck) *stuck = v;
            __builtin_trap();
        }
    }
}

static int ensure_hopk(rtx_engine* e)
{
    if (!e->hopk_mem) {
        RTX_HIP(hipMalloc((void**)&e->hopk_mem, 64));
        RTX_HIP(hipMemset(e->hopk_mem, 0, 64));
        RTX_HIP(hipStreamSynchronize(nullptr));
    }
    return RTX_OK;
}

// the join a step flagged RTX_STEP_DEFER_JOIN left open: `st` continues only after everything that step put on the side stream
static int resolve_join(rtx_engine* e, hipStream_t st)
{
    if (!e->join_pending) return RTX_OK;
    e->join_pending = false;
    e->join_fold = false;
    hipLaunchKernelGGL(k_hop_wait, dim3(1), dim3(64), 0, st, e->hopk_mem + 3, e->join_seq, e->hopk_mem + 11);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}

// The batch image A[0] (+ target row sums, + the scatter lists) of one batch, on stream `st`, into the CURRENT image set.
static int gather_batch(rtx_engine* e, const RtxCsrView* in, const RtxCsrView* tg, int B, int training, const rtx_step* step, hipStream_t st)
{
    const int Bp = rtx_pad_batch(B);
    Layer& l = e->L[0];
    RtxGatherArgs a = {};
    a.in = *in; a.target = *tg;
    a.B = B; a.Bp = Bp; a.I = e->I; a.Iin = e->Iin; a.ldx = l.inp;
    a.X = l.A; a.tsum = e->tsum;
    a.training = training; a.dropout_p = e->cfg.dropout_p;
    a.mask = step->dropout_mask; a.seed = step->seed; a.offset = step->offset;
    // The image is all zeros but for ~75 entries per user: with a resident matrix (its longest row is known) only those are
    // touched -- cleared, rewritten, listed (k_gather_scatter).  First use, a longer matrix, or another writer of A[0] in
    // between (the sparse first layer's k_in_chunks, a densified batch through k_gather): one full reset of image and lists.
    if (e->opt_gather_scatter && in->max_row_len > 0 && ((int64_t)in->max_row_len + 66) * e->Bp_alloc * 4 <= ((int64_t)256 << 20)) {
        const int need = (in->max_row_len + 2 + 63) / 64 * 64;
        if (need > e->img_cap) {
            if (e->img_written) {
                RTX_HIP(hipStreamSynchronize(st));
                for (void* q : {(void*)e->img_written, (void*)e->img_nwritten}) {
                    e->allocs.erase(std::find(e->allocs.begin(), e->allocs.end(), q));
                    (void)hipFree(q);
                }
                e->img_written = nullptr; e->img_nwritten = nullptr;
            }
            RTX_TRY(dev_alloc(e, (void**)&e->img_written, (size_t)e->Bp_alloc * need * sizeof(int32_t), false));
            RTX_TRY(dev_alloc(e, (void**)&e->img_nwritten, (size_t)e->Bp_alloc * sizeof(int32_t)));
            e->img_cap = need;
            e->img_exact = false;
        }
        if (!e->img_exact) {
            RTX_HIP(hipMemsetAsync(l.A, 0, (size_t)e->Bp_alloc * l.inp * e->esz, st));
            RTX_HIP(hipMemsetAsync(e->img_nwritten, 0, (size_t)e->Bp_alloc * sizeof(int32_t), st));
            e->img_exact = true;
        }
        a.written = e->img_written; a.n_written = e->img_nwritten; a.written_cap = e->img_cap;
    } else {
        e->img_exact = false;
    }
    TIMED("gather");
    RTX_TRY(rtx_launch_gather(a, e->bf16, st));
    return RTX_OK;
}

static void swap_img_sets(rtx_engine* e)
{
    std::swap(e->L[0].A, e->A0_alt);
    std::swap(e->tsum, e->tsum_alt);
    std::swap(e->img_written, e->img_written_alt);
    std::swap(e->img_nwritten, e->img_nwritten_alt);
    std::swap(e->img_cap, e->img_cap_alt);
    std::swap(e->img_exact, e->img_exact_alt);
}

static int run_forward(rtx_engine* e, const RtxCsrView* in, const RtxCsrView* tg, int B, int training, const rtx_step* step,
                       int want_lse, int l0, int l1, float* logits, long ldlog, float* mu_out, float* lv_out, hipStream_t st)
{
    const int Bp = rtx_pad_batch(B);
    static const rtx_step zero_step = {};
    if (!step) step = &zero_step;
    int64_t in_chunks = 0;
    const bool sparse_in = l0 == 0 && l1 > 1 && sparse_in_ok(e, in, Bp, &in_chunks);
    if (l0 == 0) e->last_sparse_in = sparse_in;
    const bool gathered = e->gather_done;
    e->gather_done = false;
    if (e->join_fold && !(l0 == 0 && l1 > 1 && !sparse_in && gathered && e->bf16)) {
        // the deferred join cannot ride on the first-layer product after all (it is not this call's first kernel): a kernel of its own
        e->join_fold = false;
        hipLaunchKernelGGL(k_hop_wait, dim3(1), dim3(64), 0, st, e->hopk_mem + 3, e->join_seq, e->hopk_mem + 11);
        RTX_HIP(hipGetLastError());
    }
    if (l0 == 0 && !sparse_in && !gathered) RTX_TRY(gather_batch(e, in, tg, B, training, step, st));
    for (int li = l0; li < l1; ++li) {
        Layer& l = e->L[li];
        if (li == e->NL - 1) {
            // logits = A x Wsh^T + b, one launch; K = hidden (short), output [Bp][n_items] f32
            RtxGemm g = {};
            g.form = RTX_FORM_NT;
            g.A = l.A; g.B = l.Wsh; g.lda = l.inp; g.ldb = l.inp;
            g.k_slices = (int)((size_t)l.inp * e->esz / 128);
            g.splits = 1; g.C = logits; g.ldc = ldlog; g.bias = e->params[2 * li + 1];
            g.M_real = B; g.N_real = l.out;
            TIMED("gemm_logits");
            if (e->bf16 && !e->opt_nt_regstage) {
                g.tile_shape = (Bp % 512 == 0) ? RTX_DMA_512x128 : RTX_DMA_128x128;
                int bm, bn;
                rtx_gemm_dma_tile_dims(g.tile_shape, &bm, &bn);
                g.m_tiles = Bp / bm; g.n_tiles = l.outp / bn;
                if (want_lse && e->opt_lse_fuse) { g.lse_part = e->lse_part; g.lse_ld = e->lse_strips; }   // (see lse_fused)
                g.xcd_block = 1;
                RTX_TRY(rtx_gemm_dma_launch(g, RTX_EPI_BIAS_ROWS, st));
            } else {
                g.tile_shape = RTX_TILE_128x128;
                g.m_tiles = Bp / 128; g.n_tiles = l.outp / 128;
                if (want_lse && e->opt_lse_fuse) { g.lse_part = e->lse_part; g.lse_ld = e->lse_strips; }
                if (want_lse && logits16_on(e)) { g.C16 = l.D; g.ldc16 = l.outp; }
                RTX_TRY(rtx_gemm_launch(g, e->bf16 ? RTX_DT_BF16 : RTX_DT_F32, RTX_EPI_BIAS_ROWS, st));
            }
            break;
        }
        if (li == 0 && sparse_in) {
            // the batch's stored entries as a chunk stream; in a training step the same launch also leaves what the gather
            // kernel would (the dense image the weight-gradient kernel reads, the target row sums)
            RTX_TRY(ensure_in_chunks(e, in_chunks, st));
            RtxInChunksArgs c = {};
            c.in = *in; c.B = B; c.I = e->I; c.Iin = e->Iin;
            c.training = training; c.dropout_p = e->cfg.dropout_p;
            c.mask = step->dropout_mask; c.seed = step->seed; c.offset = step->offset;
            c.ent = e->in_ent; c.desc = e->in_desc; c.wsplit = e->in_wsplit; c.cap_chunks = e->in_cap_chunks;
            if (training) { c.target = *tg; c.tsum = e->tsum; c.X = (bf16_t*)l.A; c.ldx = l.inp; c.Bp = Bp; e->img_exact = false; }
            {
                TIMED("in_chunks");
                RTX_TRY(rtx_launch_in_chunks(c, st));
            }
            RtxSpmmInArgs a = {};
            a.ent = e->in_ent; a.desc = e->in_desc; a.wsplit = e->in_wsplit;
            a.B = B; a.Bp = Bp;
            a.W = (const bf16_t*)l.Wsh; a.ldw = l.inp; a.Kin = e->Iin;
            a.bias = e->params[2 * li + 1]; a.N_real = l.out; a.Np = l.outp; a.tanh_act = l.tanh_act;
            a.O32 = l.O32; a.R = (bf16_t*)e->L[li + 1].A; a.ones_col = 1;
            TIMED("spmm_in");
            RTX_TRY(rtx_launch_spmm_in(a, st));
            continue;
        }
        if (li > 0 && e->bf16 && e->opt_small_fwd && rtx_small_fwd_ok(l.inp)) {
            // a hidden layer (or the VAE head): product + bias + activation + the next operand in one launch (small_layers.hip)
            Layer& nx1 = e->L[li + 1];
            RtxSmallFwdArgs a = {};
            a.A = (const bf16_t*)l.A; a.W = (const bf16_t*)l.Wsh; a.lda = l.inp; a.ldw = l.inp; a.w_rows = l.outp;
            a.B = B; a.Bp = Bp; a.bias = e->params[2 * li + 1]; a.R = (bf16_t*)nx1.A;
            if (e->vae && li == e->cfg.n_enc - 1) {
                a.Z = e->Z; a.Np = e->Zp; a.N_real = e->Z; a.training = training;
                a.mu32 = e->mu32; a.lv32 = e->lv32; a.eps32 = e->eps32; a.mu_out = mu_out; a.lv_out = lv_out;
                a.eps_in = step->eps_noise; a.seed = step->seed; a.offset = step->offset;
            } else {
                a.N_real = l.out; a.Np = l.outp; a.tanh_act = l.tanh_act; a.O32 = l.O32;
            }
            TIMED(a.Z ? "fwd_head" : "fwd_hidden");
            RTX_TRY(rtx_launch_small_fwd(a, st));
            continue;
        }
        int splits = 1;
        {
            TIMED(li == 0 ? "gemm_fwd_in" : "gemm_fwd_hidden");
            const bool fold = e->join_fold;      // (set only for a step whose FIRST kernel is this product)
            e->join_fold = false;
            RTX_TRY(gemm_to_cacc(e, RTX_FORM_NT, l.A, l.inp, l.Wsh, l.inp, Bp, l.outp, l.inp, &splits, st, nullptr, 0, fold ? e->hopk_mem + 3 : nullptr, e->join_seq));
        }
        Layer& nx = e->L[li + 1];
        if (e->vae && li == e->cfg.n_enc - 1) {
            RtxVaeFwdArgs a = {};
            a.C = e->Cacc; a.splits = splits; a.slab_stride = (long)Bp * l.outp; a.ldc = l.outp;
            a.B = B; a.Bp = Bp; a.Z = e->Z; a.Zp = e->Zp;
            a.bias = e->params[2 * li + 1];
            a.mu32 = e->mu32; a.lv32 = e->lv32; a.eps32 = e->eps32;
            a.mu_out = mu_out; a.lv_out = lv_out;
            a.Zr = nx.A;
            a.training = training; a.eps_in = step->eps_noise; a.seed = step->seed; a.offset = step->offset;
            TIMED("vae_head_fwd");
            RTX_TRY(rtx_launch_vae_fwd(a, e->bf16, st));
        } else {
            RtxPostArgs a = {};
            a.C = e->Cacc; a.splits = splits; a.slab_stride = (long)Bp * l.outp; a.ldc = l.outp;
            a.B = B; a.Bp = Bp; a.N_real = l.out; a.Np = l.outp;
            a.tanh_act = l.tanh_act; a.bias = e->params[2 * li + 1];
            a.O32 = l.O32; a.R = nx.A; a.ones_col = 1;
            TIMED("post_fwd");
            RTX_TRY(rtx_launch_post(a, RTX_POST_FWD, e->bf16, st));
        }
    }
    return RTX_OK;
}

int check_ready(rtx_engine* e, bool train)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    RTX_CHECK(e->bound, RTX_ESTATE, "rtx_engine_bind() has not been called");
    RTX_CHECK(!train || e->can_train, RTX_ESTATE, "engine was bound without gradient / Adam buffers");
    return RTX_OK;
}

int ensure_shadows(rtx_engine* e, hipStream_t st)
{
    RTX_TRY(resolve_join(e, st));   // (a join the last training step left open: before anything else of the engine runs on `st`)
    if (e->shadows_valid) return RTX_OK;
    return rtx_engine_sync_shadows(e, st);
}

// tensors of layers [l0, l1) into a (W and b per layer)
static void fill_adam_tensors(rtx_engine* e, RtxAdamArgs& a, int l0 = 0, int l1 = -1)
{
    if (l1 < 0) l1 = e->NL;
    a.n = 0;
    for (int li = l0; li < l1; ++li) {
        Layer& l = e->L[li];
        RtxAdamTensor& w = a.t[a.n++];
        w.p = e->params[2 * li];
        w.g = e->can_train ? e->grads[2 * li] : nullptr;
        w.m = e->can_train ? e->m[2 * li] : nullptr;
        w.v = e->can_train ? e->v[2 * li] : nullptr;
        w.sh = l.Wsh; w.shT = l.WshT; w.rows = l.out; w.cols = l.in; w.ld_sh = l.inp; w.ld_shT = l.WshT ? l.outp : 0;
        w.sumsq = nullptr;
        RtxAdamTensor& b = a.t[a.n++];
        b.p = e->params[2 * li + 1];
        b.g = e->can_train ? e->grads[2 * li + 1] : nullptr;
        b.m = e->can_train ? e->m[2 * li + 1] : nullptr;
        b.v = e->can_train ? e->v[2 * li + 1] : nullptr;
        b.sh = nullptr; b.shT = nullptr; b.rows = 1; b.cols = l.out; b.ld_sh = 0; b.ld_shT = 0;
        b.sumsq = nullptr;
    }
}

// torch.optim.Adam's scalars for update `step` (computed in double like torch does on the host).  The tensors of `a`
// are tensors [t0, t0 + a.n) of the network unless `ids` names them one by one (DAE: per-tensor norms).
static void fill_adam_scalars(rtx_engine* e, const rtx_step* step, RtxAdamArgs& a, int t0 /* first tensor index */, const int* ids = nullptr)
{
    a.update = 1;
    const double bc1 = 1.0 - pow((double)step->beta1, (double)step->step);
    const double bc2 = 1.0 - pow((double)step->beta2, (double)step->step);
    a.step_size = (float)((double)step->lr / bc1);
    a.bc2_sqrt = (float)sqrt(bc2);
    a.beta1 = step->beta1; a.beta2 = step->beta2; a.eps = step->eps; a.weight_decay = step->weight_decay;
    a.grad_scale = 1.f;
    a.lam = 0.f;
    if (!e->vae && step->lam != 0.f) {
        a.lam = step->lam;
        for (int k = 0; k < a.n; ++k) a.t[k].sumsq = e->sumsq + (ids ? ids[k] : t0 + k);
    }
}

static int launch_sumsq(rtx_engine* e, hipStream_t st)
{
    std::vector<const float*> ps(2 * e->NL);
    std::vector<long> sz(2 * e->NL);
    for (int li = 0; li < e->NL; ++li) {
        ps[2 * li] = e->params[2 * li];
        sz[2 * li] = (long)e->L[li].out * e->L[li].in;
        ps[2 * li + 1] = e->params[2 * li + 1];
        sz[2 * li + 1] = e->L[li].out;
    }
    return rtx_launch_sumsq(ps.data(), sz.data(), 2 * e->NL, e->sumsq, st);
}

__global__ void k_unpad_copy(const float* src, int ld, int B, int n, float* dst)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)B * n) dst[i] = src[(i / n) * ld + (i % n)];
}

template <typename T>
__global__ void k_pad_convert(const float* src, int B, int n, T* dst, int ld, int Bp)
{
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i < (long)Bp * ld) {
        const int b = (int)(i / ld), j = (int)(i % ld);
        dst[i] = Elem<T>::from((b < B && j < n) ? src[(long)b * n + j] : 0.f);
    }
}

// =================================================================================================
extern "C" {

const char* rtx_last_error(void) { return rtx_last_error_str(); }
int32_t rtx_abi_version(void) { return 8; }   // 8: rtx_engine_evaluate_topk;   // 2: rtx_cfg.cond_dim, rtx_ease_*; 3: rtx_engine_set_option, step fuses Adam by default; 4: rtx_comm_*, rtx_engine_apply_adam_rows / shadow_region; 5: rtx_engine_dp_attach / train_step_dp (the engine schedules the data-parallel step); 6: rtx_svae_set_option; 7: rtx_dp_cfg.comm_side / ops_side / shard_min_elems (bucket A's own communicator), rtx_engine_loss_mailbox / rtx_engine_wait_loss

// ---- CSR -------------------------------------------------------------------------------------------
int rtx_csr_upload(const int64_t* indptr_host, const int32_t* indices_host, const float* values_host, int64_t n_rows,
                   int32_t n_cols, rtx_csr** out)
{
    RTX_CHECK(indptr_host && out && n_rows >= 0 && n_cols > 0, RTX_EINVAL, "csr_upload: bad arguments");
    const int64_t nnz = indptr_host[n_rows];
    RTX_CHECK(nnz == 0 || indices_host, RTX_EINVAL, "csr_upload: indices is NULL");
    rtx_csr* m = new rtx_csr();
    m->n_rows = n_rows; m->n_cols = n_cols; m->nnz = nnz;
    int64_t longest = 0;
    for (int64_t r = 0; r < n_rows; ++r) longest = std::max<int64_t>(longest, indptr_host[r + 1] - indptr_host[r]);
    m->max_row_len = (int32_t)std::min<int64_t>(longest, INT32_MAX);
    int rc = dev_alloc(nullptr, (void**)&m->indptr, sizeof(int64_t) * (n_rows + 1), false);
    if (!rc) rc = dev_alloc(nullptr, (void**)&m->indices, sizeof(int32_t) * (nnz > 0 ? nnz : 1), false);
    if (!rc && values_host) rc = dev_alloc(nullptr, (void**)&m->values, sizeof(float) * (nnz > 0 ? nnz : 1), false);
    if (rc) { rtx_csr_destroy(m); return rc; }
    hipError_t h = hipMemcpy(m->indptr, indptr_host, sizeof(int64_t) * (n_rows + 1), hipMemcpyHostToDevice);
    if (h == hipSuccess && nnz) h = hipMemcpy(m->indices, indices_host, sizeof(int32_t) * nnz, hipMemcpyHostToDevice);
    if (h == hipSuccess && nnz && values_host) h = hipMemcpy(m->values, values_host, sizeof(float) * nnz, hipMemcpyHostToDevice);
    if (h != hipSuccess) {
        rtx_set_error("csr_upload: hipMemcpy failed: %s", hipGetErrorString(h));
        rtx_csr_destroy(m);
        return RTX_EHIP;
    }
    *out = m;
    return RTX_OK;
}

int rtx_csr_destroy(rtx_csr* m)
{
    if (!m) return RTX_OK;
    if (m->indptr) (void)hipFree(m->indptr);
    if (m->indices) (void)hipFree(m->indices);
    if (m->values) (void)hipFree(m->values);
    delete m;
    return RTX_OK;
}

int rtx_csr_shape(const rtx_csr* m, int64_t* n_rows, int32_t* n_cols, int64_t* nnz)
{
    RTX_CHECK(m, RTX_EINVAL, "csr is NULL");
    if (n_rows) *n_rows = m->n_rows;
    if (n_cols) *n_cols = m->n_cols;
    if (nnz) *nnz = m->nnz;
    return RTX_OK;
}

int rtx_csr_gather_dense(const rtx_csr* m, const int32_t* row_ids, int32_t batch, float* out, void* stream)
{
    RTX_CHECK(m && out, RTX_EINVAL, "csr_gather_dense: bad arguments");
    RTX_CHECK(batch >= 0 && (row_ids || batch <= m->n_rows), RTX_EINVAL, "csr_gather_dense: bad batch %d", batch);
    RtxCsrView v = {m->indptr, m->indices, m->values, row_ids};
    return rtx_launch_csr_to_dense(v, batch, m->n_cols, out, (hipStream_t)stream);
}

// ---- engine lifecycle ------------------------------------------------------------------------------
int rtx_engine_create(const rtx_cfg* cfg, rtx_engine** out)
{
    RTX_CHECK(cfg && out, RTX_EINVAL, "engine_create: NULL argument");
    RTX_CHECK(cfg->variant == RTX_VAE || cfg->variant == RTX_DAE, RTX_EINVAL, "bad variant %d", cfg->variant);
    RTX_CHECK(cfg->numerics == RTX_FP32 || cfg->numerics == RTX_BF16, RTX_EINVAL, "bad numerics %d", cfg->numerics);
    RTX_CHECK(cfg->max_batch >= 1, RTX_EINVAL, "max_batch must be >= 1");
    RTX_CHECK(cfg->dropout_p >= 0.f && cfg->dropout_p <= 1.f, RTX_EINVAL, "dropout_p outside [0,1]");
    int ndev = 0;
    hipError_t h = hipGetDeviceCount(&ndev);
    RTX_CHECK(h == hipSuccess && ndev > 0, RTX_EHIP, "no HIP device available (%s): librectorch_hip has no CPU path",
              hipGetErrorString(h));
    rtx_engine* e = new rtx_engine();
    e->cfg = *cfg;
    {
        int dev = 0, cus = 0;
        if (hipGetDevice(&dev) == hipSuccess && hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && cus > 0) e->n_cus = cus;
        (void)hipGetLastError();
    }
    int rc = build_layers(*cfg, e->L);
    if (rc) { delete e; return rc; }
    e->NL = (int)e->L.size();
    e->I = cfg->enc_dims[0];
    e->Iin = e->I + cfg->cond_dim;
    e->Z = cfg->enc_dims[cfg->n_enc];
    e->Ip = rtx_pad(e->I);
    e->Zp = rtx_pad(e->Z);
    e->bf16 = cfg->numerics == RTX_BF16;
    e->vae = cfg->variant == RTX_VAE;
    e->esz = e->bf16 ? 2 : 4;
    e->Bp_alloc = rtx_pad_batch(cfg->max_batch);
    const size_t Bp = e->Bp_alloc, es = e->esz;
    size_t cacc = 0;
#define ALLOC(ptr, bytes)                                  \
    do {                                                   \
        rc = dev_alloc(e, (void**)&(ptr), (bytes));        \
        if (rc) { rtx_engine_destroy(e); return rc; }      \
    } while (0)
    for (int li = 0; li < e->NL; ++li) {
        Layer& l = e->L[li];
        ALLOC(l.Wsh, (size_t)l.outp * l.inp * es);
        if (e->bf16) ALLOC(l.Wsh_alt, (size_t)l.outp * l.inp * es);
        if (e->bf16 && li > 0 && li < e->NL - 1 && rtx_small_fwd_ok(l.outp)) ALLOC(l.WshT, (size_t)l.inp * l.outp * es);
        ALLOC(l.A, Bp * l.inp * es);
        if (li < e->NL - 1) ALLOC(l.O32, Bp * l.outp * sizeof(float));
        ALLOC(l.D, Bp * l.outp * es);
        // scratch for the forward output and the backward-data output of this layer (with split-K slabs), at any batch
        if (li < e->NL - 1) cacc = std::max(cacc, plan_cacc_elems(e, l.outp, l.inp));
        if (li > 0) cacc = std::max(cacc, plan_cacc_elems(e, l.inp, l.outp));
    }
    e->cacc_elems = cacc;
    ALLOC(e->Cacc, cacc * sizeof(float));
    ALLOC(e->Y, Bp * e->Ip * sizeof(float));
    ALLOC(e->mu32, Bp * e->Z * sizeof(float));
    ALLOC(e->lv32, Bp * e->Z * sizeof(float));
    ALLOC(e->eps32, Bp * e->Z * sizeof(float));
    e->lse_strips = e->Ip / 64;
    ALLOC(e->lse_part, Bp * e->lse_strips * sizeof(float2));
    ALLOC(e->tsum, Bp * sizeof(float));
    ALLOC(e->lse, Bp * sizeof(float));
    ALLOC(e->row_loss, Bp * rtx_dlogits_chunks(e->Ip) * sizeof(float));
    ALLOC(e->sumsq, sizeof(float) * 2 * RTX_MAX_LAYERS * 2);
    ALLOC(e->scratch_loss, sizeof(float) * 4);
#undef ALLOC
    *out = e;
    return RTX_OK;
}

int rtx_engine_destroy(rtx_engine* e)
{
    if (!e) return RTX_OK;
    (void)hipDeviceSynchronize();
    for (void* p : e->allocs) (void)hipFree(p);
    for (auto& kv : e->sites)
        for (auto& pr : kv.second.pending) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    for (hipEvent_t ev : e->event_pool) (void)hipEventDestroy(ev);
    for (hipEvent_t ev : e->ev_d)
        if (ev) (void)hipEventDestroy(ev);
    if (e->ev_done) (void)hipEventDestroy(e->ev_done);
    if (e->hop_mem) (void)hipFree(e->hop_mem);
    if (e->hopk_mem) (void)hipFree(e->hopk_mem);
    if (e->loss_mailbox) (void)hipHostFree(e->loss_mailbox);
    for (auto& kv : e->side_cache)
        if (kv.second.first) (void)hipStreamDestroy(kv.second.first);
    delete e;
    return RTX_OK;
}

int32_t rtx_engine_n_tensors(const rtx_engine* e) { return e ? 2 * e->NL : 0; }

int rtx_engine_tensor_shape(const rtx_engine* e, int32_t t, int32_t* rows, int32_t* cols)
{
    RTX_CHECK(e && t >= 0 && t < 2 * e->NL, RTX_EINVAL, "tensor index %d out of range", t);
    const Layer& l = e->L[t / 2];
    if (rows) *rows = l.out;
    if (cols) *cols = (t & 1) ? 1 : l.in;
    return RTX_OK;
}

int rtx_engine_bind(rtx_engine* e, float* const* params, float* const* grads, float* const* exp_avg, float* const* exp_avg_sq)
{
    RTX_CHECK(e && params, RTX_EINVAL, "engine_bind: NULL argument");
    const int n = 2 * e->NL;
    for (int t = 0; t < n; ++t) {
        RTX_CHECK(params[t], RTX_EINVAL, "engine_bind: params[%d] is NULL", t);
        RTX_CHECK(((uintptr_t)params[t] & 15) == 0, RTX_EINVAL, "engine_bind: params[%d] is not 16-byte aligned", t);
    }
    e->params.assign(params, params + n);
    e->can_train = grads && exp_avg && exp_avg_sq;
    if (e->can_train) {
        for (int t = 0; t < n; ++t) {
            RTX_CHECK(grads[t] && exp_avg[t] && exp_avg_sq[t], RTX_EINVAL, "engine_bind: NULL grad/moment pointer for tensor %d", t);
            RTX_CHECK((((uintptr_t)grads[t] | (uintptr_t)exp_avg[t] | (uintptr_t)exp_avg_sq[t]) & 15) == 0, RTX_EINVAL,
                      "engine_bind: tensor %d buffers are not 16-byte aligned", t);
        }
        e->grads.assign(grads, grads + n);
        e->m.assign(exp_avg, exp_avg + n);
        e->v.assign(exp_avg_sq, exp_avg_sq + n);
    }
    e->bound = true;
    e->shadows_valid = false;
    return RTX_OK;
}

int rtx_engine_bind_grads16(rtx_engine* e, uint16_t* const* grads_bf16)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    if (!grads_bf16) { e->grads16.clear(); return RTX_OK; }
    const int n = 2 * e->NL;
    for (int t = 0; t < n; ++t) {
        RTX_CHECK(grads_bf16[t], RTX_EINVAL, "bind_grads16: tensor %d is NULL", t);
        RTX_CHECK(((uintptr_t)grads_bf16[t] & 7) == 0, RTX_EINVAL, "bind_grads16: tensor %d is not 8-byte aligned", t);
    }
    e->grads16.assign(grads_bf16, grads_bf16 + n);
    return RTX_OK;
}

int rtx_engine_join(rtx_engine* e, void* stream)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    return resolve_join(e, (hipStream_t)stream);
}

int rtx_engine_sync_shadows(rtx_engine* e, void* stream)
{
    RTX_TRY(check_ready(e, false));
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(resolve_join(e, st));
    RtxAdamArgs a = {};
    fill_adam_tensors(e, a);
    a.update = 0;
    a.grad_scale = 1.f;
    TIMED("sync_shadows");
    RTX_TRY(rtx_launch_adam(a, e->bf16, st));
    e->shadows_valid = true;
    return RTX_OK;
}

// ---- forward family --------------------------------------------------------------------------------
int rtx_engine_forward(rtx_engine* e, const rtx_batch* batch, int32_t training, const rtx_step* step, int32_t remove_train,
                       float* logits, float* mu, float* logvar, void* stream)
{
    RTX_TRY(check_ready(e, false));
    RTX_CHECK(logits, RTX_EINVAL, "forward: logits is NULL");
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(ensure_shadows(e, st));
    RtxCsrView in = {}, tg = {};
    RTX_TRY(resolve_batch(e, batch, &in, &tg, st, 0));
    RTX_TRY(run_forward(e, &in, &in, batch->batch, training, step, 0, 0, e->NL, logits, e->I, mu, logvar, st));
    if (remove_train) {
        TIMED("neg_inf");
        RTX_TRY(rtx_launch_neg_inf(in, batch->batch, logits, e->I, e->I, st));
    }
    return RTX_OK;
}

int rtx_engine_encode(rtx_engine* e, const rtx_batch* batch, int32_t training, const rtx_step* step, float* out0, float* out1,
                      void* stream)
{
    RTX_TRY(check_ready(e, false));
    RTX_CHECK(out0, RTX_EINVAL, "encode: out0 is NULL");
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(ensure_shadows(e, st));
    RtxCsrView in = {}, tg = {};
    RTX_TRY(resolve_batch(e, batch, &in, &tg, st, 0));
    const int ne = e->cfg.n_enc;
    RTX_TRY(run_forward(e, &in, &in, batch->batch, training, step, 0, 0, ne, nullptr, 0, out0, out1, st));
    if (!e->vae) {
        const Layer& l = e->L[ne - 1];
        const long n = (long)batch->batch * l.out;
        hipLaunchKernelGGL(k_unpad_copy, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, l.O32, l.outp, batch->batch, l.out, out0);
        RTX_HIP(hipGetLastError());
    }
    return RTX_OK;
}

int rtx_engine_decode(rtx_engine* e, const float* z, int32_t batch, float* logits, void* stream)
{
    RTX_TRY(check_ready(e, false));
    RTX_CHECK(z && logits, RTX_EINVAL, "decode: NULL argument");
    RTX_CHECK(batch >= 1 && batch <= e->cfg.max_batch, RTX_EINVAL, "decode: batch %d outside [1,%d]", batch, e->cfg.max_batch);
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(ensure_shadows(e, st));
    const int ne = e->cfg.n_enc, Bp = rtx_pad_batch(batch);
    Layer& l = e->L[ne];
    const long n = (long)Bp * l.inp;
    if (e->bf16)
        hipLaunchKernelGGL(k_pad_convert<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, batch, l.in, (bf16_t*)l.A, l.inp, Bp);
    else
        hipLaunchKernelGGL(k_pad_convert<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, z, batch, l.in, (float*)l.A, l.inp, Bp);
    RTX_HIP(hipGetLastError());
    return run_forward(e, nullptr, nullptr, batch, 0, nullptr, 0, ne, e->NL, logits, e->I, nullptr, nullptr, st);
}

// ---- training --------------------------------------------------------------------------------------
// forward + loss + backward.  With `fuse` (single-GPU bf16 step) the Adam update of every weight matrix whose rows are
// a multiple of 4 floats runs INSIDE its weight-gradient kernel (dw_adam.hip: the gradient never reaches HBM and the
// optimizer's HBM traffic overlaps the matrix work); biases and the remaining tensors follow in one small launch.
// Otherwise the gradients land in the bound buffers (data-parallel exchange, p.grad, float32 parity mode).
// (the float32 parity mode stores its gradients and runs one multi-tensor k_adam launch.  Round 4 measured Adam as an epilogue of its
//  TN product too -- IEEE sqrt / divisions on 64 accumulators per lane, 4-byte accesses in the MFMA layout: 1.130 ms per ml-20m step
//  against 1.052 with the separate launch, whose 110 us of streaming it replaced by ~190 us of epilogues: dropped)
static bool layer_fusable(const rtx_engine* e, const Layer& l) { return e->bf16 && l.in >= 4; }   // (rows of in % 4 != 0 floats: the strided epilogue, dw_adam.hip)
static bool layer_is_big(const Layer& l) { return (long)l.out * l.in >= (1L << 20); }

// ---- the second stream of the step ------------------------------------------------------------------------------------------
// HIP maps streams onto a handful of hardware queues (GPU_MAX_HW_QUEUES, 4 by default) in creation order.  A process that also
// runs RCCL / torch.distributed has created a dozen streams before the engine's first step, and the engine's new stream can land
// on the SAME hardware queue as the caller's: its kernels then simply queue up behind / in front of the caller's and the step
// runs serially (rocprofv3 showed both streams on queue 1: 397 us/step against 343 -- profiles/r3_dp_priority_experiment.txt).
// So the stream is PROBED: a kernel that spins for ~150 us goes on the caller's stream, an empty kernel on the candidate; the
// candidate is kept if its kernel finishes while the spinner is still running.  Up to 8 candidates at normal priority, then
// one at the highest priority (a different queue pool); with none found the step falls back to one stream.
__global__ void k_probe_spin(unsigned long long ticks, int* sink)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();   // constant 100 MHz counter
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) {}
    if (sink && ticks == 0xffffffffffffffffull) *sink = 1;
}
__global__ void k_probe_nop() {}

static int make_side_stream(rtx_engine* e, hipStream_t st)
{
    RTX_HIP(hipStreamSynchronize(st));
    if (e->side) RTX_HIP(hipStreamSynchronize(e->side));   // (the previous caller's stream keeps its side stream in the cache)
    e->side = nullptr;
    e->side_for = st;
    auto hit = e->side_cache.find(st);
    if (hit != e->side_cache.end()) {
        e->side = hit->second.first;
        e->side_concurrent = hit->second.second;
        return RTX_OK;
    }
    int prio_least = 0, prio_greatest = 0;
    RTX_HIP(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
    if (e->opt_side_low_prio) {   // measurement knob: no probing
        RTX_HIP(hipStreamCreateWithPriority(&e->side, hipStreamNonBlocking, prio_least));
        e->side_concurrent = 1;
        e->side_cache[st] = {e->side, 1};
        return RTX_OK;
    }
    hipEvent_t ev_spin = nullptr, ev_cand = nullptr;
    RTX_HIP(hipEventCreateWithFlags(&ev_spin, hipEventDisableTiming));
    RTX_HIP(hipEventCreateWithFlags(&ev_cand, hipEventDisableTiming));
    std::vector<hipStream_t> rejected;
    hipStream_t found = nullptr;
    for (int attempt = 0; attempt < 9 && !found; ++attempt) {
        hipStream_t cand = nullptr;
        if (hipStreamCreateWithPriority(&cand, hipStreamNonBlocking, attempt < 8 ? 0 : prio_greatest) != hipSuccess) break;
        hipLaunchKernelGGL(k_probe_spin, dim3(1), dim3(64), 0, st, 15000ull, (int*)nullptr);   // 150 us
        (void)hipEventRecord(ev_spin, st);
        hipLaunchKernelGGL(k_probe_nop, dim3(1), dim3(64), 0, cand);
        (void)hipEventRecord(ev_cand, cand);
        (void)hipEventSynchronize(ev_cand);
        const bool concurrent = hipEventQuery(ev_spin) == hipErrorNotReady;   // the spinner is still at it: different hardware queues
        (void)hipStreamSynchronize(st);
        (void)hipGetLastError();
        if (concurrent) found = cand;
        else rejected.push_back(cand);
    }
    for (hipStream_t r : rejected) (void)hipStreamDestroy(r);
    (void)hipEventDestroy(ev_spin);
    (void)hipEventDestroy(ev_cand);
    e->side_concurrent = found != nullptr;
    if (!found) {
        static bool said = false;   // once per process: the step silently losing its second stream costs ~20 %
        if (!said) fprintf(stderr, "rectorch_hip: no HIP stream runs beside the caller's (all candidates share its hardware queue): the training step uses ONE stream\n");
        said = true;
        RTX_HIP(hipStreamCreateWithFlags(&found, hipStreamNonBlocking));   // (keeps the code paths alive; the step still orders everything by events)
    }
    e->side = found;
    e->side_cache[st] = {found, e->side_concurrent};
    return RTX_OK;
}

// tensors of the exchange buffer in layout order (DpState): W[NL-1], b[NL-1], ..., W[1], b[1], b[0], W[0]
int dp_layout_order(const rtx_engine* e, int* order)
{
    int n = 0;
    for (int li = e->NL - 1; li >= 1; --li) { order[n++] = 2 * li; order[n++] = 2 * li + 1; }
    order[n++] = 1;
    order[n++] = 0;
    return n;
}
size_t dp_region_elems(const rtx_engine* e, const DpState& d, int t)
{
    const Layer& l = e->L[t / 2];
    if (t & 1) return (size_t)l.out;
    return (size_t)(d.shard[t / 2] ? l.outp : l.out) * l.in;
}

// `to` continues only after everything enqueued on `from` so far: a write / wait pair of stream memory operations on word `slot` of
// the engine's signal memory (monotonic sequence numbers, compare >=), or an event record + wait
static int stream_dependency(rtx_engine* e, hipStream_t from, hipStream_t to, hipEvent_t ev, int slot)
{
    if (e->opt_hop_kernels) {
        RTX_TRY(ensure_hopk(e));
        const uint32_t v = ++e->hopk_seq;       // (compared as a signed difference: wraps after 2^31 hops without any reset)
        hipLaunchKernelGGL(k_hop_set, dim3(1), dim3(64), 0, from, e->hopk_mem + slot, v);
        hipLaunchKernelGGL(k_hop_wait, dim3(1), dim3(64), 0, to, e->hopk_mem + slot, v, e->hopk_mem + 8 + slot);
        RTX_HIP(hipGetLastError());
        return RTX_OK;
    }
    if (e->opt_hop_values) {
        if (!e->hop_mem) {
            int dev = 0, ok = 0;
            RTX_HIP(hipGetDevice(&dev));
            if (hipDeviceGetAttribute(&ok, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess || !ok ||
                hipExtMallocWithFlags((void**)&e->hop_mem, 64, hipMallocSignalMemory) != hipSuccess) {
                (void)hipGetLastError();
                e->hop_mem = nullptr;
                e->opt_hop_values = 0;   // not available here: events
            } else {
                RTX_HIP(hipMemset(e->hop_mem, 0, 64));
                RTX_HIP(hipStreamSynchronize(nullptr));   // (the NULL stream's memset must not land after a write of `from`, see dev_alloc)
            }
        }
        if (e->hop_mem) {
            // two numbers per step: 2^31 is reached after ~80 hours of 270-us steps.  Before the sequence gets there (whether the
            // device compares signed or unsigned) both streams drain and the words start again from zero.
            if (e->hop_seq >= e->hop_wrap) {
                RTX_HIP(hipStreamSynchronize(from));
                RTX_HIP(hipStreamSynchronize(to));
                RTX_HIP(hipMemset(e->hop_mem, 0, 64));
                RTX_HIP(hipStreamSynchronize(nullptr));   // (the NULL stream's memset must not land after a write of `from`, see dev_alloc)
                e->hop_seq = 0;
            }
            const uint32_t v = ++e->hop_seq;
            RTX_HIP(hipStreamWriteValue32(from, e->hop_mem + slot, v, 0));
            RTX_HIP(hipStreamWaitValue32(to, e->hop_mem + slot, v, hipStreamWaitValueGte, 0xffffffffu));
            return RTX_OK;
        }
    }
    RTX_HIP(hipEventRecord(ev, from));
    RTX_HIP(hipStreamWaitEvent(to, ev, 0));
    return RTX_OK;
}

// The batch announced for the NEXT step (rtx_engine_set_next_batch): its gather on the side stream, into the other image set.
// A hint: whatever keeps it from being issued is not an error (the next step then gathers for itself).
static int prefetch_next(rtx_engine* e)
{
    const rtx_batch& nb = e->next.b;
    if (!e->opt_prefetch || !e->bf16 || !e->side || !nb.csr || !nb.row_ids || nb.x_dense || nb.target_dense) return RTX_OK;
    if (nb.batch < 1 || nb.batch > e->cfg.max_batch || nb.csr->n_cols != e->Iin || nb.csr->max_row_len <= 0) return RTX_OK;
    if (nb.target_csr ? nb.target_csr->n_cols != e->I : e->Iin != e->I) return RTX_OK;
    RtxCsrView in = {}, tg = {};
    RTX_TRY(resolve_batch(e, &nb, &in, &tg, e->side));       // (a CSR batch: views only, nothing is enqueued)
    int64_t chunks = 0;
    if (sparse_in_ok(e, &in, rtx_pad_batch(nb.batch), &chunks)) return RTX_OK;   // the sparse first layer builds its own stream
    if (!e->A0_alt) {
        RTX_TRY(dev_alloc(e, &e->A0_alt, (size_t)e->Bp_alloc * e->L[0].inp * e->esz));
        RTX_TRY(dev_alloc(e, (void**)&e->tsum_alt, (size_t)e->Bp_alloc * sizeof(float)));
        e->img_exact_alt = false;
    }
    swap_img_sets(e);
    const int rc = gather_batch(e, &in, &tg, nb.batch, 1, &e->next.s, e->side);
    swap_img_sets(e);
    RTX_TRY(rc);
    e->pre.valid = true;
    e->pre.b = nb;
    e->pre.seed = e->next.s.seed; e->pre.offset = e->next.s.offset; e->pre.mask = e->next.s.dropout_mask;
    ++e->st_prefetch_issued;
    return RTX_OK;
}

static int loss_grads_impl(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out, float* loss_accum,
                           rtx_layer_cb cb, void* user, hipStream_t st, bool fuse, DpState* dp = nullptr, bool adam_inline = false)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step, RTX_EINVAL, "loss_grads: step is NULL");
    struct ClearNext { rtx_engine* e; ~ClearNext() { e->next.valid = false; } } clear_next{e};   // an announcement is for ONE step
    if (dp) {
        RTX_CHECK(!dp->broken, RTX_ESTATE, "data parallel: a collective of an earlier step failed; attach the plan again (rtx_engine_dp_attach)");
        dp->st_all_reduce = dp->st_reduce_scatter = dp->st_all_gather = 0;
        dp->st_collectives = 0;
    }
    const bool dae_reg = !e->vae && step->lam != 0.f;
    if (dp && dae_reg)   // lam * W / ||W|| needs the norm of the WHOLE matrix; a rank of the sharded optimizer holds current rows of its shard only
        for (int li = 0; li < e->NL; ++li)
            RTX_CHECK(!dp->shard[li], RTX_EINVAL, "data parallel: Mult-DAE's norm regulariser (lam != 0) needs whole master matrices; attach with sharded = 0");
    // a join the previous step left open (RTX_STEP_DEFER_JOIN): decided below, once it is known how this step starts
    bool join_open = e->join_pending && e->shadows_valid;
    // Until the wait for that join has really been enqueued (the one-wave kernel, or the first-layer product that carries it), every
    // early return below -- a wrong batch size, a failed launch -- must leave the join OPEN: the next entry point (rtx_engine_join,
    // predict, apply_adam, the next step) still has to wait for the side stream's weight kernel before it reads what that writes.
    struct JoinGuard {
        rtx_engine* e; bool armed;
        ~JoinGuard() { if (armed) { e->join_pending = true; e->join_fold = false; } }
    } join_guard{e, join_open};
    if (join_open) e->join_pending = false;
    RTX_TRY(ensure_shadows(e, st));
    RtxCsrView in = {}, tg = {};
    RTX_TRY(resolve_batch(e, batch, &in, &tg, st));
    const int B = batch->batch, Bp = rtx_pad_batch(B), NL = e->NL;
    // (see below for what the two streams do)
    // (data parallel: the second stream carries the decoder matrix's weight kernel, its exchange and its optimizer pass; the
    //  float32 parity mode keeps one compute copy per matrix and therefore one stream)
    bool two = (fuse || (dp && e->bf16)) && e->opt_two_stream;
    if (two && (!e->side || e->side_for != st)) {
        RTX_TRY(make_side_stream(e, st));
        if (!e->ev_done) {
            // (events created with hipEventReleaseToDevice -- a device-scope release at the record -- measure the same: 328.0 vs 327.7 us)
            for (int l = 0; l < NL + 1; ++l) RTX_HIP(hipEventCreateWithFlags(&e->ev_d[l], hipEventDisableTiming));
            RTX_HIP(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
        }
    }
    if (two && !e->side_concurrent) two = false;   // no stream that really runs beside the caller's: one stream, no event traffic
    // float32 train step: Adam inside the call, the big last layer's pass on the side stream (see below)
    bool adam_side = adam_inline && !fuse && !dp && !cb && !e->bf16 && e->opt_f32_adam_overlap && NL >= 2 && layer_is_big(e->L[NL - 1]);
    if (adam_side) {
        if (!e->side || e->side_for != st) RTX_TRY(make_side_stream(e, st));
        if (!e->ev_done) {
            for (int l = 0; l < NL + 1; ++l) RTX_HIP(hipEventCreateWithFlags(&e->ev_d[l], hipEventDisableTiming));
            RTX_HIP(hipEventCreateWithFlags(&e->ev_done, hipEventDisableTiming));
        }
        if (!e->side_concurrent) adam_side = false;
    }
    const int main_li = (two && fuse && e->opt_in_on_main && NL >= 2 && layer_is_big(e->L[0]) && layer_is_big(e->L[NL - 1]) && layer_fusable(e, e->L[0])) ? 0 : -1;
    if (e->pre.valid) {
        // the batch of this step was announced one step ago and gathered on the side stream under that step's last weight
        // kernel (the step's closing stream dependency ordered it before anything enqueued now): its image set becomes the
        // current one, the gather is skipped.  Anything else than exactly the announced batch / dropout stream: a normal step.
        const rtx_batch& pb = e->pre.b;
        const bool hit = (fuse || (dp && e->bf16)) && two && st == e->side_for && batch->csr && pb.csr == batch->csr && pb.row_ids == batch->row_ids &&
                         pb.target_csr == batch->target_csr && !batch->x_dense && !batch->target_dense && pb.batch == batch->batch &&
                         e->pre.seed == step->seed && e->pre.offset == step->offset && e->pre.mask == step->dropout_mask;
        e->pre.valid = false;
        if (hit) {
            swap_img_sets(e);
            e->gather_done = true;
            ++e->st_prefetch_hits;
            // this step starts with the first-layer product: the open join rides on it (run_forward; every workgroup checks the
            // number the side stream stored -- long ago -- before it touches the prefetched image)
            if (join_open && e->opt_hop_fold && e->bf16 && fold_has_room(e, Bp, e->L[0].outp, e->L[0].inp)) { e->join_fold = true; join_open = false; ++e->st_join_folds; }
        }
    }
    if (join_open) {   // any other start: a one-wave kernel in front of the step
        e->join_pending = true;
        RTX_TRY(resolve_join(e, st));
    }
    RTX_TRY(run_forward(e, &in, &tg, B, 1, step, 1, 0, NL, e->Y, e->Ip, nullptr, nullptr, st));
    join_guard.armed = false;   // the wait is on the stream (k_hop_wait above, or inside the first-layer product)
    if (dae_reg) {
        TIMED("sumsq");
        RTX_TRY(launch_sumsq(e, st));
    }
    // loss and d loss / d logits in one pass over Y
    {
        RtxDlogitsArgs a = {};
        a.loss.Y = e->Y; a.loss.ldy = e->Ip; a.loss.B = B; a.loss.I = e->I; a.loss.target = tg; a.loss.tsum = e->tsum;
        a.loss.lse = e->lse; a.loss.row_loss = e->row_loss; a.loss.inv_batch = step->inv_batch;
        if (e->opt_lse_fuse) { a.loss.part = e->lse_part; a.loss.n_strips = e->lse_strips; a.loss.part_ld = e->lse_strips; }
        if (e->vae) { a.loss.mu32 = e->mu32; a.loss.lv32 = e->lv32; a.loss.Z = e->Z; a.loss.beta = step->beta; }
        a.Bp = Bp; a.D = e->L[NL - 1].D; a.ldd = e->Ip;
        if (logits16_on(e)) a.Y16 = a.D;   // run_forward left half-precision logits there
        TIMED("dlogits_loss");
        RTX_TRY(rtx_launch_dlogits(a, e->bf16, st));
    }
    RtxAdamArgs rest = {};   // tensors whose Adam is NOT fused into a weight-gradient kernel (odd-width matrices + their biases)
    int rest_ids[RTX_MAX_TENSORS];
    rest.n = 0;
    // Fused step, two streams.  After the loss kernel the critical path would be
    //     dX chain (short latency-bound launches)  ->  every weight-gradient + Adam kernel (long streaming launches).
    // The weight kernel of layer l needs only D[l] and A[l], so the two BIG ones run on a side stream: the decoder matrix
    // beside the whole chain, the encoder matrix as soon as the chain has produced D[0]; the small layers' kernels follow
    // the chain on the caller's stream, beside the encoder matrix.  A big layer's fused optimizer writes the NEXT step's
    // compute copy (Wsh_alt; swapped at the end), because the chain still reads this step's.
    // The encoder matrix's kernel is the END of the step's critical path (it needs D[0], the last thing the chain produces, and
    // the next step's first product needs its result).  A cross-stream dependency costs about 18 us from the event's record to
    // the first workgroup of the waiting stream and a record about 7 us on the recording stream (profiles/r2_step_timeline.txt),
    // so that kernel stays on the CALLER's stream right behind the chain -- no hop before it, none after it -- and takes the small
    // layers' weight kernels with it in the same launch (as launches of their own beside it they crawl: 53 + 33 us).
    // (a hidden layer keeps ONE transposed compute copy, WshT, which its fused optimizer epilogue overwrites and the chain's
    // k_bwd_hidden reads: such a layer's weight kernel must stay behind the chain on the caller's stream)
    auto on_side = [&](int li) {
        if (dp) return two && NL >= 2 && li == NL - 1 && layer_is_big(e->L[li]) && !e->L[li].WshT;   // bucket A of the exchange
        return two && layer_is_big(e->L[li]) && li != main_li && !e->L[li].WshT;
    };
    const bool keep_grads = (step->flags & RTX_STEP_KEEP_GRADS) != 0;
    // tile of the weight-gradient kernels: 64 x 128 for the fused Adam epilogue (an HBM streaming kernel: many small workgroups);
    // the data-parallel step stores bf16 gradient images instead and is bound by operand delivery: 128 x 128 tiles halve the
    // operand bytes per parameter (emulated 8-rank step 262.3 vs 268.8 us, one box)
    // (not so the fused epilogue, even at K = 4096 batch rows: configs[3] on one GPU 1343 us/step with 64 x 128, 1380 with 128 x 128)
    const int dw_cfg = (dp && !e->opt_dw_cfg_set) ? RTX_DW_128x128 : e->opt_dw_cfg;
    // data parallel: where tensor t's gradient is produced (the exchange buffer, in comm dtype)
    auto xg16 = [&](int t) { return (bf16_t*)dp->xg + dp->xoff[t]; };
    auto xg32 = [&](int t) { return (float*)dp->xg + dp->xoff[t]; };
    auto reduce_loss = [&](hipStream_t ws) -> int {
        ScopedTimer tm(e, "reduce_loss", ws);
        const bool reg_in_loss = dae_reg && !(step->flags & RTX_STEP_NO_REG_IN_LOSS);
        return rtx_launch_reduce_loss(e->row_loss, B * rtx_dlogits_chunks(e->Ip), step->lam, reg_in_loss ? e->sumsq : nullptr, 2 * NL, loss_out,
                                      loss_accum, ws, e->loss_mailbox, e->loss_mailbox ? ++e->loss_ticket : 0u, (uint32_t)step->step);
    };
    // weight + bias gradient of layer li on stream ws: gW[out][in] = D[Bp][outp]^T x A[Bp][inp] (both read K-major); column
    // `in` of the product (the ones column of A) is the bias gradient
    // bf16: the weight-gradient problem of layer li (fused with Adam where the layer allows it)
    auto make_dw = [&](int li, RtxDw& d) -> bool {
        Layer& l = e->L[li];
        const bool fused = fuse && layer_fusable(e, l);
        d = RtxDw{};
        d.A = l.D; d.lda = l.outp; d.B = l.A; d.ldb = l.inp;
        d.m_tiles = l.outp / rtx_dw_tile_rows(dw_cfg); d.n_tiles = (l.inp + rtx_dw_tile_cols(dw_cfg) - 1) / rtx_dw_tile_cols(dw_cfg); d.k_slices = Bp / 64;
        d.M_real = l.out; d.N_real = l.in;
        if (fused) {
            RtxAdamArgs sc = {};
            fill_adam_scalars(e, step, sc, 2 * li);
            const bool keep = (step->flags & RTX_STEP_KEEP_GRADS) != 0;
            d.adam.p = e->params[2 * li]; d.adam.m = e->m[2 * li]; d.adam.v = e->v[2 * li];
            d.adam.gkeep = keep ? e->grads[2 * li] : nullptr;
            d.gbias = keep ? e->grads[2 * li + 1] : nullptr;
            d.adam.sh = on_side(li) ? l.Wsh_alt : l.Wsh; d.adam.shT = l.WshT; d.adam.ld_sh = l.inp; d.adam.ld_shT = l.WshT ? l.outp : 0;
            d.adam.step_size = sc.step_size; d.adam.bc2_sqrt = sc.bc2_sqrt; d.adam.beta1 = sc.beta1; d.adam.beta2 = sc.beta2;
            d.adam.eps = sc.eps; d.adam.weight_decay = sc.weight_decay; d.adam.lam = sc.lam;
            d.adam.sumsq = dae_reg ? e->sumsq + 2 * li : nullptr;
            d.bias_p = e->params[2 * li + 1]; d.bias_m = e->m[2 * li + 1]; d.bias_v = e->v[2 * li + 1];
            d.bias_sumsq = dae_reg ? e->sumsq + 2 * li + 1 : nullptr;
        } else if (dp) {
            // the gradient leaves the kernel as the image the exchange sends; RTX_STEP_KEEP_GRADS also stores this rank's own
            // (unreduced) float32 gradient in the bound buffers
            if (dp->cfg.comm_dtype == RTX_BF16) {
                d.g16 = xg16(2 * li); d.gbias16 = xg16(2 * li + 1);
                if (keep_grads) { d.gW = e->grads[2 * li]; d.gbias = e->grads[2 * li + 1]; }
            } else {
                d.gW = xg32(2 * li); d.gbias = xg32(2 * li + 1);
            }
        } else if ((step->flags & RTX_STEP_GRADS_BF16) && !e->grads16.empty()) {
            // data-parallel bf16 exchange: the gradient leaves the kernel as the bf16 image the all-reduce sends (no float32
            // store, no cast pass)
            d.g16 = (bf16_t*)e->grads16[2 * li]; d.gbias16 = (bf16_t*)e->grads16[2 * li + 1];
        } else {
            d.gW = e->grads[2 * li]; d.gbias = e->grads[2 * li + 1];
        }
        return fused;
    };
    auto weight_grad = [&](int li, hipStream_t ws) -> int {
        Layer& l = e->L[li];
        const bool fused = fuse && layer_fusable(e, l);
        const char* site = li == NL - 1 ? (fused ? "dW_adam_out" : "gemm_dW_out") : (li == 0 ? (fused ? "dW_adam_in" : "gemm_dW_in") : (fused ? "dW_adam_hidden" : "gemm_dW_hidden"));
        ScopedTimer tm(e, site, ws);
        if (e->bf16) {
            RtxDw d;
            make_dw(li, d);
            // the fused step's side-stream launch runs beside the data-gradient chain: one workgroup per CU leaves the chain room (RtxDw::lds_pad)
            if (fused && !dp && two && ws == e->side && ws != st && dw_cfg == RTX_DW_64x128) d.lds_pad = e->opt_dw_side_pad;
            return rtx_dw_launch(d, fused ? RTX_DW_ADAM : RTX_DW_GRAD, dw_cfg, ws);
        }
        RtxGemm g = {};
        g.form = RTX_FORM_TN;
        g.A = l.D; g.lda = l.outp; g.B = l.A; g.ldb = l.inp;
        g.k_slices = Bp / 32; g.tile_shape = RTX_TILE_128x128; g.m_tiles = l.outp / 128; g.n_tiles = l.inp / 128;
        g.splits = 1; g.C = e->grads[2 * li]; g.gbias = e->grads[2 * li + 1];
        if (dp && dp->cfg.comm_dtype == RTX_FP32) { g.C = xg32(2 * li); g.gbias = xg32(2 * li + 1); }
        g.M_real = l.out; g.N_real = l.in;
        // A hidden layer's gradient is 10-20 tiles of 128 x 128 with K = the batch: one workgroup per tile walks 16 K slices at the
        // f32 MFMA rate of ONE CU (1.7 us per slice) while 240 CUs idle -- 40 us per launch, 80 us of the 1.05-ms float32 step
        // (profiles/r4_fp32_step_timeline.txt).  Such products are split over the batch into slabs and summed in a fixed order.
        const int tiles = g.m_tiles * g.n_tiles;
        int sp = std::min(8, g.k_slices / 2);
        while (sp > 1 && (sp - 1) * ((g.k_slices + sp - 1) / sp) >= g.k_slices) --sp;   // no empty split
        // (Cacc is shared with the data-gradient products: safe because the float32 step runs every PRODUCT on the caller's stream --
        //  the only kernel it ever puts on the side stream is k_adam, which does not touch the scratch)
        if (e->opt_f32_dw_split && tiles <= 64 && sp >= 2 && (size_t)sp * l.outp * l.inp <= e->cacc_elems) {
            float* gW = g.C;
            float* gb = g.gbias;
            g.splits = sp; g.C = e->Cacc; g.ldc = l.inp; g.slab_stride = (long)l.outp * l.inp; g.gbias = nullptr;
            RTX_TRY(rtx_gemm_f32_km_launch(g, RTX_EPI_STORE, ws));
            return rtx_launch_dw_slab_reduce(e->Cacc, sp, g.slab_stride, g.ldc, l.out, l.in, gW, gb, ws);
        }
        // The last partial wave (round 5): 158 x 5 = 790 tiles on 256 CUs are three full rounds and a fourth that is 9 % full.  The tiles
        // beyond the last full round (along the long tile dimension) go to extra workgroups, 8 per tile over K, appended to the same
        // launch; a small fixed-order reduction writes their gradient (gemm_f32.hip, RtxGemm::tail_*).
        if (e->opt_f32_tail_split && tiles > e->n_cus && g.k_slices >= 16) {
            const int long_tiles = std::max(g.m_tiles, g.n_tiles), short_tiles = std::min(g.m_tiles, g.n_tiles);
            const int t0 = (tiles / e->n_cus * e->n_cus) / short_tiles;       // tiles before it fill whole rounds
            const int tail_tiles = (long_tiles - t0) * short_tiles;
            const bool m_long = g.m_tiles > g.n_tiles;
            const int S = 8;
            const long rows = m_long ? (long)(long_tiles - t0) * 128 : (long)l.outp, cols = m_long ? (long)l.inp : (long)(long_tiles - t0) * 128;
            // worth it when the last round is less than half full, and the slabs fit the scratch
            if (g.m_tiles != g.n_tiles && t0 > 0 && tail_tiles > 0 && tail_tiles * 2 <= e->n_cus && (size_t)S * rows * cols <= e->cacc_elems) {
                g.tail_t0 = t0; g.tail_splits = S; g.tail_C = e->Cacc; g.tail_ldc = cols; g.tail_slab_stride = rows * cols;
                RTX_TRY(rtx_gemm_f32_km_launch(g, RTX_EPI_GRAD, ws));
                return rtx_launch_tail_reduce(e->Cacc, S, g.tail_slab_stride, cols, (int)rows, (int)cols, m_long ? t0 * 128 : 0, m_long ? 0 : t0 * 128,
                                              l.out, l.in, g.C, g.gbias, ws);
            }
        }
        return rtx_gemm_f32_km_launch(g, RTX_EPI_GRAD, ws);
    };
    // ---- data parallel: exchange + optimizer of layers [l_lo, l_hi) on stream ws; `alt`: the big matrices' Adam writes the NEXT
    //      step's compute copy (the chain on the other stream still reads this step's) -------------------------------------------
    auto dp_bucket = [&](int l_lo, int l_hi, hipStream_t ws, bool alt) -> int {
        DpState& d = *dp;
        const int cdt = d.cfg.comm_dtype;
        // the side stream's bucket talks through its own communicator: RCCL orders the operations of ONE communicator in issue
        // order across streams, which would make bucket B's reduce (caller's stream) wait for bucket A's all-gather
        const rtx_dp_ops& O = (ws == e->side && two) ? d.ops_side : d.ops;
        int order[2 * 2 * RTX_MAX_LAYERS];
        const int n_order = dp_layout_order(e, order);
        auto in_bucket = [&](int t) { return t / 2 >= l_lo && t / 2 < l_hi; };
        // (1) staging: float32 numerics with a bf16 exchange cast their gradients; a float32 exchange hands the caller its copy
        for (int q = 0; q < n_order; ++q) {
            const int t = order[q];
            if (!in_bucket(t)) continue;
            const Layer& l = e->L[t / 2];
            const size_t n = (t & 1) ? (size_t)l.out : (size_t)l.out * l.in;
            if (!e->bf16 && cdt == RTX_BF16) RTX_TRY(rtx_launch_cast_f32_bf16(e->grads[t], xg16(t), (long)n, ws));
            else if (keep_grads && cdt == RTX_FP32) RTX_HIP(hipMemcpyAsync(e->grads[t], xg32(t), n * sizeof(float), hipMemcpyDeviceToDevice, ws));
        }
        // (2) exchange: reduce-scatter of every sharded matrix, one all-reduce per contiguous run of replicated tensors.
        //     A failure between group_start and group_end still closes the group (an open RCCL group would swallow every later
        //     collective of the communicator) and marks the plan unusable until it is attached again.
        {
            ScopedTimer tm(e, ws == e->side && two ? "dp_exchange_side" : "dp_exchange_main", ws);
            if (O.group_start) RTX_CHECK(O.group_start(O.ctx) == 0, RTX_EHIP, "data parallel: group_start failed: %s", rtx_last_error_str());
            auto in_group = [&]() -> int {
                long run_lo = -1, run_hi = -1;
                auto flush = [&]() -> int {
                    if (run_lo >= 0 && run_hi > run_lo) {
                        RTX_CHECK(O.all_reduce(O.ctx, (char*)d.xg + (size_t)run_lo * d.xesz, run_hi - run_lo, cdt, ws) == 0, RTX_EHIP,
                                  "data parallel: all_reduce failed: %s", rtx_last_error_str());
                        d.st_all_reduce += (int64_t)(run_hi - run_lo) * (int64_t)d.xesz;
                        d.st_collectives += 1;
                    }
                    run_lo = run_hi = -1;
                    return RTX_OK;
                };
                for (int q = 0; q < n_order; ++q) {
                    const int t = order[q];
                    const bool sharded_w = !(t & 1) && d.shard[t / 2];
                    if (!in_bucket(t) || sharded_w) {
                        RTX_TRY(flush());
                        if (in_bucket(t)) {
                            RTX_CHECK(O.reduce_scatter(O.ctx, (char*)d.xg + d.xoff[t] * d.xesz, (int64_t)dp_region_elems(e, d, t), cdt, ws) == 0,
                                      RTX_EHIP, "data parallel: reduce_scatter failed: %s", rtx_last_error_str());
                            d.st_reduce_scatter += (int64_t)dp_region_elems(e, d, t) * (int64_t)d.xesz;
                            d.st_collectives += 1;
                        }
                        continue;
                    }
                    if (run_lo < 0) run_lo = (long)d.xoff[t];
                    run_hi = (long)(d.xoff[t] + dp_region_elems(e, d, t));
                }
                return flush();
            };
            const int rc = in_group();
            if (rc != RTX_OK) {
                d.broken = true;
                std::string msg = rtx_last_error_str();           // group_end may overwrite the thread's error slot
                if (O.group_end) (void)O.group_end(O.ctx);
                RTX_CHECK(false, rc, "%s", msg.c_str());
            }
            if (O.group_end && O.group_end(O.ctx) != 0) {
                d.broken = true;
                RTX_CHECK(false, RTX_EHIP, "data parallel: group_end failed: %s", rtx_last_error_str());
            }
        }
        // (3) Adam: replicated tensors in full, a sharded matrix on this rank's rows
        RtxAdamArgs a = {};
        int ids[RTX_MAX_TENSORS];
        for (int li = l_lo; li < l_hi; ++li) {
            Layer& l = e->L[li];
            RtxAdamArgs one = {};
            fill_adam_tensors(e, one, li, li + 1);
            RtxAdamTensor w = one.t[0], b = one.t[1];
            if (cdt == RTX_BF16) { w.g16 = xg16(2 * li); b.g16 = xg16(2 * li + 1); }
            else { w.g = xg32(2 * li); b.g = xg32(2 * li + 1); }
            if (alt && l.Wsh_alt) w.sh = l.Wsh_alt;
            bool any_w = true;
            if (d.shard[li]) {
                const int per = l.outp / d.cfg.world;
                const int lo = d.cfg.rank * per, hi = std::min((d.cfg.rank + 1) * per, l.out);   // padding rows hold no parameters
                any_w = lo < hi;
                const size_t off = (size_t)lo * l.in;
                w.p += off; w.m += off; w.v += off;
                if (w.g16) w.g16 += off; else w.g += off;
                w.sh = (char*)w.sh + (size_t)lo * l.inp * e->esz;
                w.rows = any_w ? hi - lo : 0;
            }
            if (any_w) { ids[a.n] = 2 * li; a.t[a.n++] = w; }
            ids[a.n] = 2 * li + 1; a.t[a.n++] = b;
        }
        fill_adam_scalars(e, step, a, 0, ids);
        {
            ScopedTimer tm(e, "adam", ws);
            RTX_TRY(rtx_launch_adam(a, e->bf16, ws));
        }
        // (4) the other ranks' rows of the compute copy
        for (int li = l_lo; li < l_hi; ++li)
            if (d.shard[li]) {
                Layer& l = e->L[li];
                ScopedTimer tm(e, ws == e->side && two ? "dp_allgather_side" : "dp_allgather_main", ws);
                if (O.all_gather(O.ctx, (alt && l.Wsh_alt) ? l.Wsh_alt : l.Wsh, (int64_t)((size_t)l.outp * l.inp * e->esz), ws) != 0) {
                    d.broken = true;
                    RTX_CHECK(false, RTX_EHIP, "data parallel: all_gather failed: %s", rtx_last_error_str());
                }
                d.st_all_gather += (int64_t)((size_t)l.outp * l.inp * e->esz);
                d.st_collectives += 1;
            }
        return RTX_OK;
    };
    const bool dp_side = dp && on_side(NL - 1);
    bool fold_hop = false;          // this layer's fork is folded into its data-gradient product (below)
    uint32_t fold_seq = 0;
    // what the side stream does for layer li once D[li] is there: the long weight kernel, and under data parallelism bucket A
    auto side_work = [&](int li) -> int {
        RTX_TRY(weight_grad(li, e->side));
        if (dp) {   // bucket A: the decoder matrix's exchange and optimizer pass run beside the chain; the loss sum rides along
            RTX_TRY(reduce_loss(e->side));
            RTX_TRY(dp_bucket(li, li + 1, e->side, true));
        }
        return RTX_OK;
    };
    if (!two) RTX_TRY(reduce_loss(st));
    for (int li = NL - 1; li >= 0; --li) {
        Layer& l = e->L[li];
        if (on_side(li)) {   // the long kernel first: it only needs D[li], which exists now
            // The fork (round 5, "hop_fold"): the side stream waits in a one-wave kernel (k_hop_wait) for a number that the NEXT kernel
            // of the caller's stream -- this layer's data-gradient product, which follows the producers of D[li] in order -- stores as
            // its first instruction.  The caller's stream, which carries the step's critical path, gets no packet of its own: the
            // 6-9 us gap behind k_dlogits (profiles/r4_step_timeline.txt) goes.
            // The number is stored by a kernel that is enqueued AFTER this point, so the side stream's work is enqueued behind it
            // (side_work below): a host that blocks on the side stream in between -- the gloo test transport drains the device inside
            // its collectives -- would otherwise wait for a number nobody has been told to write yet.
            fold_hop = e->opt_hop_fold && e->bf16 && li > 0 && !(li < NL - 1 && l.WshT && e->opt_small_bwd) &&
                       !plan_gemm(e, Bp, l.inp, l.outp, RTX_FORM_NN).regstage;
            if (fold_hop) {
                RTX_TRY(ensure_hopk(e));
                fold_seq = ++e->hopk_seq;
            } else {
                RTX_TRY(stream_dependency(e, st, e->side, e->ev_d[li], 0));
                RTX_TRY(side_work(li));
            }
        }
        // data gradient: dA[Bp][inp] = D[Bp][outp] x Wsh[outp][inp]   (Wsh read K-major).  On ONE stream it must come before
        // the weight kernel of this layer, whose fused optimizer epilogue overwrites the compute copy.
        if (li > 0 && li < NL - 1 && l.WshT && e->opt_small_bwd) {
            // a hidden layer: product with the transposed compute copy + the activation derivative (or the VAE head's
            // backward) + the bf16 gradient of the layer below in one launch (small_layers.hip)
            Layer& pv = e->L[li - 1];
            RtxSmallBwdArgs a = {};
            a.D = (const bf16_t*)l.D; a.WT = (const bf16_t*)l.WshT; a.ld = l.outp; a.wt_rows = l.inp;
            a.B = B; a.Bp = Bp; a.Np = pv.outp; a.Dout = (bf16_t*)pv.D;
            if (e->vae && li == e->cfg.n_enc) {
                a.Z = e->Z; a.training = 1; a.mu32 = e->mu32; a.lv32 = e->lv32; a.eps32 = e->eps32;
                a.beta = step->beta; a.inv_batch = step->inv_batch;
            } else {
                a.N_real = pv.out; a.tanh_act = pv.tanh_act; a.O32 = pv.O32;
            }
            TIMED(a.Z ? "bwd_head" : "bwd_hidden");
            RTX_TRY(rtx_launch_small_bwd(a, st));
        } else if (li > 0) {
            int splits = 1;
            {
                TIMED(li == NL - 1 ? "gemm_dX_out" : "gemm_dX_hidden");
                RTX_TRY(gemm_to_cacc(e, RTX_FORM_NN, l.D, l.outp, l.Wsh, l.inp, Bp, l.inp, l.outp, &splits, st, fold_hop ? e->hopk_mem + 2 : nullptr, fold_seq));
                if (fold_hop) {   // the product that stores the number is enqueued: now the side stream's wait and its work
                    fold_hop = false;
                    hipLaunchKernelGGL(k_hop_wait, dim3(1), dim3(64), 0, e->side, e->hopk_mem + 2, fold_seq, e->hopk_mem + 10);
                    RTX_HIP(hipGetLastError());
                    RTX_TRY(side_work(li));
                }
            }
            Layer& pv = e->L[li - 1];
            if (e->vae && li == e->cfg.n_enc) {
                RtxVaeBwdArgs a = {};
                a.C = e->Cacc; a.splits = splits; a.slab_stride = (long)Bp * l.inp; a.ldc = l.inp;
                a.B = B; a.Bp = Bp; a.Z = e->Z; a.Np = pv.outp;
                a.mu32 = e->mu32; a.lv32 = e->lv32; a.eps32 = e->eps32; a.training = 1;
                a.beta = step->beta; a.inv_batch = step->inv_batch; a.D = pv.D;
                TIMED("vae_head_bwd");
                RTX_TRY(rtx_launch_vae_bwd(a, e->bf16, st));
            } else {
                RtxPostArgs a = {};
                a.C = e->Cacc; a.splits = splits; a.slab_stride = (long)Bp * l.inp; a.ldc = l.inp;
                a.B = B; a.Bp = Bp; a.N_real = pv.out; a.Np = pv.outp;
                a.tanh_act = pv.tanh_act; a.O32 = pv.O32; a.R = pv.D;
                TIMED("post_bwd");
                RTX_TRY(rtx_launch_post(a, RTX_POST_BWD, e->bf16, st));
            }
        }
        if (!two) RTX_TRY(weight_grad(li, st));
        if (adam_side && li == NL - 1) {
            // float32 train step (round 5): the decoder matrix's Adam pass -- half of the optimizer's 690 MB, HBM-bound -- leaves for
            // the side stream as soon as its gradient exists and runs under the remaining float32 products, which are MFMA-bound
            // (its compute copy has no reader left in this step: the data gradient of this layer came first on this stream)
            RTX_TRY(stream_dependency(e, st, e->side, e->ev_d[li], 0));
            RtxAdamArgs a = {};
            fill_adam_tensors(e, a, li, li + 1);
            fill_adam_scalars(e, step, a, 2 * li);
            ScopedTimer tm(e, "adam", e->side);
            RTX_TRY(rtx_launch_adam(a, e->bf16, e->side));
        }
        if (fuse && !layer_fusable(e, l)) {   // what is left for the multi-tensor Adam launch at the end of the step
            RtxAdamArgs one = {};
            fill_adam_tensors(e, one, li, li + 1);
            rest_ids[rest.n] = 2 * li; rest.t[rest.n++] = one.t[0];
            rest_ids[rest.n] = 2 * li + 1; rest.t[rest.n++] = one.t[1];
        }
        if (cb) cb(li, user);
    }
    hipStream_t rs = st;    // the stream the leftover Adam launch runs on
    if (dp) {
        // bucket B behind the chain on the caller's stream: the remaining weight kernels (bf16: grouped launches), their exchange,
        // their optimizer pass -- the END of the step's critical path, so no stream hop before or between them
        const int b_hi = dp_side ? NL - 1 : NL;
        if (two) {
            RtxDw grp[RTX_DW_GROUP_MAX];
            int ng = 0;
            for (int li = b_hi - 1; li >= 0; --li) {
                make_dw(li, grp[ng++]);
                if (ng == RTX_DW_GROUP_MAX || li == 0) {
                    ScopedTimer tm(e, "gemm_dW_in", st);
                    RTX_TRY(rtx_dw_launch_group(grp, ng, RTX_DW_GRAD, dw_cfg, st));
                    ng = 0;
                }
            }
            if (!dp_side) RTX_TRY(reduce_loss(st));
        }
        RTX_TRY(dp_bucket(0, b_hi, st, false));
        if (dp_side) {
            std::swap(e->L[NL - 1].Wsh, e->L[NL - 1].Wsh_alt);
            // (round 6) the side stream has finished bucket A long before bucket B's exchange ends: the NEXT step's gather goes there,
            // behind bucket A and in front of the join -- the data-parallel step then starts with its first-layer product as well
            if (e->next.valid) RTX_TRY(prefetch_next(e));
            if ((step->flags & RTX_STEP_DEFER_JOIN) && e->opt_hop_fold && e->bf16) {
                // (round 6) as the single-GPU step: the side stream stores a number behind its last kernel, and whoever uses the engine
                // next waits for it -- the next training step inside its first-layer product
                RTX_TRY(ensure_hopk(e));
                e->join_seq = ++e->hopk_seq;
                hipLaunchKernelGGL(k_hop_set, dim3(1), dim3(64), 0, e->side, e->hopk_mem + 3, e->join_seq);
                RTX_HIP(hipGetLastError());
                e->join_pending = true;
            } else {
                // (the same form of dependency as the single-GPU step's join: stream values where the device has them, else the event)
                RTX_TRY(stream_dependency(e, e->side, st, e->ev_done, 1));
            }
        }
        e->shadows_valid = true;
    } else if (two && main_li >= 0) {
        // behind the chain, on this stream: the encoder matrix's kernel and the small layers' (their compute copies have no
        // reader left) ...
        // ... ONE launch for the encoder matrix and the small fusable layers (small problems first); small layers that are not
        // fusable store their gradients first, for the leftover Adam launch.  The loss reduction only needs what the loss kernel
        // wrote: it goes behind the decoder matrix's kernel on the side stream (no new event).
        RtxDw grp[RTX_DW_GROUP_MAX];
        int ng = 0;
        for (int li = NL - 1; li >= 1; --li) {
            if (on_side(li)) continue;
            if (fuse && layer_fusable(e, e->L[li]) && ng < RTX_DW_GROUP_MAX - 1) make_dw(li, grp[ng++]);
            else RTX_TRY(weight_grad(li, st));
        }
        make_dw(main_li, grp[ng++]);
        {
            ScopedTimer tm(e, "dW_adam_in", st);
            RTX_TRY(rtx_dw_launch_group(grp, ng, RTX_DW_ADAM, dw_cfg, st));
        }
        RTX_TRY(reduce_loss(e->side));
        // the side stream idles from here to the end of the step: the NEXT step's gather, when its batch was announced
        if (e->next.valid && !rest.n) RTX_TRY(prefetch_next(e));
        if (rest.n > 0) {   // gradients from both streams feed the leftover Adam launch: the side stream waits for this one, then runs it
            RTX_HIP(hipEventRecord(e->ev_d[NL], st));
            RTX_HIP(hipStreamWaitEvent(e->side, e->ev_d[NL], 0));
            rs = e->side;
        }
    } else if (two) {
        // behind the chain, beside the encoder matrix's kernel: the small layers' weight kernels (their compute copies have
        // no reader left on this stream) and the loss reduction
        for (int li = NL - 1; li >= 0; --li)
            if (!on_side(li)) RTX_TRY(weight_grad(li, st));
        RTX_TRY(reduce_loss(st));
        if (rest.n > 0) {   // gradients from both streams feed it: the side stream waits for this one, then runs it
            RTX_HIP(hipEventRecord(e->ev_d[NL], st));
            RTX_HIP(hipStreamWaitEvent(e->side, e->ev_d[NL], 0));
            rs = e->side;
        }
    }
    if (adam_inline && !fuse && !dp) {
        // the rest of the optimizer on the caller's stream (everything when the side stream took nothing), then the join
        RtxAdamArgs a = {};
        fill_adam_tensors(e, a, 0, adam_side ? NL - 1 : NL);
        fill_adam_scalars(e, step, a, 0);
        {
            TIMED("adam");
            RTX_TRY(rtx_launch_adam(a, e->bf16, st));
        }
        if (adam_side) RTX_TRY(stream_dependency(e, e->side, st, e->ev_done, 1));
        e->shadows_valid = true;
    }
    if (fuse) {
        if (rest.n > 0) {
            fill_adam_scalars(e, step, rest, 0, rest_ids);
            ScopedTimer tm(e, "adam_small", rs);
            RTX_TRY(rtx_launch_adam(rest, e->bf16, rs));
        }
        if (two) {
            for (int li = 0; li < NL; ++li)
                if (on_side(li) && layer_fusable(e, e->L[li])) std::swap(e->L[li].Wsh, e->L[li].Wsh_alt);
            if ((step->flags & RTX_STEP_DEFER_JOIN) && e->opt_hop_fold && e->bf16 && !dp) {
                // the caller will not touch parameters / losses outside the engine before its next engine call (or rtx_engine_join):
                // the side stream stores a number behind its last kernel, and whoever uses the engine next waits for it -- the next
                // training step inside its first kernel (no packet, no gap between two steps on the caller's stream)
                RTX_TRY(ensure_hopk(e));
                e->join_seq = ++e->hopk_seq;
                hipLaunchKernelGGL(k_hop_set, dim3(1), dim3(64), 0, e->side, e->hopk_mem + 3, e->join_seq);
                RTX_HIP(hipGetLastError());
                e->join_pending = true;
            } else {
                // everything the step did is ordered on the caller's stream when the call returns
                RTX_TRY(stream_dependency(e, e->side, st, e->ev_done, 1));
            }
        }
        e->shadows_valid = true;
    }
    return RTX_OK;
}

int rtx_engine_loss_grads(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out, float* loss_accum,
                          rtx_layer_cb cb, void* user, void* stream)
{
    return loss_grads_impl(e, batch, step, loss_out, loss_accum, cb, user, (hipStream_t)stream, false);
}

int rtx_engine_apply_adam(rtx_engine* e, const rtx_step* step, void* stream)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "apply_adam: step count must be >= 1");
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(resolve_join(e, st));
    RtxAdamArgs a = {};
    fill_adam_tensors(e, a);
    fill_adam_scalars(e, step, a, 0);
    TIMED("adam");
    RTX_TRY(rtx_launch_adam(a, e->bf16, st));
    e->shadows_valid = true;
    return RTX_OK;
}

int rtx_engine_apply_adam_layers(rtx_engine* e, const rtx_step* step, int32_t layer_lo, int32_t layer_hi,
                                 const uint16_t* const* grads_bf16, void* stream)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "apply_adam: step count must be >= 1");
    RTX_CHECK(layer_lo >= 0 && layer_lo < layer_hi && layer_hi <= e->NL, RTX_EINVAL, "apply_adam_layers: bad layer range [%d, %d) of %d",
              layer_lo, layer_hi, e->NL);
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(resolve_join(e, st));
    RtxAdamArgs a = {};
    fill_adam_tensors(e, a, layer_lo, layer_hi);
    if (grads_bf16)
        for (int t = 0; t < a.n; ++t) a.t[t].g16 = grads_bf16[2 * layer_lo + t];
    fill_adam_scalars(e, step, a, 2 * layer_lo);
    TIMED("adam");
    RTX_TRY(rtx_launch_adam(a, e->bf16, st));
    // the compute copies are whole again once every layer has been visited; callers cover [0, NL) each step
    e->shadows_valid = true;
    return RTX_OK;
}

// Sharded optimizer (data parallel, ZeRO-1 style): this rank owns rows [row_lo, row_hi) of layer `layer`'s weight matrix.
// After the reduce-scatter of the gradient it updates only those rows (p, exp_avg, exp_avg_sq and the rows of the compute
// copy) -- the optimizer's 28 B/param of HBM traffic shrink by the number of ranks -- and, with with_bias, the (replicated,
// all-reduced) bias in full.  The caller then all-gathers the compute copy (rtx_engine_shadow_region).
int rtx_engine_apply_adam_rows(rtx_engine* e, const rtx_step* step, int32_t layer, int32_t row_lo, int32_t row_hi, int32_t with_bias,
                               const uint16_t* w_grad_bf16, const uint16_t* b_grad_bf16, void* stream)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "apply_adam_rows: step count must be >= 1");
    RTX_CHECK(layer >= 0 && layer < e->NL, RTX_EINVAL, "apply_adam_rows: layer %d out of range", layer);
    const Layer& l = e->L[layer];
    RTX_CHECK(row_lo >= 0 && row_lo <= row_hi && row_hi <= l.outp, RTX_EINVAL, "apply_adam_rows: bad row range [%d, %d) of %d", row_lo, row_hi, l.outp);
    // a hidden layer keeps a transposed compute copy (WshT, a column-block layout the caller's row-block all-gather cannot
    // complete): its rows cannot be updated piecewise
    RTX_CHECK(!l.WshT || (row_lo == 0 && row_hi >= l.out), RTX_EINVAL,
              "apply_adam_rows: layer %d keeps a transposed compute copy and cannot be sharded by rows (shard the first / last layer only)", layer);
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(resolve_join(e, st));
    RtxAdamArgs full = {}, a = {};
    fill_adam_tensors(e, full, layer, layer + 1);
    int ids[2];
    const int hi = std::min(row_hi, l.out);     // padding rows hold no parameters
    if (row_lo < hi) {
        RtxAdamTensor w = full.t[0];
        const size_t off = (size_t)row_lo * l.in;
        w.p += off; w.g += off; w.m += off; w.v += off;
        if (w_grad_bf16) w.g16 = w_grad_bf16 + off;
        w.sh = (char*)l.Wsh + (size_t)row_lo * l.inp * e->esz;
        w.rows = hi - row_lo;
        ids[a.n] = 2 * layer;
        a.t[a.n++] = w;
    }
    if (with_bias) {
        RtxAdamTensor b = full.t[1];
        if (b_grad_bf16) b.g16 = b_grad_bf16;
        ids[a.n] = 2 * layer + 1;
        a.t[a.n++] = b;
    }
    if (a.n == 0) return RTX_OK;
    fill_adam_scalars(e, step, a, 0, ids);
    TIMED("adam");
    RTX_TRY(rtx_launch_adam(a, e->bf16, st));
    e->shadows_valid = true;    // (whole again once the caller has all-gathered the compute copies of this step)
    return RTX_OK;
}

// the compute copy of layer `layer`'s weight matrix: [padded_rows][ld] elements of `elem_bytes` bytes, padded_rows a multiple
// of 128 -- the buffer a sharded-optimizer caller all-gathers in place (equal row blocks per rank)
int rtx_engine_shadow_region(rtx_engine* e, int32_t layer, void** base, int32_t* padded_rows, int32_t* ld, int32_t* elem_bytes)
{
    RTX_CHECK(e && layer >= 0 && layer < e->NL, RTX_EINVAL, "shadow_region: bad arguments");
    const Layer& l = e->L[layer];
    if (base) *base = l.Wsh;
    if (padded_rows) *padded_rows = l.outp;
    if (ld) *ld = l.inp;
    if (elem_bytes) *elem_bytes = (int32_t)e->esz;
    return RTX_OK;
}

int rtx_cast_f32_bf16(const float* src, uint16_t* dst, int64_t n, void* stream)
{
    RTX_CHECK(src && dst && n >= 0, RTX_EINVAL, "cast_f32_bf16: bad arguments");
    return rtx_launch_cast_f32_bf16(src, dst, (long)n, (hipStream_t)stream);
}

int rtx_engine_train_step_dp(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out, float* loss_accum, void* stream)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "train_step_dp: step count must be >= 1");
    RTX_CHECK(e->dp.on, RTX_ESTATE, "train_step_dp: rtx_engine_dp_attach() has not been called");
    return loss_grads_impl(e, batch, step, loss_out, loss_accum, nullptr, nullptr, (hipStream_t)stream, false, &e->dp);
}

int rtx_engine_train_step(rtx_engine* e, const rtx_batch* batch, const rtx_step* step, float* loss_out, float* loss_accum,
                          void* stream)
{
    RTX_TRY(check_ready(e, true));
    RTX_CHECK(step && step->step >= 1, RTX_EINVAL, "train_step: step count must be >= 1");
    if (e->bf16 && e->opt_fuse_adam)
        return loss_grads_impl(e, batch, step, loss_out, loss_accum, nullptr, nullptr, (hipStream_t)stream, true);
    // (float32, or bf16 with the fused optimizer switched off: Adam as launches of its own inside the same call)
    return loss_grads_impl(e, batch, step, loss_out, loss_accum, nullptr, nullptr, (hipStream_t)stream, false, nullptr, true);
}

// The reference's train_batch ends in `return loss.item()` (models.py:835): the host needs THIS step's loss.  Draining the stream
// for it also waits for the weight-gradient + Adam kernels behind the loss (a third of the step) and leaves the GPU idle while
// the host enqueues the next step.  With the mailbox on, the loss reduction of every training step also stores {loss, step count}
// into coherent host memory (system-scope release), and rtx_engine_wait_loss spins on the step count: the float it returns is
// this step's loss, the stream keeps running.
int rtx_engine_loss_mailbox(rtx_engine* e, int32_t enable)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    if (enable && !e->loss_mailbox) {
        void* p = nullptr;
        RTX_HIP(hipHostMalloc(&p, 64, hipHostMallocCoherent | hipHostMallocMapped));
        memset(p, 0, 64);
        e->loss_mailbox = (uint32_t*)p;
        e->loss_ticket = 0;
    } else if (!enable && e->loss_mailbox) {
        RTX_HIP(hipDeviceSynchronize());
        (void)hipHostFree(e->loss_mailbox);
        e->loss_mailbox = nullptr;
    }
    return RTX_OK;
}

int rtx_engine_wait_loss(rtx_engine* e, int32_t step, float* loss_host, double timeout_s)
{
    RTX_CHECK(e && loss_host, RTX_EINVAL, "wait_loss: NULL argument");
    RTX_CHECK(e->loss_mailbox, RTX_ESTATE, "wait_loss: rtx_engine_loss_mailbox(e, 1) has not been called");
    volatile uint32_t* mb = e->loss_mailbox;
    const auto t0 = std::chrono::steady_clock::now();
    // the word waited for is the engine's own ticket of the LAST reduction it enqueued (monotonic), not the caller's step count: a
    // count that restarts (a reloaded checkpoint, a new optimizer on the same engine) or repeats (step 0 twice) cannot match an old entry
    RTX_CHECK(e->loss_ticket != 0, RTX_ESTATE, "wait_loss: no training step has reported to the mailbox yet");
    const uint32_t want = e->loss_ticket;
    for (unsigned spin = 0;; ++spin) {
        if (__atomic_load_n(&mb[1], __ATOMIC_ACQUIRE) == want) break;
        if ((spin & 0x3ff) == 0x3ff) {
            const double el = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            RTX_CHECK(el < (timeout_s > 0 ? timeout_s : 60.0), RTX_EHIP, "wait_loss: step %d did not report its loss within %.1f s (mailbox holds ticket %u of %u, step %u)",
                      step, el, (unsigned)mb[1], (unsigned)want, (unsigned)mb[2]);
            if (el > 0.002) sched_yield();      // a long wait is a long kernel: stop burning the core
        }
    }
    RTX_CHECK(__atomic_load_n(&mb[2], __ATOMIC_RELAXED) == (uint32_t)step, RTX_ESTATE,
              "wait_loss: the mailbox holds the LAST step enqueued (step %u), not step %d -- steps are waited for in order", (unsigned)mb[2], step);
    const uint32_t bits = __atomic_load_n(&mb[0], __ATOMIC_RELAXED);
    memcpy(loss_host, &bits, 4);
    return RTX_OK;
}

// ABI 7: the batch (and the dropout stream: seed / offset / dropout_mask of `next_step`; its other fields are ignored) of the
// training step AFTER the next rtx_engine_train_step call.  That call then also gathers the announced batch -- on the engine's
// side stream, under its last weight-gradient + Adam launch, into a second batch image -- and the step that is then given exactly
// this batch (same csr / target_csr / row_ids pointers, batch size, seed, offset, mask) starts with the first-layer product.
// A hint: a step that gets anything else gathers for itself.  The row ids must not change between the announcement and that
// step.  NULL cancels.  (The reference has no counterpart: its sampler densifies on the host, samplers.py:99-100.)
int rtx_engine_set_next_batch(rtx_engine* e, const rtx_batch* next, const rtx_step* next_step)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    e->next.valid = false;
    if (!next) return RTX_OK;
    RTX_CHECK(next_step, RTX_EINVAL, "set_next_batch: next_step is NULL");
    e->next.b = *next;
    e->next.s = *next_step;
    e->next.valid = true;
    return RTX_OK;
}

// The body of evaluation.evaluate's loop (rectorch/evaluation.py:100-106) for EVERY batch of a held-out loader in one call: the host
// enqueues batch after batch without returning to Python in between (round 6: the selection kernel's 20 us left the host's ~100 us of
// per-batch Python and ctypes work as the limit of evaluate_device).
int rtx_engine_evaluate_topk(rtx_engine* e, const rtx_csr* train, const rtx_csr* heldout, const int32_t* row_ids, const int64_t* batch_offsets,
                             int32_t n_batches, const int32_t* ks_host, int32_t n_k, float* scores_scratch, double* ndcg, double* recall,
                             void* stream)
{
    RTX_TRY(check_ready(e, false));
    RTX_CHECK(train && heldout && row_ids && batch_offsets && ks_host && scores_scratch, RTX_EINVAL, "evaluate_topk: NULL argument");
    RTX_CHECK(n_batches >= 0 && n_k >= 1, RTX_EINVAL, "evaluate_topk: bad counts");
    RTX_CHECK(heldout->n_cols == e->I, RTX_EINVAL, "evaluate_topk: held-out matrix has %d columns, the network scores %d items", heldout->n_cols, e->I);
    hipStream_t st = (hipStream_t)stream;
    RTX_TRY(ensure_shadows(e, st));
    const int64_t total = batch_offsets[n_batches] - batch_offsets[0];
    int km = 0;
    for (int q = 0; q < n_k; ++q) km = std::max(km, (int)ks_host[q]);
    for (int32_t i = 0; i < n_batches; ++i) {
        const int64_t lo = batch_offsets[i], n = batch_offsets[i + 1] - lo;
        RTX_CHECK(n >= 1 && n <= e->cfg.max_batch, RTX_EINVAL, "evaluate_topk: batch %d has %lld rows (max_batch = %d)", i, (long long)n, e->cfg.max_batch);
        rtx_batch b = {};
        b.csr = train; b.row_ids = row_ids + lo; b.batch = (int32_t)n;
        RtxCsrView in = {}, tg = {};
        RTX_TRY(resolve_batch(e, &b, &in, &tg, st, 0));
        RTX_TRY(run_forward(e, &in, &in, (int)n, 0, nullptr, 0, 0, e->NL, scores_scratch, e->I, nullptr, nullptr, st));
        // (the train items' -inf: inside the selection kernel, which takes the train rows as an exclusion list)
        RtxCsrView hv = {heldout->indptr, heldout->indices, heldout->values, row_ids + lo};
        const int64_t col = lo - batch_offsets[0];
        RTX_TRY(rtx_launch_topk_metrics(scores_scratch, (long)e->I, (int)n, e->I, hv, ks_host, n_k, km, ndcg ? ndcg + col : nullptr,
                                        recall ? recall + col : nullptr, nullptr, st, (long)total, &in));
    }
    return RTX_OK;
}

}  // extern "C"
