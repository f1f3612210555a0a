// engine_dp.hip -- the data-parallel plan of the engine (rtx_engine_dp_attach): the exchange buffer's layout, which matrices are
// sharded, and the transports the step's collectives go through (the engine's own RCCL communicators, caller-supplied operations,
// same-size device copies for one-GPU emulation).  The step that USES the plan is rtx_engine_train_step_dp -> loss_grads_impl in
// engine.hip.  (The reference has no collective code: rectorch/models.py:409-419 is a single-device loop.)
#include "engine_internal.h"

extern "C" {

// ---- data parallel: attach / step --------------------------------------------------------------------------------
static int dp_rccl_all_reduce(void* c, void* buf, int64_t n, int32_t dt, void* st) { return rtx_comm_allreduce((rtx_comm*)c, buf, n, dt, st); }
static int dp_rccl_reduce_scatter(void* c, void* buf, int64_t n, int32_t dt, void* st) { return rtx_comm_reduce_scatter((rtx_comm*)c, buf, n, dt, st); }
static int dp_rccl_all_gather(void* c, void* buf, int64_t bytes, void* st) { return rtx_comm_allgather((rtx_comm*)c, buf, bytes, st); }
static int dp_rccl_group_start(void* c) { return rtx_comm_group_start((rtx_comm*)c); }
static int dp_rccl_group_end(void* c) { return rtx_comm_group_end((rtx_comm*)c); }

// emulate: the bytes one rank of `world` sends + receives in a ring collective, as device copies through a scratch buffer
// (reduce-scatter / all-gather: (world - 1) / world of the buffer read and written once; all-reduce: twice -- there and back,
// numerically a no-op).  One launch per collective, or per group of collectives, like RCCL's own kernels.
struct EmuCopyArgs {
    struct { const uint4* src; uint4* dst; unsigned long n16; int back; } p[8];
    int n;
};
__global__ __launch_bounds__(256) void k_emu_copy(const EmuCopyArgs a)
{
    for (int k = 0; k < a.n; ++k) {
        const uint4* __restrict__ src = a.p[k].src;
        uint4* __restrict__ dst = a.p[k].dst;
        for (unsigned long i = (unsigned long)blockIdx.x * 256 + threadIdx.x; i < a.p[k].n16; i += (unsigned long)gridDim.x * 256) {
            const uint4 v = src[i];
            dst[i] = v;
            if (a.p[k].back) ((uint4*)src)[i] = v;   // the all-gather half of an all-reduce writes the block back
        }
    }
}
static int dp_emu_flush(DpState* d)
{
    if (d->emu_n == 0) return RTX_OK;
    EmuCopyArgs a = {};
    size_t used = 0;
    for (int k = 0; k < d->emu_n; ++k) {
        const size_t w = (size_t)d->cfg.world, bytes = d->emu_q[k].bytes;
        size_t s = (bytes / w * (w - 1)) & ~(size_t)15;
        s = std::min(s, d->emu_bytes - used);
        if (s == 0) continue;
        a.p[a.n].src = (const uint4*)((char*)d->emu_q[k].buf + ((bytes - s) & ~(size_t)15));   // "the other ranks' blocks"
        a.p[a.n].dst = (uint4*)((char*)d->emu_scratch + used);
        a.p[a.n].n16 = s / 16;
        a.p[a.n].back = d->emu_q[k].back;
        used += s;
        ++a.n;
    }
    d->emu_n = 0;
    if (a.n == 0) return RTX_OK;
    hipLaunchKernelGGL(k_emu_copy, dim3(2048), dim3(256), 0, d->emu_stream, a);
    RTX_HIP(hipGetLastError());
    return RTX_OK;
}
static int dp_emu_move(DpState* d, void* buf, size_t bytes, bool back, hipStream_t st)
{
    if (d->emu_n == 8) RTX_TRY(dp_emu_flush(d));
    d->emu_stream = st;
    d->emu_q[d->emu_n++] = DpState::EmuPiece{buf, bytes, back ? 1 : 0};
    return d->emu_grouped ? RTX_OK : dp_emu_flush(d);
}
static int dp_emu_group_start(void* c) { ((DpState*)c)->emu_grouped = 1; return RTX_OK; }
static int dp_emu_group_end(void* c) { ((DpState*)c)->emu_grouped = 0; return dp_emu_flush((DpState*)c); }
static int dp_emu_all_reduce(void* c, void* buf, int64_t n, int32_t dt, void* st)
{
    return dp_emu_move((DpState*)c, buf, (size_t)n * (dt == RTX_BF16 ? 2 : 4), true, (hipStream_t)st);
}
static int dp_emu_reduce_scatter(void* c, void* buf, int64_t n, int32_t dt, void* st)
{
    return dp_emu_move((DpState*)c, buf, (size_t)n * (dt == RTX_BF16 ? 2 : 4), false, (hipStream_t)st);
}
static int dp_emu_all_gather(void* c, void* buf, int64_t bytes, void* st) { return dp_emu_move((DpState*)c, buf, (size_t)bytes, false, (hipStream_t)st); }

}  // extern "C"
void dp_release(rtx_engine* e)
{
    DpState& d = e->dp;
    for (void* p : {d.xg, d.emu_scratch})
        if (p) {
            auto it = std::find(e->allocs.begin(), e->allocs.end(), p);
            if (it != e->allocs.end()) e->allocs.erase(it);
            (void)hipFree(p);
        }
    d = DpState();
}

extern "C" {

int rtx_engine_dp_attach(rtx_engine* e, const rtx_dp_cfg* cfg)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    RTX_HIP(hipDeviceSynchronize());
    e->join_pending = e->join_fold = false;   // (every stream has drained)
    dp_release(e);
    if (!cfg) return RTX_OK;
    RTX_CHECK(cfg->world >= 1 && cfg->rank >= 0 && cfg->rank < cfg->world, RTX_EINVAL, "dp_attach: rank %d of %d", cfg->rank, cfg->world);
    RTX_CHECK(cfg->comm_dtype == RTX_FP32 || cfg->comm_dtype == RTX_BF16, RTX_EINVAL, "dp_attach: comm_dtype must be RTX_FP32 or RTX_BF16");
    RTX_CHECK((cfg->emulate != 0) + (cfg->comm != nullptr) + (cfg->ops != nullptr) == 1, RTX_EINVAL,
              "dp_attach: give exactly one of comm (RCCL), ops (caller's collectives) or emulate");
    DpState& d = e->dp;
    d.cfg = *cfg;
    if (cfg->emulate) {
        d.ops = rtx_dp_ops{dp_emu_all_reduce, dp_emu_reduce_scatter, dp_emu_all_gather, dp_emu_group_start, dp_emu_group_end, &e->dp};
    } else if (cfg->comm) {
        int32_t r = -1, w = -1;
        RTX_TRY(rtx_comm_rank(cfg->comm, &r, &w));
        RTX_CHECK(r == cfg->rank && w == cfg->world, RTX_EINVAL, "dp_attach: the communicator is rank %d of %d, the plan says %d of %d", r, w, cfg->rank, cfg->world);
        d.ops = rtx_dp_ops{dp_rccl_all_reduce, dp_rccl_reduce_scatter, dp_rccl_all_gather, dp_rccl_group_start, dp_rccl_group_end, cfg->comm};
    } else {
        RTX_CHECK(cfg->ops->all_reduce && cfg->ops->reduce_scatter && cfg->ops->all_gather, RTX_EINVAL, "dp_attach: ops needs all_reduce, reduce_scatter and all_gather");
        d.ops = *cfg->ops;
    }
    d.cfg.ops = nullptr;
    // bucket A's table: a second communicator / function table when the plan brings one, else the same as bucket B's
    d.ops_side = d.ops;
    d.two_comms = false;
    if (!cfg->emulate && cfg->comm && cfg->comm_side && !e->opt_dp_one_comm) {
        int32_t r = -1, w = -1;
        RTX_TRY(rtx_comm_rank(cfg->comm_side, &r, &w));
        RTX_CHECK(r == cfg->rank && w == cfg->world, RTX_EINVAL, "dp_attach: comm_side is rank %d of %d, the plan says %d of %d", r, w, cfg->rank, cfg->world);
        RTX_CHECK(cfg->comm_side != cfg->comm, RTX_EINVAL, "dp_attach: comm_side must be a communicator of its own (or NULL)");
        d.ops_side.ctx = cfg->comm_side;
        d.two_comms = true;
    } else if (!cfg->emulate && cfg->ops && cfg->ops_side && !e->opt_dp_one_comm) {
        RTX_CHECK(cfg->ops_side->all_reduce && cfg->ops_side->reduce_scatter && cfg->ops_side->all_gather, RTX_EINVAL,
                  "dp_attach: ops_side needs all_reduce, reduce_scatter and all_gather");
        d.ops_side = *cfg->ops_side;
        d.two_comms = true;
    }
    d.cfg.ops_side = nullptr;
    d.xesz = cfg->comm_dtype == RTX_BF16 ? 2 : 4;
    size_t biggest = 0;
    for (int li = 0; li < e->NL; ++li) {
        const Layer& l = e->L[li];
        // a hidden layer that keeps a transposed compute copy (WshT) is never sharded: that copy is a column-block layout
        const long min_elems = cfg->shard_min_elems > 0 ? (long)cfg->shard_min_elems : (long)e->opt_dp_shard_min_elems;
        d.shard[li] = cfg->sharded && (long)l.out * l.in >= min_elems && !l.WshT && l.outp % cfg->world == 0;
        biggest = std::max(biggest, (size_t)l.outp * l.inp * std::max(e->esz, d.xesz));
    }
    int order[2 * 2 * RTX_MAX_LAYERS];
    const int n_order = dp_layout_order(e, order);
    size_t off = 0;
    for (int q = 0; q < n_order; ++q) {
        d.xoff[order[q]] = off;
        off += (dp_region_elems(e, d, order[q]) + 63) / 64 * 64;
    }
    d.xbytes = off * d.xesz;
    RTX_TRY(dev_alloc(e, &d.xg, d.xbytes));
    if (cfg->emulate) {
        d.emu_bytes = 2 * biggest;
        RTX_TRY(dev_alloc(e, &d.emu_scratch, d.emu_bytes, false));
        // rows no rank updates here keep their weights in BOTH compute copies (the step alternates between them)
        RTX_TRY(ensure_shadows(e, nullptr));
        for (auto& l : e->L)
            if (l.Wsh_alt) RTX_HIP(hipMemcpy(l.Wsh_alt, l.Wsh, (size_t)l.outp * l.inp * e->esz, hipMemcpyDeviceToDevice));
    }
    d.on = true;
    return RTX_OK;
}

int rtx_engine_dp_owned_rows(const rtx_engine* e, int32_t layer, int32_t* row_lo, int32_t* row_hi, int32_t* sharded_out)
{
    RTX_CHECK(e && layer >= 0 && layer < e->NL, RTX_EINVAL, "dp_owned_rows: bad arguments");
    const Layer& l = e->L[layer];
    int lo = 0, hi = l.out, sh = 0;
    if (e->dp.on && e->dp.shard[layer]) {
        const int per = l.outp / e->dp.cfg.world;
        lo = std::min(e->dp.cfg.rank * per, l.out);
        hi = std::min((e->dp.cfg.rank + 1) * per, l.out);
        sh = 1;
    }
    if (row_lo) *row_lo = lo;
    if (row_hi) *row_hi = hi;
    if (sharded_out) *sharded_out = sh;
    return RTX_OK;
}

}  // extern "C"
