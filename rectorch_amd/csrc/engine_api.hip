// engine_api.hip -- the parts of the C ABI that are not the step: measurement knobs (rtx_engine_set_option / get_option), the
// stand-alone operators (loss, norms, top-k metrics) and the instrumentation (event-timed sites, algorithmic cost of a step).
#include "engine_internal.h"

extern "C" {

// measurement knobs: one entry point instead of environment variables scattered over the kernels' launchers
int rtx_engine_set_option(rtx_engine* e, const char* key, int32_t value)
{
    RTX_CHECK(e && key, RTX_EINVAL, "set_option: NULL argument");
    const std::string k(key);
    if (k == "fuse_adam") e->opt_fuse_adam = value != 0;
    else if (k == "lse_fuse") e->opt_lse_fuse = value != 0;
    else if (k == "logits16") e->opt_logits16 = value != 0;
    else if (k == "hop_values") e->opt_hop_values = value != 0;
    else if (k == "hop_kernels") e->opt_hop_kernels = value != 0;
    else if (k == "hop_fold") e->opt_hop_fold = value != 0;
    else if (k == "dw_side_pad") e->opt_dw_side_pad = value > 0 ? value : 0;
    else if (k == "small_kw") rtx_small_set_kw(value);         // (process-wide: K split of small_layers.hip's kernels over waves)
    else if (k == "small_waves") rtx_small_set_waves(value);   // (process-wide: a launch-shape knob of small_layers.hip)
    else if (k == "timing_calibrate") e->opt_timing_calibrate = value != 0;
    else if (k == "hop_wrap") {
        RTX_CHECK(value >= 2, RTX_EINVAL, "set_option: hop_wrap must be >= 2");
        e->hop_wrap = (uint32_t)value;
    }
    else if (k == "f32_dw_split") e->opt_f32_dw_split = value != 0;
    else if (k == "f32_tail_split") e->opt_f32_tail_split = value != 0;
    else if (k == "f32_adam_overlap") e->opt_f32_adam_overlap = value != 0;
    else if (k == "splitk_fwd") e->opt_splitk_fwd = value;
    else if (k == "splitk_bwd") e->opt_splitk_bwd = value;
    else if (k == "gather_scatter") e->opt_gather_scatter = value != 0;
    else if (k == "dp_shard_min_elems") {
        RTX_CHECK(!e->dp.on && value >= 1, RTX_ESTATE, "set_option: dp_shard_min_elems (>= 1) must be set before rtx_engine_dp_attach");
        e->opt_dp_shard_min_elems = value;
    }
    else if (k == "dp_one_comm") {
        RTX_CHECK(!e->dp.on, RTX_ESTATE, "set_option: dp_one_comm must be set before rtx_engine_dp_attach");
        e->opt_dp_one_comm = value != 0;
    }
    else if (k == "prefetch") e->opt_prefetch = value != 0;
    else if (k == "two_stream") e->opt_two_stream = value != 0;
    else if (k == "side_low_prio") {
        RTX_CHECK(e->side_cache.empty(), RTX_ESTATE, "set_option: side_low_prio must be set before the first training step");
        e->opt_side_low_prio = value != 0;
    }
    else if (k == "nt_regstage") e->opt_nt_regstage = value != 0;
    else if (k == "in_on_main") e->opt_in_on_main = value != 0;
    else if (k == "sparse_in") e->opt_sparse_in = value != 0;
    else if (k == "small_fwd") e->opt_small_fwd = value != 0;
    else if (k == "small_bwd") e->opt_small_bwd = value != 0;
    else if (k == "big_batch_tiles") e->opt_big_batch_tiles = value != 0;
    else if (k == "dw_cfg") {
        RTX_CHECK(value >= RTX_DW_64x128 && value <= RTX_DW_128x128_K32, RTX_EINVAL, "set_option: dw_cfg must be 0..8");
        e->opt_dw_cfg = value;
        e->opt_dw_cfg_set = 1;
    } else if (k == "splitk") {
        RTX_CHECK(value >= 0, RTX_EINVAL, "set_option: splitk must be >= 0");
        // the scratch was sized for the automatic choice: only accept factors it can hold
        const int old = e->cfg.splitk;
        e->cfg.splitk = value;
        size_t need = 0;
        for (int li = 0; li < e->NL; ++li) {
            if (li < e->NL - 1) need = std::max(need, plan_cacc_elems(e, e->L[li].outp, e->L[li].inp));
            if (li > 0) need = std::max(need, plan_cacc_elems(e, e->L[li].inp, e->L[li].outp));
        }
        if (need > e->cacc_elems) {
            e->cfg.splitk = old;
            rtx_set_error("set_option: split factor %d needs %zu scratch floats, the engine holds %zu", value, need, e->cacc_elems);
            return RTX_EINVAL;
        }
    } else {
        rtx_set_error("set_option: unknown key '%s' (fuse_adam, lse_fuse, two_stream, side_low_prio, nt_regstage, in_on_main, sparse_in, small_fwd, small_bwd, big_batch_tiles, dw_cfg, splitk)", key);
        return RTX_EINVAL;
    }
    return RTX_OK;
}

int rtx_engine_get_option(const rtx_engine* e, const char* key, int32_t* value)
{
    RTX_CHECK(e && key && value, RTX_EINVAL, "get_option: NULL argument");
    const std::string k(key);
    if (k == "fuse_adam") *value = e->opt_fuse_adam;
    else if (k == "lse_fuse") *value = e->opt_lse_fuse;
    else if (k == "logits16") *value = e->opt_logits16;
    else if (k == "hop_values") *value = e->opt_hop_values;
    else if (k == "hop_kernels") *value = e->opt_hop_kernels;
    else if (k == "hop_fold") *value = e->opt_hop_fold;
    else if (k == "gather_scatter") *value = e->opt_gather_scatter;
    else if (k == "dp_shard_min_elems") *value = e->opt_dp_shard_min_elems;
    else if (k == "dp_bytes_all_reduce") *value = (int32_t)std::min<int64_t>(e->dp.st_all_reduce, INT32_MAX);       // per rank, last step
    else if (k == "dp_bytes_reduce_scatter") *value = (int32_t)std::min<int64_t>(e->dp.st_reduce_scatter, INT32_MAX);
    else if (k == "dp_bytes_all_gather") *value = (int32_t)std::min<int64_t>(e->dp.st_all_gather, INT32_MAX);
    else if (k == "dp_collectives") *value = e->dp.st_collectives;
    else if (k == "dp_one_comm") *value = e->opt_dp_one_comm;
    else if (k == "prefetch") *value = e->opt_prefetch;
    else if (k == "join_folds") *value = e->st_join_folds;             // deferred joins that rode on a first-layer product
    else if (k == "prefetch_hits") *value = e->st_prefetch_hits;       // steps that started from a prefetched batch image
    else if (k == "prefetch_issued") *value = e->st_prefetch_issued;
    else if (k == "dp_two_comms") *value = e->dp.on && e->dp.two_comms;   // bucket A's collectives have a communicator of their own
    else if (k == "two_stream") *value = e->opt_two_stream;
    else if (k == "side_low_prio") *value = e->opt_side_low_prio;
    else if (k == "nt_regstage") *value = e->opt_nt_regstage;
    else if (k == "in_on_main") *value = e->opt_in_on_main;
    else if (k == "sparse_in") *value = e->opt_sparse_in;
    else if (k == "small_fwd") *value = e->opt_small_fwd;
    else if (k == "small_bwd") *value = e->opt_small_bwd;
    else if (k == "big_batch_tiles") *value = e->opt_big_batch_tiles;
    else if (k == "dw_cfg") *value = e->opt_dw_cfg;
    else if (k == "splitk") *value = e->cfg.splitk;
    else if (k == "last_sparse_in") *value = e->last_sparse_in;
    else if (k == "side_concurrent") *value = e->side_concurrent;   // 1: the step's second stream was seen to run beside the caller's   // 1: the last forward pass ran the first layer as the sparse product
    else {
        rtx_set_error("get_option: unknown key '%s'", key);
        return RTX_EINVAL;
    }
    return RTX_OK;
}

int rtx_multinomial_loss(const float* recon, const float* x, int32_t batch, int32_t n_items, const float* mu, const float* logvar,
                         int32_t latent, float beta, float* loss_out, void* stream)
{
    RTX_CHECK(recon && x && loss_out && batch >= 1 && n_items >= 1, RTX_EINVAL, "multinomial_loss: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* row_loss = nullptr;
    RTX_HIP(hipMallocAsync((void**)&row_loss, sizeof(float) * batch, st));
    int rc = rtx_launch_dense_loss(recon, x, batch, n_items, (mu && logvar) ? mu : nullptr, logvar, latent, beta, 1.f / (float)batch,
                                   row_loss, st);
    if (!rc) rc = rtx_launch_reduce_loss(row_loss, batch, 0.f, nullptr, 0, loss_out, nullptr, st);
    (void)hipFreeAsync(row_loss, st);
    return rc;
}

int rtx_sum_l2_norms(const float* const* tensors, const int64_t* sizes, int32_t n, float* out, void* stream)
{
    RTX_CHECK(tensors && sizes && out && n >= 1 && n <= RTX_MAX_TENSORS, RTX_EINVAL, "sum_l2_norms: bad arguments");
    hipStream_t st = (hipStream_t)stream;
    float* sumsq = nullptr;
    RTX_HIP(hipMallocAsync((void**)&sumsq, sizeof(float) * n, st));
    std::vector<long> sz(sizes, sizes + n);
    int rc = rtx_launch_sumsq(tensors, sz.data(), n, sumsq, st);
    if (!rc) rc = rtx_launch_reduce_loss(nullptr, 0, 1.f, sumsq, n, out, nullptr, st);
    (void)hipFreeAsync(sumsq, st);
    return rc;
}

int rtx_topk_metrics(const float* scores, int64_t ld, int32_t batch, int32_t n_items, const rtx_csr* heldout,
                     const int32_t* row_ids, const int32_t* ks_host, int32_t n_k, double* ndcg, double* recall,
                     int32_t* topk_idx, int32_t kmax, void* stream)
{
    RTX_CHECK(scores && heldout && ks_host, RTX_EINVAL, "topk_metrics: NULL argument");
    RTX_CHECK(heldout->n_cols == n_items, RTX_EINVAL, "topk_metrics: held-out matrix has %d columns, scores have %d", heldout->n_cols, n_items);
    RTX_CHECK(row_ids || batch <= heldout->n_rows, RTX_EINVAL, "topk_metrics: batch larger than the held-out matrix");
    int km = kmax;
    for (int q = 0; q < n_k; ++q) km = std::max(km, (int)ks_host[q]);
    RtxCsrView v = {heldout->indptr, heldout->indices, heldout->values, row_ids};
    return rtx_launch_topk_metrics(scores, (long)ld, batch, n_items, v, ks_host, n_k, km, ndcg, recall, topk_idx, (hipStream_t)stream);
}

// ---- instrumentation -------------------------------------------------------------------------------
int rtx_engine_set_timing(rtx_engine* e, const char* site, int32_t enable)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    if (!site) {
        e->timing_all = enable != 0;
        if (!enable) e->timing_sites.clear();
    } else if (enable) {
        e->timing_sites[site] = enable;
        e->timing_seen[site] = 0;
    } else {
        e->timing_sites.erase(site);
    }
    return RTX_OK;
}

int rtx_engine_get_timings(rtx_engine* e, int32_t cap, char (*names)[48], float* total_ms, int32_t* launches, int32_t* n_out)
{
    RTX_CHECK(e && n_out, RTX_EINVAL, "get_timings: NULL argument");
    RTX_HIP(hipDeviceSynchronize());
    int n = 0;
    for (auto& kv : e->sites) {
        TimingSite& s = kv.second;
        for (auto& pr : s.pending) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) {
                s.total_ms += ms;
                s.launches += 1;
            }
            e->event_pool.push_back(pr.first);
            e->event_pool.push_back(pr.second);
        }
        s.pending.clear();
        if (n < cap && names && total_ms && launches) {
            strncpy(names[n], kv.first.c_str(), 47);
            names[n][47] = 0;
            total_ms[n] = (float)s.total_ms;
            launches[n] = s.launches;
            ++n;
        }
        s.total_ms = 0;
        s.launches = 0;
    }
    *n_out = n;
    return RTX_OK;
}

int rtx_engine_step_cost(const rtx_engine* e, int32_t batch, double* hbm_bytes, double* flops)
{
    RTX_CHECK(e, RTX_EINVAL, "engine is NULL");
    // bytes = 38*P + 12*B*I  (fp32 master params + Adam state, logits written once and read twice)
    //               flops = forward 2*sum(in*out) + weight grads 2*sum(in*out) + data grads 2*sum_{l>0}(in*out), per user
    double P = 0, f_all = 0, f_rest = 0;
    for (int li = 0; li < e->NL; ++li) {
        const double w = (double)e->L[li].in * e->L[li].out;
        P += w + e->L[li].out;
        f_all += w;
        if (li > 0) f_rest += w;
    }
    if (hbm_bytes) *hbm_bytes = 38.0 * P + 12.0 * (double)batch * e->I;
    if (flops) *flops = (double)batch * 2.0 * (2.0 * f_all + f_rest);
    return RTX_OK;
}

}  // extern "C"
