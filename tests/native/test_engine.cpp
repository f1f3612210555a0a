// Native (no Python) parity check of librectorch_hip against the C oracle (oracle/mvae_oracle.c) through
// the public C ABI, plus a per-kernel timing table at the ml-20m shape.  Test infrastructure: this is
// the only place where the oracle and the HIP path meet in one process.
//
//   test_engine            -> parity cases (seconds)
//   test_engine perf [B]   -> parity + ml-20m shape timing (MultiVAE [20108,600,200], batch B=500)
#include "../../include/rectorch_hip.h"
#include "../../oracle/mvae_oracle.h"
#include "../../rectorch_amd/csrc/rtx_common.h"  // host replica of the Philox decision (test only)

#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#define CK(x)                                                                                  \
    do {                                                                                       \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess) {                                                                \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);     \
            exit(2);                                                                           \
        }                                                                                      \
    } while (0)
#define RT(x)                                                                   \
    do {                                                                        \
        int rc_ = (x);                                                          \
        if (rc_ != 0) {                                                         \
            printf("RTX error %d (%s) at %s:%d\n", rc_, rtx_last_error(), __FILE__, __LINE__); \
            exit(3);                                                            \
        }                                                                       \
    } while (0)

static uint64_t rng_s = 0x9E3779B97F4A7C15ull;
static uint32_t rnd()
{
    rng_s ^= rng_s << 13; rng_s ^= rng_s >> 7; rng_s ^= rng_s << 17;
    return (uint32_t)(rng_s >> 32);
}
static float frand() { return (rnd() >> 8) * (1.0f / 16777216.0f); }
static float nrand()
{
    float u1 = frand() + 1e-7f, u2 = frand();
    return sqrtf(-2.f * logf(u1)) * cosf(6.2831853f * u2);
}

struct Net {
    std::vector<int> enc, dec;
    int variant;
    float p;
    std::vector<std::vector<float>> params;  // host
    std::vector<int> rows, cols;
};

static Net make_net(std::vector<int> enc, std::vector<int> dec, int variant, float p, float bias_std)
{
    Net n;
    n.enc = enc; n.dec = dec; n.variant = variant; n.p = p;
    orc_cfg c = {};
    c.n_enc = (int)enc.size() - 1; c.n_dec = (int)dec.size() - 1;
    for (size_t i = 0; i < enc.size(); ++i) c.enc_dims[i] = enc[i];
    for (size_t i = 0; i < dec.size(); ++i) c.dec_dims[i] = dec[i];
    c.variant = variant;
    int nt = orc_n_tensors(&c);
    for (int t = 0; t < nt; ++t) {
        int r, cc;
        orc_tensor_shape(&c, t, &r, &cc);
        n.rows.push_back(r); n.cols.push_back(cc);
        std::vector<float> w((size_t)r * cc);
        if (t & 1) for (auto& v : w) v = nrand() * bias_std;
        else { float a = sqrtf(6.f / (r + cc)); for (auto& v : w) v = (2.f * frand() - 1.f) * a; }
        n.params.push_back(w);
    }
    return n;
}

static orc_cfg ocfg(const Net& n)
{
    orc_cfg c = {};
    c.n_enc = (int)n.enc.size() - 1; c.n_dec = (int)n.dec.size() - 1;
    for (size_t i = 0; i < n.enc.size(); ++i) c.enc_dims[i] = n.enc[i];
    for (size_t i = 0; i < n.dec.size(); ++i) c.dec_dims[i] = n.dec[i];
    c.variant = n.variant; c.dropout_p = n.p;
    return c;
}

static rtx_cfg rcfg(const Net& n, int numerics, int max_batch)
{
    rtx_cfg c = {};
    c.n_enc = (int)n.enc.size() - 1; c.n_dec = (int)n.dec.size() - 1;
    for (size_t i = 0; i < n.enc.size(); ++i) c.enc_dims[i] = n.enc[i];
    for (size_t i = 0; i < n.dec.size(); ++i) c.dec_dims[i] = n.dec[i];
    c.variant = n.variant; c.numerics = numerics; c.dropout_p = n.p; c.max_batch = max_batch; c.splitk = 0;
    return c;
}

struct DevTensors {
    std::vector<float*> p, g, m, v;
    void alloc(const Net& n)
    {
        for (size_t t = 0; t < n.params.size(); ++t) {
            float *a, *b, *c, *d;
            size_t bytes = n.params[t].size() * sizeof(float);
            CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes)); CK(hipMalloc(&c, bytes)); CK(hipMalloc(&d, bytes));
            CK(hipMemcpy(a, n.params[t].data(), bytes, hipMemcpyHostToDevice));
            CK(hipMemset(b, 0xff, bytes));  // poison the gradients: every entry must be written
            CK(hipMemset(c, 0, bytes)); CK(hipMemset(d, 0, bytes));
            p.push_back(a); g.push_back(b); m.push_back(c); v.push_back(d);
        }
    }
    void release()
    {
        for (auto* x : p) hipFree(x);
        for (auto* x : g) hipFree(x);
        for (auto* x : m) hipFree(x);
        for (auto* x : v) hipFree(x);
    }
};

struct Csr {
    std::vector<int64_t> indptr;
    std::vector<int32_t> indices;
    std::vector<float> values;
    int rows, cols;
    std::vector<float> dense(const std::vector<int>& ids) const
    {
        std::vector<float> d(ids.size() * (size_t)cols, 0.f);
        for (size_t b = 0; b < ids.size(); ++b)
            for (int64_t k = indptr[ids[b]]; k < indptr[ids[b] + 1]; ++k) d[b * (size_t)cols + indices[k]] = values.empty() ? 1.f : values[k];
        return d;
    }
};

static Csr make_csr(int rows, int cols, float density, bool weighted, int empty_row)
{
    Csr c;
    c.rows = rows; c.cols = cols;
    c.indptr.push_back(0);
    for (int r = 0; r < rows; ++r) {
        if (r != empty_row)
            for (int j = 0; j < cols; ++j)
                if (frand() < density) {
                    c.indices.push_back(j);
                    if (weighted) c.values.push_back((float)(1 + rnd() % 3));
                }
        c.indptr.push_back((int64_t)c.indices.size());
    }
    return c;
}

static double rel_err(const float* got, const float* ref, size_t n, double* max_ref_out = nullptr)
{
    double me = 0, mr = 0;
    for (size_t i = 0; i < n; ++i) {
        double e = fabs((double)got[i] - (double)ref[i]);
        if (!(e == e)) return 1e30;  // NaN
        me = std::max(me, e);
        mr = std::max(mr, fabs((double)ref[i]));
    }
    if (max_ref_out) *max_ref_out = mr;
    return me / std::max(mr, 1e-30);
}

static std::vector<float> d2h(const float* d, size_t n)
{
    std::vector<float> h(n);
    CK(hipMemcpy(h.data(), d, n * sizeof(float), hipMemcpyDeviceToHost));
    return h;
}

static int g_fail = 0;
static int g_opt_fuse = 1, g_opt_dw_cfg = 0, g_opt_lse = 1, g_opt_two = 1, g_opt_ntreg = 1, g_opt_lowprio = 1, g_opt_inmain = 1, g_opt_sparse = 1;   // the shipped defaults   // rtx_engine_set_option values applied to every engine a case creates
static void apply_options(rtx_engine* eng)
{
    rtx_engine_set_option(eng, "fuse_adam", g_opt_fuse);
    rtx_engine_set_option(eng, "dw_cfg", g_opt_dw_cfg);
    rtx_engine_set_option(eng, "lse_fuse", g_opt_lse);
    rtx_engine_set_option(eng, "two_stream", g_opt_two);
    rtx_engine_set_option(eng, "nt_regstage", g_opt_ntreg);
    rtx_engine_set_option(eng, "side_low_prio", g_opt_lowprio);
    rtx_engine_set_option(eng, "in_on_main", g_opt_inmain);
    rtx_engine_set_option(eng, "sparse_in", g_opt_sparse);
}
static void check(const char* what, double err, double tol)
{
    const bool ok = err <= tol;
    printf("    %-28s rel_err=%.3e (tol %.1e) %s\n", what, err, tol, ok ? "ok" : "FAIL");
    if (!ok) ++g_fail;
}

// one full parity scenario: eval forward, predict, 2 training steps with injected RNG
static void parity_case(const char* name, Net net, int numerics, int B, int n_users, float density, bool weighted, bool use_te,
                        bool dense_api, float beta, float lam)
{
    printf("[%s] %s enc=", name, numerics ? "bf16" : "fp32");
    for (int d : net.enc) printf("%d,", d);
    printf(" dec=");
    for (int d : net.dec) printf("%d,", d);
    printf(" B=%d %s%s%s\n", B, net.variant ? "DAE" : "VAE", use_te ? " te" : "", dense_api ? " dense-api" : "");
    const int I = net.enc[0], Z = net.enc.back();
    const bool vae = net.variant == ORC_VAE;
    const double tol_fwd = numerics ? 3e-2 : 1e-5, tol_grad = numerics ? 1e-1 : 2e-4, tol_loss = numerics ? 2e-2 : 1e-5;
    Csr tr = make_csr(n_users, I, density, weighted, 1);
    Csr te = make_csr(n_users, I, density * 0.5f, false, -1);
    te.indptr = te.indptr;  // same row count
    std::vector<int> ids(B);
    for (int b = 0; b < B; ++b) ids[b] = (b * 7 + 1) % n_users;  // includes the empty row 1
    std::vector<float> x = tr.dense(ids), gt = te.dense(ids);
    std::vector<uint8_t> mask((size_t)B * I);
    for (auto& m : mask) m = frand() >= net.p;
    std::vector<float> eps((size_t)B * Z);
    for (auto& v : eps) v = nrand();

    orc_cfg oc = ocfg(net);
    std::vector<const float*> pp;
    for (auto& w : net.params) pp.push_back(w.data());
    const int nt = (int)net.params.size();

    // ---- engine setup
    rtx_cfg rc = rcfg(net, numerics, B + 3);
    rtx_engine* eng = nullptr;
    RT(rtx_engine_create(&rc, &eng));
    apply_options(eng);
    DevTensors dt;
    dt.alloc(net);
    RT(rtx_engine_bind(eng, dt.p.data(), dt.g.data(), dt.m.data(), dt.v.data()));
    rtx_csr *ctr = nullptr, *cte = nullptr;
    RT(rtx_csr_upload(tr.indptr.data(), tr.indices.data(), tr.values.empty() ? nullptr : tr.values.data(), tr.rows, I, &ctr));
    RT(rtx_csr_upload(te.indptr.data(), te.indices.data(), nullptr, te.rows, I, &cte));
    int32_t* d_ids; uint8_t* d_mask; float *d_eps, *d_logits, *d_mu, *d_lv, *d_loss, *d_x, *d_gt;
    CK(hipMalloc(&d_ids, B * 4)); CK(hipMemcpy(d_ids, ids.data(), B * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_mask, mask.size())); CK(hipMemcpy(d_mask, mask.data(), mask.size(), hipMemcpyHostToDevice));
    CK(hipMalloc(&d_eps, eps.size() * 4)); CK(hipMemcpy(d_eps, eps.data(), eps.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_logits, (size_t)B * I * 4)); CK(hipMalloc(&d_mu, (size_t)B * Z * 4)); CK(hipMalloc(&d_lv, (size_t)B * Z * 4));
    CK(hipMalloc(&d_loss, 8)); CK(hipMemset(d_loss, 0, 8));
    CK(hipMalloc(&d_x, x.size() * 4)); CK(hipMemcpy(d_x, x.data(), x.size() * 4, hipMemcpyHostToDevice));
    CK(hipMalloc(&d_gt, gt.size() * 4)); CK(hipMemcpy(d_gt, gt.data(), gt.size() * 4, hipMemcpyHostToDevice));

    rtx_batch bt = {};
    bt.batch = B;
    if (dense_api) { bt.x_dense = d_x; if (use_te) bt.target_dense = d_gt; }
    else { bt.csr = ctr; bt.row_ids = d_ids; if (use_te) bt.target_csr = cte; }

    // ---- eval forward + predict
    std::vector<float> o_logits((size_t)B * I), o_mu((size_t)B * Z), o_lv((size_t)B * Z);
    orc_forward_backward(&oc, pp.data(), x.data(), nullptr, B, 0, nullptr, nullptr, 0.f, 0.f, 1.f / B, o_logits.data(), o_mu.data(),
                         o_lv.data(), nullptr, nullptr);
    CK(hipMemset(d_logits, 0xff, (size_t)B * I * 4));
    RT(rtx_engine_forward(eng, &bt, 0, nullptr, 0, d_logits, vae ? d_mu : nullptr, vae ? d_lv : nullptr, nullptr));
    CK(hipDeviceSynchronize());
    auto g_logits = d2h(d_logits, (size_t)B * I);
    check("eval logits", rel_err(g_logits.data(), o_logits.data(), g_logits.size()), tol_fwd);
    if (vae) {
        auto g_mu = d2h(d_mu, (size_t)B * Z), g_lv = d2h(d_lv, (size_t)B * Z);
        check("eval mu", rel_err(g_mu.data(), o_mu.data(), g_mu.size()), tol_fwd);
        check("eval logvar", rel_err(g_lv.data(), o_lv.data(), g_lv.size()), tol_fwd);
    }
    RT(rtx_engine_forward(eng, &bt, 0, nullptr, 1, d_logits, nullptr, nullptr, nullptr));
    CK(hipDeviceSynchronize());
    {
        auto pr = d2h(d_logits, (size_t)B * I);
        long bad = 0;
        for (size_t k = 0; k < pr.size(); ++k) {
            const bool inf = std::isinf(pr[k]) && pr[k] < 0;
            if (inf != (x[k] != 0.f)) ++bad;
        }
        check("predict -inf mask", (double)bad, 0.0);
        // ---- rtx_engine_evaluate_topk (ABI 8): the loop of evaluation.evaluate in one call -- two batches of the same users, nDCG@k / Recall@k
        //      against the held-out matrix, compared with the reference's formulas (rectorch/metrics.py:136-147, 187-196) evaluated on
        //      the host from the scores rtx_engine_forward(remove_train = 1) has just returned (same bits: the ranking cannot differ)
        if (!dense_api && B >= 4) {
            const int ks[2] = {5, 20};
            const int64_t offs[3] = {0, B / 2, B};
            float* d_scratch; double *d_nd, *d_rc;
            CK(hipMalloc(&d_scratch, (size_t)(B + 3) * I * 4)); CK(hipMalloc(&d_nd, 2 * B * 8)); CK(hipMalloc(&d_rc, 2 * B * 8));
            RT(rtx_engine_evaluate_topk(eng, ctr, cte, d_ids, offs, 2, ks, 2, d_scratch, d_nd, d_rc, nullptr));
            CK(hipDeviceSynchronize());
            std::vector<double> nd(2 * B), rc(2 * B);
            CK(hipMemcpy(nd.data(), d_nd, 2 * B * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(rc.data(), d_rc, 2 * B * 8, hipMemcpyDeviceToHost));
            double worst = 0;
            for (int q = 0; q < 2; ++q)
                for (int b = 0; b < B; ++b) {
                    const int k = std::min(ks[q], I);
                    std::vector<int> order(I);
                    for (int i = 0; i < I; ++i) order[i] = i;
                    const float* sc = pr.data() + (size_t)b * I;
                    std::stable_sort(order.begin(), order.end(), [&](int u, int v) { return sc[u] > sc[v]; });   // score descending, index ascending among ties
                    const float* held = gt.data() + (size_t)b * I;
                    double dcg = 0, hits = 0, n = 0, npos = 0;
                    for (int i = 0; i < I; ++i) { n += held[i]; npos += held[i] > 0.f; }
                    for (int r = 0; r < k; ++r) { dcg += held[order[r]] / log2((double)r + 2.0); hits += held[order[r]] > 0.f; }
                    double idcg = 0;
                    for (int r = 0; r < std::min((int)n, k); ++r) idcg += 1.0 / log2((double)r + 2.0);
                    const double want_n = dcg / idcg, want_r = hits / std::min((double)k, npos);
                    const double got_n = nd[(size_t)q * B + b], got_r = rc[(size_t)q * B + b];
                    auto diff = [](double a, double c) { return (std::isnan(a) && std::isnan(c)) ? 0.0 : (std::isnan(a) != std::isnan(c)) ? 1.0 : fabs(a - c) / std::max(1e-300, fabs(c)); };
                    worst = std::max(worst, std::max(diff(got_n, want_n), diff(got_r, want_r)));
                }
            check("evaluate_topk ndcg / recall vs host formulas", worst, 1e-12);
            hipFree(d_scratch); hipFree(d_nd); hipFree(d_rc);
        }
    }

    // ---- two training steps with injected RNG
    std::vector<std::vector<float>> o_p = net.params, o_m, o_v, o_g;
    for (auto& w : net.params) { o_m.emplace_back(w.size(), 0.f); o_v.emplace_back(w.size(), 0.f); o_g.emplace_back(w.size(), 0.f); }
    const float wd = vae ? 0.f : 0.001f;
    for (int step = 1; step <= 2; ++step) {
        std::vector<const float*> cp;
        std::vector<float*> gp;
        for (int t = 0; t < nt; ++t) { cp.push_back(o_p[t].data()); gp.push_back(o_g[t].data()); }
        double o_loss = 0;
        orc_forward_backward(&oc, cp.data(), x.data(), use_te ? gt.data() : nullptr, B, 1, mask.data(), vae ? eps.data() : nullptr, beta,
                             lam, 1.f / B, o_logits.data(), o_mu.data(), o_lv.data(), &o_loss, gp.data());
        rtx_step sp = {};
        sp.beta = beta; sp.lam = lam; sp.inv_batch = 1.f / B;
        sp.lr = 1e-3f; sp.beta1 = 0.9f; sp.beta2 = 0.999f; sp.eps = 1e-8f; sp.weight_decay = wd; sp.step = step;
        sp.dropout_mask = d_mask; sp.eps_noise = vae ? d_eps : nullptr;
        RT(rtx_engine_loss_grads(eng, &bt, &sp, d_loss, d_loss + 1, nullptr, nullptr, nullptr));
        CK(hipDeviceSynchronize());
        float g_loss = d2h(d_loss, 1)[0];
        char nm[64];
        snprintf(nm, sizeof nm, "step%d loss (%.5f)", step, o_loss);
        check(nm, fabs(g_loss - o_loss) / fabs(o_loss), tol_loss);
        std::vector<std::vector<float>> eng_g(nt);
        for (int t = 0; t < nt; ++t) {
            auto gg = d2h(dt.g[t], o_g[t].size());
            // DAE: the oracle folds lam*W/||W|| into the gradient; the engine folds it into Adam -> add it before comparing
            if (!vae && lam != 0.f) {
                double ss = 0;
                for (float w : o_p[t]) ss += (double)w * w;
                double nrm = sqrt(ss);
                for (size_t k = 0; k < gg.size(); ++k) gg[k] += (float)(lam * o_p[t][k] / nrm);
            }
            snprintf(nm, sizeof nm, "step%d grad[%d] %dx%d", step, t, net.rows[t], net.cols[t]);
            check(nm, rel_err(gg.data(), o_g[t].data(), gg.size()), tol_grad);
            eng_g[t] = gg;
        }
        RT(rtx_engine_apply_adam(eng, &sp, nullptr));
        CK(hipDeviceSynchronize());
        double worst = 0;
        for (int t = 0; t < nt; ++t) {
            orc_adam((int64_t)o_p[t].size(), o_p[t].data(), eng_g[t].data(), o_m[t].data(), o_v[t].data(), step, 1e-3f, 0.9f, 0.999f, 1e-8f, wd);
            auto gp2 = d2h(dt.p[t], o_p[t].size());
            double me = 0;
            for (size_t k = 0; k < gp2.size(); ++k) me = std::max(me, fabs((double)gp2[k] - o_p[t][k]));
            worst = std::max(worst, me);
        }
        snprintf(nm, sizeof nm, "step%d params max|diff|", step);
        // the oracle's Adam is fed the ENGINE's gradients, so this isolates the fused Adam kernel (f32 in both modes).
        // Mult-DAE: the regulariser's lam * p / ||p|| is added on the host in double here and in float (with the norm from
        // float atomics) on the device; where it CANCELS the likelihood gradient to |g| ~ 1e-8 the first Adam step
        // lr * g / (|g| + eps) amplifies that rounding (a handful of the 1.8 M elements at the 3000-item shape).
        check(nm, worst, (!vae && lam != 0.f) ? 5e-5 : 2e-6);
        // feed the engine's own updated parameters back to the oracle so step 2 tests the refreshed shadows, not drift
        for (int t = 0; t < nt; ++t) {
            o_p[t] = d2h(dt.p[t], o_p[t].size());
            o_m[t] = d2h(dt.m[t], o_m[t].size());
            o_v[t] = d2h(dt.v[t], o_v[t].size());
        }
    }
    {
        float acc = d2h(d_loss + 1, 1)[0];
        printf("    loss_accum after 2 steps = %.5f\n", acc);
    }
    // ---- step 3 through rtx_engine_train_step: the fused path (Adam of the big matrices inside the dW GEMM epilogue,
    //      data gradient before weight gradient), gradients kept for the comparison
    {
        const int step = 3;
        std::vector<const float*> cp;
        std::vector<float*> gp;
        for (int t = 0; t < nt; ++t) { cp.push_back(o_p[t].data()); gp.push_back(o_g[t].data()); }
        double o_loss = 0;
        orc_forward_backward(&oc, cp.data(), x.data(), use_te ? gt.data() : nullptr, B, 1, mask.data(), vae ? eps.data() : nullptr, beta,
                             lam, 1.f / B, o_logits.data(), o_mu.data(), o_lv.data(), &o_loss, gp.data());
        rtx_step sp = {};
        sp.beta = beta; sp.lam = lam; sp.inv_batch = 1.f / B;
        sp.lr = 1e-3f; sp.beta1 = 0.9f; sp.beta2 = 0.999f; sp.eps = 1e-8f; sp.weight_decay = wd; sp.step = step;
        sp.flags = RTX_STEP_KEEP_GRADS;
        sp.dropout_mask = d_mask; sp.eps_noise = vae ? d_eps : nullptr;
        for (int t = 0; t < nt; ++t) CK(hipMemset(dt.g[t], 0xff, o_g[t].size() * sizeof(float)));
        RT(rtx_engine_train_step(eng, &bt, &sp, d_loss, nullptr, nullptr));
        CK(hipDeviceSynchronize());
        char nm[64];
        snprintf(nm, sizeof nm, "fused step3 loss (%.5f)", o_loss);
        check(nm, fabs(d2h(d_loss, 1)[0] - o_loss) / fabs(o_loss), tol_loss);
        double worst_g = 0, worst_p = 0;
        for (int t = 0; t < nt; ++t) {
            auto gg = d2h(dt.g[t], o_g[t].size());
            if (!vae && lam != 0.f) {
                double ss = 0;
                for (float w : o_p[t]) ss += (double)w * w;
                double nrm = sqrt(ss);
                for (size_t k = 0; k < gg.size(); ++k) gg[k] += (float)(lam * o_p[t][k] / nrm);
            }
            worst_g = std::max(worst_g, rel_err(gg.data(), o_g[t].data(), gg.size()));
            orc_adam((int64_t)o_p[t].size(), o_p[t].data(), gg.data(), o_m[t].data(), o_v[t].data(), step, 1e-3f, 0.9f, 0.999f, 1e-8f, wd);
            auto gp2 = d2h(dt.p[t], o_p[t].size());
            auto gm2 = d2h(dt.m[t], o_p[t].size());
            for (size_t k = 0; k < gp2.size(); ++k) {
                worst_p = std::max(worst_p, fabs((double)gp2[k] - o_p[t][k]));
                worst_p = std::max(worst_p, fabs((double)gm2[k] - o_m[t][k]));
            }
        }
        check("fused step3 grads", worst_g, tol_grad);
        check("fused step3 params+exp_avg max|diff|", worst_p, 2e-6);
        // the refreshed compute copies must be the updated weights: an eval forward agrees with the oracle on them
        orc_forward_backward(&oc, cp.data(), x.data(), nullptr, B, 0, nullptr, nullptr, 0.f, 0.f, 1.f / B, o_logits.data(), o_mu.data(),
                             o_lv.data(), nullptr, nullptr);
        RT(rtx_engine_forward(eng, &bt, 0, nullptr, 0, d_logits, nullptr, nullptr, nullptr));
        CK(hipDeviceSynchronize());
        auto fl = d2h(d_logits, (size_t)B * I);
        check("logits after fused step", rel_err(fl.data(), o_logits.data(), fl.size()), tol_fwd);
    }
    rtx_engine_destroy(eng);
    rtx_csr_destroy(ctr); rtx_csr_destroy(cte);
    dt.release();
    hipFree(d_ids); hipFree(d_mask); hipFree(d_eps); hipFree(d_logits); hipFree(d_mu); hipFree(d_lv); hipFree(d_loss); hipFree(d_x); hipFree(d_gt);
}

// Philox path: determinism under the same (seed, offset), change under a different one, keep-rate ~ 1-p
static void philox_case()
{
    printf("[philox] dropout / eps from the counter RNG\n");
    Net net = make_net({300, 64, 16}, {16, 64, 300}, ORC_VAE, 0.5f, 0.1f);
    const int B = 40, I = 300;
    Csr tr = make_csr(64, I, 0.2f, false, -1);
    rtx_cfg rc = rcfg(net, RTX_FP32, B);
    rtx_engine* eng;
    RT(rtx_engine_create(&rc, &eng));
    apply_options(eng);
    DevTensors dt;
    dt.alloc(net);
    RT(rtx_engine_bind(eng, dt.p.data(), dt.g.data(), dt.m.data(), dt.v.data()));
    rtx_csr* ctr;
    RT(rtx_csr_upload(tr.indptr.data(), tr.indices.data(), nullptr, tr.rows, I, &ctr));
    float* d_loss;
    CK(hipMalloc(&d_loss, 4));
    rtx_batch bt = {};
    bt.csr = ctr; bt.batch = B;
    float l[3];
    uint64_t seeds[3] = {42, 42, 43};
    for (int k = 0; k < 3; ++k) {
        rtx_step sp = {};
        sp.beta = 0.2f; sp.inv_batch = 1.f / B; sp.lr = 1e-3f; sp.beta1 = .9f; sp.beta2 = .999f; sp.eps = 1e-8f; sp.step = 1;
        sp.seed = seeds[k]; sp.offset = 7;
        RT(rtx_engine_loss_grads(eng, &bt, &sp, d_loss, nullptr, nullptr, nullptr, nullptr));
        CK(hipDeviceSynchronize());
        l[k] = d2h(d_loss, 1)[0];
    }
    printf("    loss(seed42)=%.6f loss(seed42)=%.6f loss(seed43)=%.6f\n", l[0], l[1], l[2]);
    check("same seed -> same loss", l[0] == l[1] ? 0.0 : 1.0, 0.0);
    check("other seed -> other loss", l[0] != l[2] ? 0.0 : 1.0, 0.0);
    // keep-rate: host replica of the device decision
    long keep = 0, tot = 200000;
    for (long i = 0; i < tot; ++i) keep += rtx_dropout_keep(42, 7, (uint64_t)i, 0.5f);
    check("keep rate ~0.5", fabs((double)keep / tot - 0.5), 0.01);
    rtx_engine_destroy(eng); rtx_csr_destroy(ctr); dt.release(); hipFree(d_loss);
}

static void perf_case(int numerics, int B, int steps, int splitk)
{
    printf("[perf] MultiVAE [20108,600,200] %s B=%d splitk=%d fuse=%d two_stream=%d in_on_main=%d nt_regstage=%d dw_cfg=%d sparse_in=%d\n", numerics ? "bf16" : "fp32", B, splitk, g_opt_fuse, g_opt_two, g_opt_inmain, g_opt_ntreg, g_opt_dw_cfg, g_opt_sparse);
    Net net = make_net({20108, 600, 200}, {200, 600, 20108}, ORC_VAE, 0.5f, 0.1f);
    const int I = 20108, U = 4096;
    Csr tr;
    tr.rows = U; tr.cols = I;
    tr.indptr.push_back(0);
    for (int r = 0; r < U; ++r) {  // ~74 items per user, popularity-skewed
        int deg = 20 + (int)(rnd() % 110);
        std::vector<int> it;
        for (int k = 0; k < deg; ++k) { float u = frand(); it.push_back((int)(u * u * u * (I - 1))); }
        std::sort(it.begin(), it.end());
        it.erase(std::unique(it.begin(), it.end()), it.end());
        for (int j : it) tr.indices.push_back(j);
        tr.indptr.push_back((int64_t)tr.indices.size());
    }
    rtx_cfg rc = rcfg(net, numerics, B);
    rc.splitk = splitk;
    rtx_engine* eng;
    RT(rtx_engine_create(&rc, &eng));
    apply_options(eng);
    DevTensors dt;
    dt.alloc(net);
    RT(rtx_engine_bind(eng, dt.p.data(), dt.g.data(), dt.m.data(), dt.v.data()));
    rtx_csr* ctr;
    RT(rtx_csr_upload(tr.indptr.data(), tr.indices.data(), nullptr, U, I, &ctr));
    std::vector<int32_t> ids(B);
    for (int b = 0; b < B; ++b) ids[b] = (int32_t)(rnd() % U);
    int32_t* d_ids;
    CK(hipMalloc(&d_ids, B * 4)); CK(hipMemcpy(d_ids, ids.data(), B * 4, hipMemcpyHostToDevice));
    float* d_loss;
    CK(hipMalloc(&d_loss, 8)); CK(hipMemset(d_loss, 0, 8));
    rtx_batch bt = {};
    bt.csr = ctr; bt.row_ids = d_ids; bt.batch = B;
    rtx_step sp = {};
    sp.beta = 0.1f; sp.inv_batch = 1.f / B; sp.lr = 1e-3f; sp.beta1 = .9f; sp.beta2 = .999f; sp.eps = 1e-8f; sp.seed = 1;
    for (int i = 0; i < 5; ++i) { sp.step = i + 1; sp.offset = i; RT(rtx_engine_train_step(eng, &bt, &sp, d_loss, d_loss + 1, nullptr)); }
    CK(hipDeviceSynchronize());
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, 0));
    for (int i = 0; i < steps; ++i) { sp.step = 6 + i; sp.offset = 6 + i; RT(rtx_engine_train_step(eng, &bt, &sp, d_loss, d_loss + 1, nullptr)); }
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
    float ms;
    CK(hipEventElapsedTime(&ms, e0, e1));
    double us = ms * 1000.0 / steps, bytes, flops;
    rtx_engine_step_cost(eng, B, &bytes, &flops);
    printf("    %.1f us/step  %.0f users/s | algorithmic %.3f GB/step -> %.2f TB/s (%.1f%% of 8 TB/s), %.1f GFLOP/step -> %.1f TFLOP/s | loss %.4f\n",
           us, B / us * 1e6, bytes * 1e-9, bytes / us * 1e-6, bytes / us * 1e-6 / 8.0 * 100, flops * 1e-9, flops / us * 1e-6, d2h(d_loss, 1)[0]);
    // per-kernel table
    RT(rtx_engine_set_timing(eng, nullptr, 1));
    const int tsteps = 10;
    for (int i = 0; i < tsteps; ++i) { sp.step = 100 + i; sp.offset = 100 + i; RT(rtx_engine_train_step(eng, &bt, &sp, d_loss, d_loss + 1, nullptr)); }
    char names[64][48];
    float tot[64];
    int32_t cnt[64], n = 0;
    RT(rtx_engine_get_timings(eng, 64, names, tot, cnt, &n));
    double sum = 0;
    for (int i = 0; i < n; ++i) sum += tot[i];
    for (int i = 0; i < n; ++i)
        printf("      %-18s %3d launches/step  %8.1f us/step  %5.1f%%\n", names[i], cnt[i] / tsteps, tot[i] * 1000.0 / tsteps, 100.0 * tot[i] / sum);
    printf("      %-18s %26.1f us/step (event-bracketed sum)\n", "TOTAL", sum * 1000.0 / tsteps);
    RT(rtx_engine_set_timing(eng, nullptr, 0));
    rtx_engine_destroy(eng); rtx_csr_destroy(ctr); dt.release(); hipFree(d_ids); hipFree(d_loss);
}

int main(int argc, char** argv)
{
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("device: %s CUs=%d arch=%s | abi v%d\n", prop.name, prop.multiProcessorCount, prop.gcnArchName, rtx_abi_version());
    const bool perf_only = argc > 1 && !strcmp(argv[1], "perfonly");   // the perf table without the (CPU-oracle-bound) parity cases
    // parity cases run rtx_engine_train_step (step 3): in bf16 numerics that is the fused dW + Adam path (the default)
    const bool quick32 = argc > 1 && !strcmp(argv[1], "quick32");   // ... the same in float32 numerics, no perf table
    const bool quick = quick32 || (argc > 1 && !strcmp(argv[1], "quick"));   // bf16 numerics only, without the widest case and the variants
    for (int numerics = (quick && !quick32) ? 1 : 0; numerics < (quick32 ? 1 : 2) && !perf_only; ++numerics) {
        parity_case("small-vae", make_net({64, 16, 8}, {8, 16, 64}, ORC_VAE, 0.5f, 1.0f), numerics, 5, 9, 0.2f, false, false, false, 0.2f, 0.f);
        parity_case("small-vae-te-weighted", make_net({64, 16, 8}, {8, 16, 64}, ORC_VAE, 0.5f, 1.0f), numerics, 6, 9, 0.2f, true, true, false, 1.0f, 0.f);
        parity_case("deep-odd-vae", make_net({77, 21, 13, 5}, {5, 9, 77}, ORC_VAE, 0.3f, 1.0f), numerics, 7, 11, 0.15f, true, false, false, 0.3f, 0.f);
        parity_case("tiny-1layer-vae", make_net({2, 1}, {1, 2}, ORC_VAE, 0.1f, 1.0f), numerics, 2, 3, 0.7f, false, false, true, 1.0f, 0.f);
        parity_case("small-dae", make_net({64, 16, 8}, {8, 16, 64}, ORC_DAE, 0.5f, 1.0f), numerics, 5, 9, 0.2f, false, false, false, 0.f, 0.2f);
        parity_case("dense-api-vae-te", make_net({130, 40, 12}, {12, 40, 130}, ORC_VAE, 0.5f, 0.5f), numerics, 33, 40, 0.1f, true, true, true, 0.5f, 0.f);
        parity_case("mid-vae", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_VAE, 0.5f, 0.3f), numerics, 300, 400, 0.02f, false, false, false, 0.2f, 0.f);
        // wide enough that the engine picks the 8-wave tiles (256x128 / 128x256) and split-K in multiples of 8
        if (!quick) parity_case("wide-vae", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), numerics, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
    }
    if (!perf_only) philox_case();
    if (!perf_only && !quick) {   // the two-kernel train_step (gradients stored, one multi-tensor Adam launch) and the other tile configurations of the
        // weight-gradient kernel, on a shape with big layers, against the same oracle
        g_opt_fuse = 0;
        parity_case("wide-vae-unfused-step", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        g_opt_fuse = 1;
        for (int cfg : {1, 2, 3, 5, 7}) {   // (5, 7: the 256-column tiles of round 6; in the grouped launch the hidden matrices' rows are an odd number of 128-column blocks)
            g_opt_dw_cfg = cfg;
            parity_case("wide-vae-dwcfg", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
            parity_case("mid-dae-dwcfg", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_DAE, 0.5f, 0.3f), RTX_BF16, 300, 400, 0.02f, false, false, false, 0.f, 0.2f);
        }
        g_opt_dw_cfg = 0;
        g_opt_lse = 0;
        parity_case("wide-vae-nolsefuse", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        g_opt_lse = 1;
        g_opt_inmain = 0;    // the encoder matrix's kernel on the side stream as well (small layers + loss reduction on the caller's)
        parity_case("wide-vae-in-on-side", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        parity_case("mid-dae-in-on-side", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_DAE, 0.5f, 0.3f), RTX_BF16, 300, 400, 0.02f, false, false, false, 0.f, 0.2f);
        g_opt_inmain = 1;
        g_opt_two = 0;
        parity_case("wide-vae-one-stream", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        parity_case("mid-dae-one-stream", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_DAE, 0.5f, 0.3f), RTX_BF16, 300, 400, 0.02f, false, false, false, 0.f, 0.2f);
        g_opt_two = 1;
        g_opt_ntreg = 0;     // every NT contraction on the LDS-DMA kernel (log-sum-exp partials from its epilogue)
        parity_case("wide-vae-nt-dma", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        g_opt_ntreg = 1;
        g_opt_sparse = 0;    // the first layer as the dense split-K product (what float32 numerics and densified batches use)
        parity_case("wide-vae-dense-in", make_net({8100, 600, 200}, {200, 600, 8100}, ORC_VAE, 0.5f, 0.3f), RTX_BF16, 512, 600, 0.01f, false, false, false, 0.2f, 0.f);
        parity_case("mid-dae-dense-in", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_DAE, 0.5f, 0.3f), RTX_BF16, 300, 400, 0.02f, false, false, false, 0.f, 0.2f);
        g_opt_sparse = 1;
        parity_case("mid-dae", make_net({3000, 600, 200}, {200, 600, 3000}, ORC_DAE, 0.5f, 0.3f), RTX_BF16, 300, 400, 0.02f, false, false, false, 0.f, 0.2f);
    }
    if (perf_only || (quick && !quick32)) {
        const int B = argc > 2 ? atoi(argv[2]) : 500;
        perf_case(RTX_BF16, B, 50, 0);
        g_opt_sparse = 0; perf_case(RTX_BF16, B, 50, 0); g_opt_sparse = 1;    // dense first layer
        perf_case(RTX_BF16, B, 50, 0);
    }
    if (argc > 1 && !strcmp(argv[1], "perf")) {
        const int B = argc > 2 ? atoi(argv[2]) : 500;
        perf_case(RTX_BF16, B, 50, 0);                                  // shipped configuration
        perf_case(RTX_BF16, B, 50, 0);                                  // shipped configuration again (box drift)
        g_opt_inmain = 0; perf_case(RTX_BF16, B, 50, 0); g_opt_inmain = 1;   // encoder matrix's kernel on the side stream
        g_opt_two = 0; perf_case(RTX_BF16, B, 50, 0); g_opt_two = 1;          // one stream
        g_opt_sparse = 0; perf_case(RTX_BF16, B, 50, 0); g_opt_sparse = 1;    // dense first layer
        if (argc > 3) {
            g_opt_fuse = 0; perf_case(RTX_BF16, B, 50, 0); g_opt_fuse = 1;
            perf_case(RTX_FP32, B, 20, 0);
        }
    }
    printf("%s (%d failing checks)\n", g_fail ? "ENGINE TESTS FAILED" : "ENGINE TESTS PASSED", g_fail);
    return g_fail ? 1 : 0;
}
