/*
 * oracle/mvae_oracle.c -- see mvae_oracle.h.  TEST INFRASTRUCTURE ONLY (the checker, never the
 * product path).  Plain loops, double arithmetic; OpenMP only to make full-size checks bearable.
 */
#include "mvae_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
    int in, out, tanh_act;
} layer_t;

static int build_layers(const orc_cfg* c, layer_t* L)
{
    if (c->n_enc < 1 || c->n_dec < 1 || c->n_enc > ORC_MAX_LAYERS || c->n_dec > ORC_MAX_LAYERS) return -1;
    int n = 0;
    for (int e = 0; e < c->n_enc; ++e, ++n) {
        L[n].in = c->enc_dims[e] + (e == 0 ? c->cond_dim : 0); /* CMultiVAE_net: temp_dims[0] += cond_dim (nets.py:459-460) */
        L[n].out = c->enc_dims[e + 1];
        L[n].tanh_act = 1;
        if (e == c->n_enc - 1 && c->variant == ORC_VAE) {
            /* VAE_net.__init__ (nets.py:262-265): last encoder layer emits mu|logvar, no tanh
             * (MultiVAE_net.encode nets.py:398-404) */
            L[n].out = 2 * c->enc_dims[e + 1];
            L[n].tanh_act = 0;
        }
    }
    for (int d = 0; d < c->n_dec; ++d, ++n) {
        L[n].in = c->dec_dims[d];
        L[n].out = c->dec_dims[d + 1];
        L[n].tanh_act = (d != c->n_dec - 1); /* decode: tanh on all but the last (nets.py:227-233, 413-417) */
    }
    return n;
}

int orc_n_tensors(const orc_cfg* cfg) { return 2 * (cfg->n_enc + cfg->n_dec); }

void orc_tensor_shape(const orc_cfg* cfg, int t, int* rows, int* cols)
{
    layer_t L[2 * ORC_MAX_LAYERS];
    build_layers(cfg, L);
    const layer_t* l = &L[t / 2];
    *rows = l->out;
    *cols = (t & 1) ? 1 : l->in;
}

/* out[B,N] = act(in[B,K] * W[N,K]^T + b)   (nn.Linear + optional tanh) */
static void linear_fwd(const double* in, const float* W, const float* b, double* out, int B, int K, int N,
                       int tanh_act)
{
#pragma omp parallel for schedule(static)
    for (long bn = 0; bn < (long)B * N; ++bn) {
        int r = (int)(bn / N), n = (int)(bn % N);
        const double* a = in + (long)r * K;
        const float* w = W + (long)n * K;
        double s = 0.0;
        for (int k = 0; k < K; ++k) s += a[k] * (double)w[k];
        s += (double)b[n];
        out[bn] = tanh_act ? tanh(s) : s;
    }
}

int orc_forward_backward(const orc_cfg* cfg, const float* const* params, const float* x, const float* gt,
                         int B, int training, const uint8_t* mask, const float* eps, float beta, float lam,
                         float inv_batch, float* logits, float* mu_out, float* logvar_out, double* loss_out,
                         float* const* grads)
{
    layer_t L[2 * ORC_MAX_LAYERS];
    const int NL = build_layers(cfg, L);
    if (NL < 0) return -1;
    const int I = cfg->enc_dims[0];
    const int Z = cfg->enc_dims[cfg->n_enc]; /* latent */
    if (cfg->dec_dims[0] != Z || cfg->dec_dims[cfg->n_dec] != I) return -1;
    const int vae = (cfg->variant == ORC_VAE);
    const int Iin = I + cfg->cond_dim; /* CMultiVAE_net.encode (nets.py:467-471): x = [items | condition] */
    if (!gt) {
        if (cfg->cond_dim) return -1;  /* a conditioned row is not a valid target */
        gt = x;
    }

    /* activations: act[l] = input of layer l, act[NL] = logits.  For the VAE the decoder input is z. */
    double** act = (double**)calloc((size_t)NL + 1, sizeof(double*));
    act[0] = (double*)malloc(sizeof(double) * (size_t)B * Iin);
    /* F.normalize (nets.py:395 / 220): x / max(||x||_2, 1e-12); dropout (nets.py:396-397 / 221-222) */
    const double scale = (training && cfg->dropout_p > 0.f)
                             ? (cfg->dropout_p < 1.f ? 1.0 / (1.0 - (double)cfg->dropout_p) : 0.0)
                             : 1.0;
#pragma omp parallel for schedule(static)
    for (int b = 0; b < B; ++b) {
        double ss = 0.0;
        for (int i = 0; i < I; ++i) ss += (double)x[(long)b * Iin + i] * (double)x[(long)b * Iin + i];
        double nrm = sqrt(ss);
        if (nrm < 1e-12) nrm = 1e-12;
        for (int i = 0; i < I; ++i) {
            double v = (double)x[(long)b * Iin + i] / nrm;
            if (training && cfg->dropout_p > 0.f) {
                int keep = mask ? mask[(long)b * I + i] != 0 : 1;
                v = keep ? v * scale : 0.0;
            }
            act[0][(long)b * Iin + i] = v;
        }
        /* the condition columns are concatenated raw, after normalisation and dropout */
        for (int i = I; i < Iin; ++i) act[0][(long)b * Iin + i] = (double)x[(long)b * Iin + i];
    }
    double* mu = NULL;      /* views into the last encoder output */
    double* h_enc = NULL;   /* raw output of the last encoder layer [B, out] */
    double* std_ = NULL;
    for (int l = 0; l < NL; ++l) {
        double* out = (double*)malloc(sizeof(double) * (size_t)B * L[l].out);
        linear_fwd(act[l], params[2 * l], params[2 * l + 1], out, B, L[l].in, L[l].out, L[l].tanh_act);
        if (l == cfg->n_enc - 1 && vae) {
            /* split mu | logvar (nets.py:403-404), reparameterise (nets.py:317-320, 407-411) */
            h_enc = out;
            double* z = (double*)malloc(sizeof(double) * (size_t)B * Z);
            std_ = (double*)malloc(sizeof(double) * (size_t)B * Z);
            for (int b = 0; b < B; ++b)
                for (int j = 0; j < Z; ++j) {
                    double m = out[(long)b * 2 * Z + j], lv = out[(long)b * 2 * Z + Z + j];
                    double sd = exp(0.5 * lv);
                    std_[(long)b * Z + j] = sd;
                    double e = (training && eps) ? (double)eps[(long)b * Z + j] : 0.0;
                    z[(long)b * Z + j] = training ? m + e * sd : m;
                    if (mu_out) mu_out[(long)b * Z + j] = (float)m;
                    if (logvar_out) logvar_out[(long)b * Z + j] = (float)lv;
                }
            mu = out;
            act[l + 1] = z;
        } else {
            act[l + 1] = out;
        }
    }
    const double* Y = act[NL];
    if (logits)
        for (long k = 0; k < (long)B * I; ++k) logits[k] = (float)Y[k];

    if (loss_out || grads) {
        /* multinomial NLL (models.py:813 / 701): -mean_b sum_i log_softmax(Y)_bi * gt_bi */
        double* dY = (double*)malloc(sizeof(double) * (size_t)B * I);
        double nll = 0.0;
        for (int b = 0; b < B; ++b) {
            const double* y = Y + (long)b * I;
            double mx = y[0];
            for (int i = 1; i < I; ++i) mx = y[i] > mx ? y[i] : mx;
            double se = 0.0;
            for (int i = 0; i < I; ++i) se += exp(y[i] - mx);
            double lse = mx + log(se);
            double s = 0.0, dot = 0.0;
            for (int i = 0; i < I; ++i) {
                double g = (double)gt[(long)b * I + i];
                s += g;
                dot += g * y[i];
            }
            nll += s * lse - dot;
            for (int i = 0; i < I; ++i)
                dY[(long)b * I + i] = (s * exp(y[i] - lse) - (double)gt[(long)b * I + i]) * (double)inv_batch;
        }
        double loss = nll * (double)inv_batch;
        if (vae) {
            /* KLD = -0.5 * mean_b sum_j (1 + logvar - mu^2 - exp(logvar))   (models.py:814) */
            double kl = 0.0;
            for (int b = 0; b < B; ++b)
                for (int j = 0; j < Z; ++j) {
                    double m = h_enc[(long)b * 2 * Z + j], lv = h_enc[(long)b * 2 * Z + Z + j];
                    kl += 1.0 + lv - m * m - exp(lv);
                }
            loss += (double)beta * (-0.5 * kl * (double)inv_batch);
        } else {
            /* l2_reg = sum_W ||W||_2 over ALL parameters incl. biases (models.py:702-706) */
            for (int t = 0; t < 2 * NL; ++t) {
                long n = (long)L[t / 2].out * ((t & 1) ? 1 : L[t / 2].in);
                double ss = 0.0;
                for (long k = 0; k < n; ++k) ss += (double)params[t][k] * (double)params[t][k];
                loss += (double)lam * sqrt(ss);
            }
        }
        if (loss_out) *loss_out = loss;

        if (grads) {
            double* dout = dY; /* gradient w.r.t. the pre-activation of the current layer */
            for (int l = NL - 1; l >= 0; --l) {
                const int K = L[l].in, N = L[l].out;
                const double* a_in = act[l];
                /* for tanh layers convert d(out) into d(pre-activation) */
                if (L[l].tanh_act) {
                    const double* o = act[l + 1];
                    for (long k = 0; k < (long)B * N; ++k) dout[k] *= (1.0 - o[k] * o[k]);
                }
                float* gW = grads[2 * l];
                float* gb = grads[2 * l + 1];
#pragma omp parallel for schedule(static)
                for (int n = 0; n < N; ++n) {
                    double sb = 0.0;
                    for (int b = 0; b < B; ++b) sb += dout[(long)b * N + n];
                    gb[n] = (float)sb;
                    for (int k = 0; k < K; ++k) {
                        double s = 0.0;
                        for (int b = 0; b < B; ++b) s += dout[(long)b * N + n] * a_in[(long)b * K + k];
                        gW[(long)n * K + k] = (float)s;
                    }
                }
                if (l == 0) break;
                /* d(input of layer l) = dout * W */
                double* din = (double*)malloc(sizeof(double) * (size_t)B * K);
                const float* W = params[2 * l];
#pragma omp parallel for schedule(static)
                for (long bk = 0; bk < (long)B * K; ++bk) {
                    int b = (int)(bk / K), k = (int)(bk % K);
                    double s = 0.0;
                    for (int n = 0; n < N; ++n) s += dout[(long)b * N + n] * (double)W[(long)n * K + k];
                    din[bk] = s;
                }
                free(dout);
                dout = din;
                if (l == cfg->n_enc && vae) {
                    /* dout is dz [B,Z]; map to d[mu|logvar] [B,2Z]:
                     * dmu = dz + beta*mu/B ; dlogvar = dz*eps*0.5*std + beta*0.5*(exp(logvar)-1)/B */
                    double* dh = (double*)malloc(sizeof(double) * (size_t)B * 2 * Z);
                    for (int b = 0; b < B; ++b)
                        for (int j = 0; j < Z; ++j) {
                            double dz = dout[(long)b * Z + j];
                            double m = h_enc[(long)b * 2 * Z + j], lv = h_enc[(long)b * 2 * Z + Z + j];
                            double e = (training && eps) ? (double)eps[(long)b * Z + j] : 0.0;
                            double dzm = training ? dz : dz; /* eval: z = mu */
                            dh[(long)b * 2 * Z + j] = dzm + (double)beta * m * (double)inv_batch;
                            dh[(long)b * 2 * Z + Z + j] =
                                (training ? dz * e * 0.5 * std_[(long)b * Z + j] : 0.0) +
                                (double)beta * 0.5 * (exp(lv) - 1.0) * (double)inv_batch;
                        }
                    free(dout);
                    dout = dh;
                }
            }
            if (!vae && lam != 0.f) {
                for (int t = 0; t < 2 * NL; ++t) {
                    long n = (long)L[t / 2].out * ((t & 1) ? 1 : L[t / 2].in);
                    double ss = 0.0;
                    for (long k = 0; k < n; ++k) ss += (double)params[t][k] * (double)params[t][k];
                    double nrm = sqrt(ss);
                    if (nrm > 0.0)
                        for (long k = 0; k < n; ++k)
                            grads[t][k] = (float)((double)grads[t][k] + (double)lam * (double)params[t][k] / nrm);
                }
            }
            free(dout);
        } else {
            free(dY);
        }
    }

    for (int l = 0; l <= NL; ++l) free(act[l]);
    free(act);
    if (vae) {
        free(mu); /* == h_enc */
        free(std_);
    }
    return 0;
}

int orc_predict(const orc_cfg* cfg, const float* const* params, const float* x, int B, int remove_train,
                float* logits, float* mu, float* logvar)
{
    /* scoring reads no target: hand the input's item block as a placeholder when the rows are conditioned */
    int rc = orc_forward_backward(cfg, params, x, cfg->cond_dim ? x : NULL, B, 0, NULL, NULL, 0.f, 0.f, 0.f, logits, mu,
                                  logvar, NULL, NULL);
    if (rc) return rc;
    if (remove_train) {
        /* recon_x[x.nonzero()] = -inf   (models.py:623-624 / 471-472) */
        /* CMultiVAE.predict masks x[:, :-cond_dim].nonzero() only (models.py:952-953) */
        const int I = cfg->enc_dims[0], Iin = I + cfg->cond_dim;
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < I; ++i)
                if (x[(long)b * Iin + i] != 0.f) logits[(long)b * I + i] = -INFINITY;
    }
    return 0;
}

void orc_adam(int64_t n, float* p, const float* g, float* m, float* v, int step, float lr, float beta1,
              float beta2, float eps, float weight_decay)
{
    /* torch/optim/adam.py _single_tensor_adam, amsgrad=False, maximize=False */
    const double bc1 = 1.0 - pow((double)beta1, (double)step);
    const double bc2 = 1.0 - pow((double)beta2, (double)step);
    const double step_size = (double)lr / bc1;
    const double bc2_sqrt = sqrt(bc2);
    for (int64_t k = 0; k < n; ++k) {
        double gk = (double)g[k];
        if (weight_decay != 0.f) gk += (double)weight_decay * (double)p[k];
        double mk = (double)m[k] + (gk - (double)m[k]) * (1.0 - (double)beta1);
        double vk = (double)v[k] * (double)beta2 + (1.0 - (double)beta2) * gk * gk;
        double denom = sqrt(vk) / bc2_sqrt + (double)eps;
        p[k] = (float)((double)p[k] - step_size * (mk / denom));
        m[k] = (float)mk;
        v[k] = (float)vk;
    }
}

void orc_csr_rows_to_dense(const int64_t* indptr, const int32_t* indices, const double* values,
                           const int64_t* row_ids, int B, int n_cols, float* out)
{
    memset(out, 0, sizeof(float) * (size_t)B * n_cols);
    for (int b = 0; b < B; ++b) {
        int64_t u = row_ids[b];
        for (int64_t k = indptr[u]; k < indptr[u + 1]; ++k)
            out[(long)b * n_cols + indices[k]] += (float)(values ? values[k] : 1.0);
    }
}
