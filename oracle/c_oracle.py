"""ctypes binding of oracle/mvae_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None
MAXL = 8


class OrcCfg(C.Structure):
    _fields_ = [("n_enc", C.c_int32), ("n_dec", C.c_int32),
                ("enc_dims", C.c_int32 * (MAXL + 1)), ("dec_dims", C.c_int32 * (MAXL + 1)),
                ("variant", C.c_int32), ("dropout_p", C.c_float), ("cond_dim", C.c_int32)]


def build():
    """Compile the C restatement with gcc (no GPU needed)."""
    subprocess.check_call(["make", "-C", _HERE, "-s"])


def lib():
    global _LIB
    if _LIB is None:
        path = os.path.join(_HERE, "libmvae_oracle.so")
        if not os.path.exists(path):
            build()
        _LIB = C.CDLL(path)
        _LIB.orc_forward_backward.restype = C.c_int
        _LIB.orc_predict.restype = C.c_int
    return _LIB


def make_cfg(enc_dims, dec_dims, variant="vae", dropout=0.5, cond_dim=0):
    cfg = OrcCfg()
    cfg.n_enc, cfg.n_dec = len(enc_dims) - 1, len(dec_dims) - 1
    for i, d in enumerate(enc_dims):
        cfg.enc_dims[i] = int(d)
    for i, d in enumerate(dec_dims):
        cfg.dec_dims[i] = int(d)
    cfg.variant = 0 if variant == "vae" else 1
    cfg.dropout_p = float(dropout)
    cfg.cond_dim = int(cond_dim)
    return cfg


def _fptrs(arrs):
    arr_t = C.POINTER(C.c_float) * len(arrs)
    return arr_t(*[a.ctypes.data_as(C.POINTER(C.c_float)) for a in arrs])


def _p(a, t):
    return None if a is None else a.ctypes.data_as(C.POINTER(t))


def param_shapes(enc_dims, dec_dims, variant="vae", cond_dim=0):
    enc_dims = [enc_dims[0] + cond_dim] + list(enc_dims[1:])
    enc_out = list(enc_dims[1:])
    if variant == "vae":
        enc_out[-1] *= 2
    shapes = []
    for i, o in zip(enc_dims[:-1], enc_out):
        shapes += [(o, i), (o,)]
    for i, o in zip(dec_dims[:-1], dec_dims[1:]):
        shapes += [(o, i), (o,)]
    return shapes


def forward_backward(enc_dims, dec_dims, params, x, gt=None, training=False, mask=None, eps=None,
                     beta=0.0, lam=0.0, inv_batch=None, variant="vae", dropout=0.5, want_grads=True, cond_dim=0):
    """Returns dict(logits, mu, logvar, loss, grads).  params: list of float32 arrays (W0,b0,W1,b1..)."""
    cfg = make_cfg(enc_dims, dec_dims, variant, dropout, cond_dim)
    params = [np.ascontiguousarray(p, dtype=np.float32) for p in params]
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, I = x.shape[0], x.shape[1] - cond_dim
    Z = enc_dims[-1]
    gt_ = None if gt is None else np.ascontiguousarray(gt, dtype=np.float32)
    mask_ = None if mask is None else np.ascontiguousarray(mask, dtype=np.uint8)
    eps_ = None if eps is None else np.ascontiguousarray(eps, dtype=np.float32)
    logits = np.empty((B, I), np.float32)
    mu = np.empty((B, Z), np.float32)
    logvar = np.empty((B, Z), np.float32)
    loss = C.c_double(0.0)
    grads = [np.empty_like(p) for p in params] if want_grads else None
    rc = lib().orc_forward_backward(
        C.byref(cfg), _fptrs(params), _p(x, C.c_float), _p(gt_, C.c_float), C.c_int(B), C.c_int(int(training)),
        _p(mask_, C.c_uint8), _p(eps_, C.c_float), C.c_float(beta), C.c_float(lam),
        C.c_float(1.0 / B if inv_batch is None else inv_batch),
        _p(logits, C.c_float), _p(mu, C.c_float), _p(logvar, C.c_float), C.byref(loss),
        _fptrs(grads) if want_grads else None)
    assert rc == 0, "bad oracle configuration"
    return dict(logits=logits, mu=mu if variant == "vae" else None, logvar=logvar if variant == "vae" else None,
                loss=loss.value, grads=grads)


def predict(enc_dims, dec_dims, params, x, remove_train=True, variant="vae", cond_dim=0):
    cfg = make_cfg(enc_dims, dec_dims, variant, 0.0, cond_dim)
    params = [np.ascontiguousarray(p, dtype=np.float32) for p in params]
    x = np.ascontiguousarray(x, dtype=np.float32)
    B, I = x.shape[0], x.shape[1] - cond_dim
    Z = enc_dims[-1]
    logits = np.empty((B, I), np.float32)
    mu = np.empty((B, Z), np.float32)
    logvar = np.empty((B, Z), np.float32)
    rc = lib().orc_predict(C.byref(cfg), _fptrs(params), _p(x, C.c_float), C.c_int(B), C.c_int(int(remove_train)),
                           _p(logits, C.c_float), _p(mu, C.c_float), _p(logvar, C.c_float))
    assert rc == 0
    if variant == "vae":
        return logits, mu, logvar
    return (logits,)


def adam(p, g, m, v, step, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0):
    """In-place Adam update of float32 arrays p, m, v."""
    for a in (p, m, v):
        assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"]
    g = np.ascontiguousarray(g, dtype=np.float32)
    lib().orc_adam(C.c_int64(p.size), _p(p, C.c_float), _p(g, C.c_float), _p(m, C.c_float), _p(v, C.c_float),
                   C.c_int(step), C.c_float(lr), C.c_float(beta1), C.c_float(beta2), C.c_float(eps),
                   C.c_float(weight_decay))


def csr_rows_to_dense(csr, row_ids):
    indptr = np.ascontiguousarray(csr.indptr, dtype=np.int64)
    indices = np.ascontiguousarray(csr.indices, dtype=np.int32)
    values = np.ascontiguousarray(csr.data, dtype=np.float64)
    rows = np.ascontiguousarray(row_ids, dtype=np.int64)
    out = np.empty((rows.size, csr.shape[1]), np.float32)
    lib().orc_csr_rows_to_dense(_p(indptr, C.c_int64), _p(indices, C.c_int32), _p(values, C.c_double),
                                _p(rows, C.c_int64), C.c_int(rows.size), C.c_int(csr.shape[1]), _p(out, C.c_float))
    return out


class OracleTrainer:
    """Multi-step trainer on top of the C oracle: what MultiVAE/MultiDAE.train_batch does
    (reference models.py:817-835 / 424-447) with injected RNG.  Keeps params + Adam state."""

    def __init__(self, enc_dims, dec_dims, params, variant="vae", dropout=0.5, beta=1.0, anneal_steps=0,
                 lam=0.2, lr=1e-3, weight_decay=None, cond_dim=0):
        self.enc_dims, self.dec_dims, self.variant, self.dropout = list(enc_dims), list(dec_dims), variant, dropout
        self.cond_dim = cond_dim
        self.params = [np.array(p, dtype=np.float32, copy=True) for p in params]
        self.m = [np.zeros_like(p) for p in self.params]
        self.v = [np.zeros_like(p) for p in self.params]
        self.beta, self.anneal_steps, self.lam, self.lr = beta, anneal_steps, lam, lr
        self.wd = (0.0 if variant == "vae" else 0.001) if weight_decay is None else weight_decay
        self.gradient_updates = 0.0
        self.step = 0

    def anneal_beta(self):
        if self.variant != "vae":
            return 0.0
        if self.anneal_steps > 0:
            return min(self.beta, self.gradient_updates / self.anneal_steps)
        return self.beta

    def train_batch(self, x, gt=None, mask=None, eps=None, inv_batch=None):
        out = forward_backward(self.enc_dims, self.dec_dims, self.params, x, gt, True, mask, eps,
                               beta=self.anneal_beta(), lam=self.lam if self.variant == "dae" else 0.0,
                               inv_batch=inv_batch, variant=self.variant, dropout=self.dropout, cond_dim=self.cond_dim)
        self.step += 1
        for p, g, m, v in zip(self.params, out["grads"], self.m, self.v):
            adam(p.reshape(-1), g.reshape(-1), m.reshape(-1), v.reshape(-1), self.step, self.lr,
                 weight_decay=self.wd)
        self.gradient_updates += 1.0
        self.last = out
        return out["loss"]

    def predict(self, x, remove_train=True):
        return predict(self.enc_dims, self.dec_dims, self.params, x, remove_train, self.variant, self.cond_dim)
