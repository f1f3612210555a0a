"""PyTorch-ROCm custom ops over the C ABI (``torch.ops.rectorch_hip.*``).

The north star asks that rectorch's evaluation/metrics reach the new path "via PyTorch-ROCm custom ops through a
thin C-ABI": these ops are that registration.  Each op is a thin adapter: tensors in, tensors out, the body is ONE
call into librectorch_hip on torch's current HIP stream (no torch arithmetic).  Engines and resident CSR
matrices are passed as integer handles (``Engine.op_handle`` / ``CsrMatrix.op_handle``) because dispatcher
schemas cannot carry opaque pointers.  The model classes call the same C entry points directly; the ops exist
for callers that want to stay inside the dispatcher (profilers, ``torch.ops`` users).

    torch.ops.rectorch_hip.csr_gather_dense(csr_handle, row_ids)                      -> Tensor [B, n_items]
    torch.ops.rectorch_hip.mvae_forward(engine_handle, x, training, remove_train, seed) -> (logits, mu, logvar)
    torch.ops.rectorch_hip.mdae_forward(engine_handle, x, training, remove_train, seed) -> logits
    torch.ops.rectorch_hip.multinomial_loss(recon, x, mu, logvar, beta)               -> Tensor []
    torch.ops.rectorch_hip.train_step_dense(engine_handle, x, target, step_scalars...) -> Tensor [] (loss)
"""
import weakref

import torch

from . import engine as _engine

_LIB = torch.library.Library("rectorch_hip", "DEF")
# integer handle -> object, WEAKLY: a handle lives as long as its Engine / CsrMatrix does and not a moment longer (an
# engine replaced by a larger one, or the engines of a discarded model, release their HBM buffers as usual)
_HANDLES = weakref.WeakValueDictionary()


def register_handle(obj):
    """Make ``obj`` (an Engine or CsrMatrix) reachable from an integer the dispatcher can carry, for as long as the caller
    keeps ``obj`` alive."""
    h = id(obj)
    _HANDLES[h] = obj
    return h


def _get(h):
    try:
        return _HANDLES[int(h)]
    except KeyError:
        raise RuntimeError("rectorch_hip op: handle %d does not name a live Engine / CsrMatrix" % int(h)) from None


_LIB.define("csr_gather_dense(int csr, Tensor row_ids) -> Tensor")
_LIB.define("mvae_forward(int engine, Tensor x, bool training, bool remove_train, int seed) -> (Tensor, Tensor, Tensor)")
_LIB.define("mdae_forward(int engine, Tensor x, bool training, bool remove_train, int seed) -> Tensor")
_LIB.define("multinomial_loss(Tensor recon, Tensor x, Tensor? mu, Tensor? logvar, float beta) -> Tensor")
_LIB.define("train_step_dense(int engine, Tensor x, Tensor? target, float beta, float lam, float lr, float beta1, "
            "float beta2, float eps, float weight_decay, int step, int seed) -> Tensor")


def _csr_gather_dense(csr, row_ids):
    return _get(csr).gather_dense(row_ids.to(torch.int32))


def _mvae_forward(engine, x, training, remove_train, seed):
    return _get(engine).forward(x, training=training, remove_train=remove_train, seed=seed)


def _mdae_forward(engine, x, training, remove_train, seed):
    return _get(engine).forward(x, training=training, remove_train=remove_train, seed=seed)[0]


def _multinomial_loss(recon, x, mu, logvar, beta):
    return _engine.multinomial_loss(recon, x, mu, logvar, beta)


def _train_step_dense(engine, x, target, beta, lam, lr, beta1, beta2, eps, weight_decay, step, seed):
    eng = _get(engine)
    loss = torch.zeros(1, dtype=torch.float32, device=x.device)
    st = eng._step(seed=seed, beta=beta, lam=lam, inv_batch=1.0 / x.shape[0], lr=lr, beta1=beta1, beta2=beta2, eps=eps,
                   weight_decay=weight_decay, step=step)
    eng.train_step(x, target, st, loss)
    return loss[0]


for _name, _fn in (("csr_gather_dense", _csr_gather_dense), ("mvae_forward", _mvae_forward),
                   ("mdae_forward", _mdae_forward), ("multinomial_loss", _multinomial_loss),
                   ("train_step_dense", _train_step_dense)):
    _LIB.impl(_name, _fn, "CUDA")       # "CUDA" is the HIP dispatch key on PyTorch-ROCm
