"""Python handles over the C ABI: device-resident CSR matrices and the Mult-VAE/DAE engine.

PyTorch is plumbing here (device memory, streams): tensors are passed to librectorch_hip as raw device
pointers on torch's current HIP stream.  No arithmetic of the path is done by torch.
"""
import ctypes as C
import os

import numpy as np
import torch

from . import _lib
from ._lib import Batch, Step, check, lib, stream_ptr


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


class CsrMatrix:
    """A scipy CSR matrix uploaded once to HBM (indptr int64, indices int32, values float32 or
    implicit ones).  Replaces the per-batch ``sparse[idx].toarray()`` of the reference's DataSampler
    (rectorch/samplers.py:99-105)."""

    def __init__(self, sparse):
        _lib.require_gpu()
        m = sparse.tocsr().copy()
        m.sum_duplicates()
        m.sort_indices()          # the device metrics binary-search the held-out rows
        self.shape = m.shape
        self.nnz = int(m.nnz)
        indptr = np.ascontiguousarray(m.indptr, dtype=np.int64)
        indices = np.ascontiguousarray(m.indices, dtype=np.int32)
        data = np.asarray(m.data)
        self.binary = bool(np.all(data == 1))
        values = None if self.binary else np.ascontiguousarray(data, dtype=np.float32)
        h = C.c_void_p()
        check(lib().rtx_csr_upload(indptr.ctypes.data_as(C.c_void_p), indices.ctypes.data_as(C.c_void_p),
                                   None if values is None else values.ctypes.data_as(C.c_void_p),
                                   C.c_int64(m.shape[0]), C.c_int32(m.shape[1]), C.byref(h)))
        self.handle = h

    @property
    def op_handle(self):
        """integer handle for the ``torch.ops.rectorch_hip`` custom ops (rectorch_amd/ops.py)"""
        from . import ops
        return ops.register_handle(self)

    def gather_dense(self, row_ids, out=None):
        """float32 [len(row_ids), n_cols] device tensor holding the given rows (K1 in dense form)."""
        n = int(row_ids.numel())
        if out is None:
            out = torch.empty((n, self.shape[1]), dtype=torch.float32, device=row_ids.device)
        if n:
            check(lib().rtx_csr_gather_dense(self.handle, _ptr(row_ids), n, _ptr(out), stream_ptr()))
        return out

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                lib().rtx_csr_destroy(h)
            except Exception:
                pass
            self.handle = None


class RowBatch:
    """What the resident DataSampler hands to the trainer on the fast path: row numbers only."""
    __slots__ = ("tr", "te", "rows", "global_len")

    def __init__(self, tr, te, rows, global_len=None):
        self.tr, self.te, self.rows = tr, te, rows
        # data parallel: users in the GLOBAL batch this one is a rank's slice of (rectorch_amd.parallel.shard_batch): the step's
        # 1 / global-batch scale then needs no collective and no host sync
        self.global_len = global_len

    def __len__(self):
        return int(self.rows.numel())


def tag_rows(t, rb):
    """Mark the dense tensor ``t`` as the image of the resident rows ``rb`` (what the device DataSampler yields).  The tag
    carries the tensor's version counter: once the caller edits ``t`` in place (``x.clamp_``, ``x[:, cold] = 0`` ...)
    the tag is stale and the tensor's CONTENTS are used, as the reference would (rectorch/models.py:819-822)."""
    t._rtx_rows = rb
    t._rtx_ver = t._version
    return t


def tagged_rows(t):
    """The RowBatch ``t`` was gathered from, or None when ``t`` is not tagged or was modified since."""
    rb = getattr(t, "_rtx_rows", None)
    if rb is not None and getattr(t, "_rtx_ver", None) == t._version:
        return rb
    return None


def _check_width(t, n_items, what):
    if n_items is not None and (t.dim() != 2 or t.shape[1] != n_items):
        raise _lib.RtxError("%s must be [batch, %d] (n_items of the network), got %s" % (what, n_items, tuple(t.shape)))


def make_batch(x, target=None, keep=None, n_items=None, n_in=None):
    """Build the C ``rtx_batch`` for ``x`` (a RowBatch, or a dense [B, n_items] device tensor, or a dense
    tensor carrying the RowBatch it was gathered from).  ``keep`` collects tensors that must outlive the call."""
    b = Batch()
    rb = x if isinstance(x, RowBatch) else tagged_rows(x)
    if rb is not None:
        b.csr = rb.tr.handle
        b.row_ids = rb.rows.data_ptr()
        b.batch = len(rb)
        if target is None and rb.te is not None and isinstance(x, RowBatch) and x.te is not None:
            b.target_csr = rb.te.handle
    else:
        _check_width(x, n_in if n_in is not None else n_items, "the input batch")
        x = x.to(torch.float32).contiguous()
        if keep is not None:
            keep.append(x)
        b.x_dense = x.data_ptr()
        b.batch = x.shape[0]
    if target is not None:
        trb = None if isinstance(target, RowBatch) else tagged_rows(target)
        if isinstance(target, RowBatch) and rb is not None and target.rows is rb.rows:
            b.target_csr = target.tr.handle      # the (input rows, target rows) pair of a sparse sampler
        elif trb is not None and rb is not None and trb.rows is rb.rows and trb.te is not None:
            b.target_csr = trb.te.handle
        else:
            if isinstance(target, RowBatch):
                target = target.tr.gather_dense(target.rows)
            _check_width(target, n_items, "the target batch")
            target = target.to(torch.float32).contiguous()
            if keep is not None:
                keep.append(target)
            if rb is not None:
                # input by rows, target dense: densify the input as well (mixed forms are not in the ABI)
                xin = rb.tr.gather_dense(rb.rows)
                if keep is not None:
                    keep.append(xin)
                b.csr = None
                b.row_ids = None
                b.x_dense = xin.data_ptr()
            b.target_dense = target.data_ptr()
    return b


class Engine:
    """One ``rtx_engine``: a network's compute state at one numerics mode, bound to the network's float32
    master parameters (the nn.Parameters themselves) and, for training, to gradient buffers and the Adam
    moments held in ``torch.optim.Adam``'s state."""

    def __init__(self, enc_dims, dec_dims, variant, dropout, numerics="bf16", max_batch=512, splitk=0, cond_dim=0):
        _lib.require_gpu()
        self.enc_dims, self.dec_dims = [int(d) for d in enc_dims], [int(d) for d in dec_dims]
        self.variant, self.numerics = variant, numerics
        self.max_batch = int(max_batch)
        self.n_items, self.latent = self.enc_dims[0], self.enc_dims[-1]
        self.cond_dim = int(cond_dim)
        self.n_in = self.n_items + self.cond_dim      # input columns: items, then the condition one-hot (CMultiVAE)
        cfg = _lib.make_cfg(self.enc_dims, self.dec_dims, variant, numerics, dropout, max_batch, splitk, cond_dim)
        h = C.c_void_p()
        check(lib().rtx_engine_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.n_tensors = lib().rtx_engine_n_tensors(h)
        self._bound = None
        self._keep = []
        # measurement / debugging: RTX_ENGINE_OPTS="hop_values=0,prefetch=0" sets engine options (rtx_engine_set_option) on
        # every engine this process creates -- A/B runs of whole test files without touching them
        for kv in filter(None, os.environ.get("RTX_ENGINE_OPTS", "").split(",")):
            k, v = kv.split("=")
            check(lib().rtx_engine_set_option(h, k.strip().encode(), int(v)))

    @property
    def op_handle(self):
        """integer handle for the ``torch.ops.rectorch_hip`` custom ops (rectorch_amd/ops.py)"""
        from . import ops
        return ops.register_handle(self)

    # ---- binding --------------------------------------------------------------------------------
    def bind(self, params, grads=None, exp_avg=None, exp_avg_sq=None):
        n = self.n_tensors
        assert len(params) == n, "expected %d parameter tensors, got %d" % (n, len(params))
        rows, cols = C.c_int32(), C.c_int32()
        for t, p in enumerate(params):
            check(lib().rtx_engine_tensor_shape(self.handle, t, C.byref(rows), C.byref(cols)))
            want = (rows.value, cols.value) if p.dim() == 2 else (rows.value,)
            if tuple(p.shape) != want or p.dtype != torch.float32 or not p.is_contiguous() or not p.is_cuda:
                raise _lib.RtxError("parameter %d must be a contiguous float32 HIP tensor of shape %s, got %s %s on %s"
                                    % (t, want, tuple(p.shape), p.dtype, p.device))
        arr_t = C.c_void_p * n

        def arr(ts):
            return None if ts is None else arr_t(*[t.data_ptr() for t in ts])
        check(lib().rtx_engine_bind(self.handle, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq)))
        self._bound = (list(params), grads, exp_avg, exp_avg_sq)   # keep the tensors alive

    def sync_shadows(self):
        check(lib().rtx_engine_sync_shadows(self.handle, stream_ptr()))

    # ---- forward family ---------------------------------------------------------------------------
    def _step(self, seed=0, offset=0, mask=None, noise=None, **kw):
        s = Step()
        s.seed, s.offset = int(seed) & (2 ** 64 - 1), int(offset)
        s.dropout_mask = None if mask is None else mask.data_ptr()
        s.eps_noise = None if noise is None else noise.data_ptr()
        for k, v in kw.items():
            setattr(s, k, v)
        return s

    def forward(self, x, training=False, remove_train=False, seed=0, offset=0, mask=None, noise=None, out=None, want_latent=True):
        """``out``: a float32 ``[>= batch, n_items]`` tensor to receive the scores (a loop over many batches reuses two of them instead
        of allocating 40 MB per batch); ``want_latent=False`` skips the (mu, logvar) outputs of a VAE."""
        keep = []
        b = make_batch(x, keep=keep, n_items=self.n_items, n_in=self.n_in)
        dev = torch.device("cuda", torch.cuda.current_device())
        if out is not None:
            assert out.dtype == torch.float32 and out.is_contiguous() and out.shape[0] >= b.batch and out.shape[1] == self.n_items
            logits = out[:b.batch]
        else:
            logits = torch.empty((b.batch, self.n_items), dtype=torch.float32, device=dev)
        mu = logvar = None
        if self.variant == "vae" and want_latent:
            mu = torch.empty((b.batch, self.latent), dtype=torch.float32, device=dev)
            logvar = torch.empty_like(mu)
        st = self._step(seed, offset, mask, noise)
        check(lib().rtx_engine_forward(self.handle, C.byref(b), int(training), C.byref(st), int(remove_train),
                                       _ptr(logits), _ptr(mu), _ptr(logvar), stream_ptr()))
        return logits, mu, logvar

    def evaluate_topk(self, tr, te, rows, offsets, ks, scratch=None):
        """``rtx_engine_evaluate_topk``: every batch ``rows[offsets[i]:offsets[i + 1]]`` (device int32 row numbers into the resident
        :class:`CsrMatrix` pair ``tr`` / ``te``) scored in eval mode with the train items at -inf and reduced to nDCG@k / Recall@k, all
        enqueued by ONE call.  Returns float64 device tensors ``(ndcg, recall)`` of shape ``[len(ks), len(rows)]``."""
        offs = (C.c_int64 * len(offsets))(*[int(o) for o in offsets])
        ks = [int(k) for k in ks]
        arr = (C.c_int32 * len(ks))(*ks)
        total = int(offsets[-1] - offsets[0])
        bmax = max(int(b) - int(a) for a, b in zip(offsets[:-1], offsets[1:])) if len(offsets) > 1 else 0
        dev = rows.device
        if scratch is None or scratch.shape[0] < bmax:
            scratch = torch.empty((max(bmax, 1), self.n_items), dtype=torch.float32, device=dev)
        ndcg = torch.empty((len(ks), total), dtype=torch.float64, device=dev)
        recall = torch.empty_like(ndcg)
        check(lib().rtx_engine_evaluate_topk(self.handle, tr.handle, te.handle, _ptr(rows), offs, len(offsets) - 1, arr, len(ks),
                                             _ptr(scratch), _ptr(ndcg), _ptr(recall), stream_ptr()))
        return ndcg, recall

    def encode(self, x, training=False, seed=0, offset=0, mask=None):
        keep = []
        b = make_batch(x, keep=keep, n_items=self.n_items, n_in=self.n_in)
        dev = torch.device("cuda", torch.cuda.current_device())
        o0 = torch.empty((b.batch, self.latent), dtype=torch.float32, device=dev)
        o1 = torch.empty_like(o0) if self.variant == "vae" else None
        st = self._step(seed, offset, mask, None)
        check(lib().rtx_engine_encode(self.handle, C.byref(b), int(training), C.byref(st), _ptr(o0), _ptr(o1), stream_ptr()))
        return o0, o1

    def decode(self, z):
        _check_width(z, self.latent, "z")
        z = z.contiguous().float()
        logits = torch.empty((z.shape[0], self.n_items), dtype=torch.float32, device=z.device)
        check(lib().rtx_engine_decode(self.handle, _ptr(z), z.shape[0], _ptr(logits), stream_ptr()))
        return logits

    # ---- training ---------------------------------------------------------------------------------
    def loss_grads(self, x, target, step, loss_out, loss_accum=None, layer_cb=None):
        keep = []
        b = make_batch(x, target, keep=keep, n_items=self.n_items, n_in=self.n_in)
        cb = _lib.LAYER_CB(layer_cb) if layer_cb is not None else _lib.LAYER_CB()
        check(lib().rtx_engine_loss_grads(self.handle, C.byref(b), C.byref(step), _ptr(loss_out), _ptr(loss_accum),
                                          cb, None, stream_ptr()))

    def bind_grads16(self, ptrs):
        """device addresses of the bf16 gradient images, one per tensor (None unbinds): steps flagged RTX_STEP_GRADS_BF16 write
        their gradients there instead of into the float32 buffers"""
        if ptrs is None:
            check(lib().rtx_engine_bind_grads16(self.handle, None))
            self._g16_key = None
            return
        key = tuple(int(p) for p in ptrs)
        if getattr(self, "_g16_key", None) != key:
            check(lib().rtx_engine_bind_grads16(self.handle, (C.c_void_p * len(key))(*key)))
            self._g16_key = key

    def apply_adam(self, step):
        check(lib().rtx_engine_apply_adam(self.handle, C.byref(step), stream_ptr()))

    def apply_adam_layers(self, step, layer_lo, layer_hi, grads_bf16=None):
        """Adam + shadow refresh of layers [layer_lo, layer_hi) on the CURRENT stream; ``grads_bf16``: list of device
        addresses (one per bound tensor) of the bf16 image of the reduced gradients, or None."""
        arr = None
        if grads_bf16 is not None:
            arr = (C.c_void_p * self.n_tensors)(*[C.c_void_p(int(a)) for a in grads_bf16])
        check(lib().rtx_engine_apply_adam_layers(self.handle, C.byref(step), int(layer_lo), int(layer_hi), arr, stream_ptr()))

    def apply_adam_rows(self, step, layer, row_lo, row_hi, with_bias=True, w_grad_bf16=None, b_grad_bf16=None):
        """sharded optimizer: Adam on rows [row_lo, row_hi) of layer ``layer``'s weight matrix (+ its replicated bias) on the
        CURRENT stream; ``*_grad_bf16``: device addresses of the bf16 images of the reduced gradients, or None."""
        check(lib().rtx_engine_apply_adam_rows(self.handle, C.byref(step), int(layer), int(row_lo), int(row_hi), int(bool(with_bias)),
                                               None if w_grad_bf16 is None else C.c_void_p(int(w_grad_bf16)),
                                               None if b_grad_bf16 is None else C.c_void_p(int(b_grad_bf16)), stream_ptr()))

    def shadow_tensor(self, layer):
        """the compute copy of layer ``layer``'s weight matrix as a torch tensor ALIASING the engine's buffer
        ([padded_rows, ld], bfloat16 or float32): what a sharded-optimizer step all-gathers in place."""
        base, rows, ld, eb = C.c_void_p(), C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().rtx_engine_shadow_region(self.handle, int(layer), C.byref(base), C.byref(rows), C.byref(ld), C.byref(eb)))

        class _Alias:     # __cuda_array_interface__ is honoured by PyTorch-ROCm as well
            pass
        al = _Alias()
        n = rows.value * ld.value
        al.__cuda_array_interface__ = {"shape": (n,), "typestr": "<i2" if eb.value == 2 else "<f4", "data": (base.value, False), "version": 2}
        t = torch.as_tensor(al, device="cuda")
        if eb.value == 2:
            t = t.view(torch.bfloat16)
        return t.view(rows.value, ld.value)

    def train_step(self, x, target, step, loss_out, loss_accum=None):
        keep = []
        b = make_batch(x, target, keep=keep, n_items=self.n_items, n_in=self.n_in)
        check(lib().rtx_engine_train_step(self.handle, C.byref(b), C.byref(step), _ptr(loss_out), _ptr(loss_accum),
                                          stream_ptr()))

    def set_next_batch(self, x, target=None, seed=0, offset=0):
        """announce the batch (a :class:`RowBatch`) and the dropout seed of the training step AFTER the next ``train_step`` call:
        that call gathers it on the engine's side stream under its last weight kernel (``rtx_engine_set_next_batch``; a hint).
        Returns False (and announces nothing) for batches the engine cannot prefetch (dense tensors)."""
        if x is None:
            check(lib().rtx_engine_set_next_batch(self.handle, None, None))
            self._next_keep = None
            return False
        if not isinstance(x, RowBatch) or (target is not None and not isinstance(target, RowBatch)):
            return False
        b = make_batch(x, target, keep=None, n_items=self.n_items, n_in=self.n_in)
        if not b.csr or b.x_dense or b.target_dense:
            return False
        st = self._step(seed=seed, offset=offset)
        check(lib().rtx_engine_set_next_batch(self.handle, C.byref(b), C.byref(st)))
        self._next_keep = (x, target)         # the row ids stay alive (and unchanged) until that step has run
        return True

    def join(self):
        """resolve a join left open by a training step flagged ``RTX_STEP_DEFER_JOIN``: the current stream continues only after
        everything that step put on the engine's side stream (no-op when nothing is open)"""
        check(lib().rtx_engine_join(self.handle, stream_ptr()))

    def loss_mailbox(self, enable=True):
        """every training step also reports {loss, step count} to coherent host memory (``wait_loss``)"""
        check(lib().rtx_engine_loss_mailbox(self.handle, int(bool(enable))))
        self._mailbox = bool(enable)

    def wait_loss(self, step_count, timeout_s=0.0):
        """the loss of the training step whose Adam step count is ``step_count``: spins on the host mailbox, no stream drain
        (reference models.py:835 ``return loss.item()``)"""
        out = C.c_float()
        check(lib().rtx_engine_wait_loss(self.handle, int(step_count), C.byref(out), float(timeout_s)))
        return out.value

    # ---- data parallel, scheduled by the engine (rtx_engine_dp_attach / rtx_engine_train_step_dp) --------------------
    def dp_attach(self, plan):
        """``plan``: a :class:`rectorch_amd.parallel.NativePlan` (rank, world, sharded, comm dtype and ONE of an RCCL
        communicator, caller-supplied collectives, or emulation), or None to detach."""
        if plan is None:
            check(lib().rtx_engine_dp_attach(self.handle, None))
            old, self._dp_plan, self._dp_any_sharded = getattr(self, "_dp_plan", None), None, False
            if old is not None and hasattr(old, "_forget"):
                old._forget(self)
            return
        cfg = plan.c_cfg()            # (the sharding threshold travels in the cfg: nothing of an earlier plan stays in the engine)
        old = getattr(self, "_dp_plan", None)
        check(lib().rtx_engine_dp_attach(self.handle, C.byref(cfg)))     # (the C side releases an earlier attachment first)
        if old is not None and old is not plan and hasattr(old, "_forget"):
            old._forget(self)         # the old plan's close() must not detach this engine from its new plan
        self._dp_plan = plan          # keeps the communicator / callback objects alive as long as the engine uses them
        # what the ENGINE decided to shard (a plan that asks for the sharded optimizer on a network of small matrices shards nothing)
        self._dp_any_sharded = any(self.dp_owned_rows(l)[2] for l in range(self.n_tensors // 2))
        if hasattr(plan, "_remember"):
            plan._remember(self)

    def train_step_dp(self, x, target, step, loss_out, loss_accum=None):
        """one rank's train_batch of a data-parallel job: forward + loss + backward on this rank's users, the gradient exchange
        and the optimizer in ONE call (``step.inv_batch`` = 1 / global batch)"""
        keep = []
        b = make_batch(x, target, keep=keep, n_items=self.n_items, n_in=self.n_in)
        check(lib().rtx_engine_train_step_dp(self.handle, C.byref(b), C.byref(step), _ptr(loss_out), _ptr(loss_accum), stream_ptr()))

    def dp_owned_rows(self, layer):
        """(row_lo, row_hi, sharded) of layer ``layer``'s weight matrix: the rows whose float32 master and Adam moments this rank
        keeps current under the sharded optimizer"""
        lo, hi, sh = C.c_int32(), C.c_int32(), C.c_int32()
        check(lib().rtx_engine_dp_owned_rows(self.handle, int(layer), C.byref(lo), C.byref(hi), C.byref(sh)))
        return lo.value, hi.value, bool(sh.value)

    # ---- instrumentation --------------------------------------------------------------------------
    def set_option(self, key, value):
        """measurement knobs of the engine ("fuse_adam", "two_stream", "lse_fuse", "dw_cfg", "splitk", ...; include/rectorch_hip.h)"""
        check(lib().rtx_engine_set_option(self.handle, key.encode(), int(value)))

    def get_option(self, key):
        v = C.c_int32()
        check(lib().rtx_engine_get_option(self.handle, key.encode(), C.byref(v)))
        return v.value

    def set_timing(self, site=None, enable=True):
        check(lib().rtx_engine_set_timing(self.handle, None if site is None else site.encode(), int(enable)))

    def get_timings(self):
        cap = 64
        names = (C.c_char * 48 * cap)()
        ms = (C.c_float * cap)()
        cnt = (C.c_int32 * cap)()
        n = C.c_int32()
        check(lib().rtx_engine_get_timings(self.handle, cap, names, ms, cnt, C.byref(n)))
        return {names[i].value.decode(): (float(ms[i]), int(cnt[i])) for i in range(n.value)}

    def step_cost(self, batch):
        b, f = C.c_double(), C.c_double()
        check(lib().rtx_engine_step_cost(self.handle, int(batch), C.byref(b), C.byref(f)))
        return b.value, f.value

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                lib().rtx_engine_destroy(h)
            except Exception:
                pass
            self.handle = None


def multinomial_loss(recon, x, mu=None, logvar=None, beta=0.0):
    """``-mean_b sum_i log_softmax(recon)_bi x_bi + beta * KLD`` as a 0-dim device tensor
    (reference MultiVAE.loss_function, rectorch/models.py:813-815)."""
    _lib.require_gpu()
    dev = torch.device("cuda", torch.cuda.current_device())
    recon = recon.to(dev, torch.float32).contiguous()
    x = x.to(dev, torch.float32).contiguous()
    if mu is not None:
        mu = mu.to(dev, torch.float32).contiguous()
        logvar = logvar.to(dev, torch.float32).contiguous()
    out = torch.empty((), dtype=torch.float32, device=dev)
    check(lib().rtx_multinomial_loss(_ptr(recon), _ptr(x), recon.shape[0], recon.shape[1], _ptr(mu), _ptr(logvar),
                                     0 if mu is None else mu.shape[1], float(beta), _ptr(out), stream_ptr()))
    return out


def sum_l2_norms(tensors):
    """``sum_t ||tensor_t||_2`` as a 0-dim device tensor (the regulariser of MultiDAE.loss_function,
    reference rectorch/models.py:702-706)."""
    _lib.require_gpu()
    ts = [t.contiguous() for t in tensors]
    n = len(ts)
    ptrs = (C.c_void_p * n)(*[t.data_ptr() for t in ts])
    sizes = (C.c_int64 * n)(*[t.numel() for t in ts])
    out = torch.empty((), dtype=torch.float32, device=ts[0].device)
    check(lib().rtx_sum_l2_norms(ptrs, sizes, n, _ptr(out), stream_ptr()))
    return out


def topk_metrics(scores, heldout, rows, ks, want_topk=False, out=None):
    """nDCG@k and Recall@k for every k in ``ks`` (reference rectorch/metrics.py:136-147, 187-196) computed on the
    device from a score tensor ``[B, n_items]`` and the users' held-out rows of a resident :class:`CsrMatrix`.
    Returns ``(ndcg [len(ks), B], recall [len(ks), B])`` as float64 device tensors (+ the sorted top-k item ids).
    ``out``: a pair of contiguous float64 ``[len(ks), B]`` tensors to write into (no allocation per batch)."""
    _lib.require_gpu()
    scores = scores.contiguous()
    B, n_items = scores.shape
    ks = [int(k) for k in ks]
    arr = (C.c_int32 * len(ks))(*ks)
    if out is not None:
        ndcg, recall = out
        assert ndcg.shape == (len(ks), B) and recall.shape == (len(ks), B) and ndcg.is_contiguous() and recall.is_contiguous() and ndcg.dtype == torch.float64
    else:
        ndcg = torch.empty((len(ks), B), dtype=torch.float64, device=scores.device)
        recall = torch.empty_like(ndcg)
    kmax = min(max(ks), n_items)
    topk = torch.empty((B, kmax), dtype=torch.int32, device=scores.device) if want_topk else None
    check(lib().rtx_topk_metrics(_ptr(scores), n_items, B, n_items, heldout.handle, _ptr(rows), arr, len(ks),
                                 _ptr(ndcg), _ptr(recall), _ptr(topk), 0, stream_ptr()))
    return (ndcg, recall, topk) if want_topk else (ndcg, recall)


class EaseSolver:
    """Device-resident EASE model: the item-item matrix ``B`` of reference ``EASE.train`` (rectorch/models.py:
    1015-1024) computed by ``rtx_ease_fit`` (Gram matrix, f64 Cholesky, inverse -- all MFMA) and kept in HBM."""

    def __init__(self, train, lam):
        _lib.require_gpu()
        self.train = train if isinstance(train, CsrMatrix) else CsrMatrix(train)
        self.lam = float(lam)
        h = C.c_void_p()
        check(lib().rtx_ease_fit(self.train.handle, C.c_double(self.lam), C.byref(h), stream_ptr()))
        self.handle = h
        self.n_items = int(self.train.shape[1])

    def timings(self):
        """HIP-event durations (ms) of the fit: total, Gram matrix, Cholesky, inverse + P = W^T W."""
        v = [C.c_double() for _ in range(4)]
        check(lib().rtx_ease_timings(self.handle, *[C.byref(x) for x in v]))
        return dict(zip(("fit_ms", "gram_ms", "chol_ms", "inv_ms"), (x.value for x in v)))

    def weights(self):
        """``B`` as a float64 device tensor [n_items, n_items] (a copy)."""
        out = torch.empty((self.n_items, self.n_items), dtype=torch.float64, device="cuda")
        check(lib().rtx_ease_copy_weights(self.handle, _ptr(out), stream_ptr()))
        return out

    def scores(self, row_ids, mask=None, out=None):
        """``(X B)[row_ids]`` (reference models.py:1025 + 1054) as a float64 device tensor, ``-inf`` at the non-zero
        entries of ``mask`` (a :class:`CsrMatrix` whose row b belongs to ``row_ids[b]``; models.py:1055-1056)."""
        row_ids = torch.as_tensor(row_ids, dtype=torch.int32).to("cuda").contiguous()
        n = int(row_ids.numel())
        if n and (int(row_ids.min()) < 0 or int(row_ids.max()) >= self.train.shape[0]):
            raise IndexError("user index out of range for the training matrix (%d users)" % self.train.shape[0])
        if mask is not None and (mask.shape[0] != n or mask.shape[1] != self.n_items):
            raise ValueError("mask matrix has shape %s, expected (%d, %d)" % (mask.shape, n, self.n_items))
        if out is None:
            out = torch.empty((n, self.n_items), dtype=torch.float64, device="cuda")
        if n:
            check(lib().rtx_ease_scores(self.handle, self.train.handle, _ptr(row_ids), n,
                                        None if mask is None else mask.handle, None, _ptr(out), stream_ptr()))
        return out

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                lib().rtx_ease_destroy(h)
            except Exception:
                pass
            self.handle = None


def cast_f32_bf16(src, dst):
    """``dst`` (bfloat16) = round-to-nearest-even of ``src`` (float32), same number of elements, on the current stream."""
    assert src.dtype == torch.float32 and dst.dtype == torch.bfloat16 and src.numel() == dst.numel()
    check(lib().rtx_cast_f32_bf16(_ptr(src), _ptr(dst), src.numel(), stream_ptr()))


class SvaeTarget:
    """Compact loss target of one SVAE sequence: the multi-hot rows of the reference's ``y_batch_s`` [1, T, n_items]
    (samplers.py:539-562) as a CSR over the T steps, plus the likelihood normaliser ``d`` the reference's training path
    ends up with (the ones of the FIRST step: ``train_batch`` flattens the target before ``loss_function`` sums
    ``x[0, :n_items]``, models.py:822 + 1623)."""
    __slots__ = ("indptr", "indices", "d", "n_steps")

    def __init__(self, rows, device="cuda"):
        """``rows``: list (one per time step) of distinct item ids"""
        ptr = np.zeros(len(rows) + 1, dtype=np.int64)
        for t, r in enumerate(rows):
            ptr[t + 1] = ptr[t] + len(r)
        idx = np.fromiter((i for r in rows for i in r), dtype=np.int32, count=int(ptr[-1]))
        self.indptr = torch.from_numpy(ptr).to(device)
        self.indices = torch.from_numpy(idx).to(device)
        self.d = float(len(rows[0])) if rows else 0.0
        self.n_steps = len(rows)


class SvaePack:
    """Several user sequences for ONE optimizer step (``SVAE_Sampler(pack=N)``; not in the reference): the input items of all
    users concatenated on the device, the row range of every user, and the targets of all time steps as one CSR.

    ``seqs``: list of item-id lists (the model inputs, i.e. all but each user's last item); ``target_rows``: per user the list
    (one entry per time step) of distinct target items.  ``d`` keeps, per user, the likelihood normaliser the reference's
    training path uses (the ones of the user's FIRST step, see :class:`SvaeTarget`)."""
    __slots__ = ("items", "seq_ptr", "lens", "indptr", "indices", "d", "n_steps", "users")

    def __init__(self, seqs, target_rows, users=None, device="cuda"):
        assert len(seqs) == len(target_rows) and len(seqs) >= 1
        self.lens = [len(q) for q in seqs]
        for q, rows in zip(seqs, target_rows):
            assert len(q) == len(rows) and len(q) >= 1, "every sequence needs one target row per time step"
        sp = np.zeros(len(seqs) + 1, dtype=np.int32)
        sp[1:] = np.cumsum(self.lens)
        self.n_steps = int(sp[-1])
        self.items = torch.from_numpy(np.fromiter((i for q in seqs for i in q), dtype=np.int32, count=self.n_steps)).to(device)
        self.seq_ptr = torch.from_numpy(sp).to(device)
        ptr = np.zeros(self.n_steps + 1, dtype=np.int64)
        ptr[1:] = np.cumsum(np.fromiter((len(r) for rows in target_rows for r in rows), dtype=np.int64, count=self.n_steps))
        self.indptr = torch.from_numpy(ptr).to(device)
        self.indices = torch.from_numpy(np.fromiter((i for rows in target_rows for r in rows for i in r), dtype=np.int32,
                                                    count=int(ptr[-1]))).to(device)
        self.d = [float(len(rows[0])) for rows in target_rows]
        self.users = list(users) if users is not None else None

    def __len__(self):
        return len(self.lens)

    def row_scales(self, beta):
        """per-row loss factors of the pack's step: (1 / (d_u N), beta / (T_u N)) repeated over each user's rows, as ONE device
        tensor [2, n_steps] (numpy on the host: torch.repeat_interleave on CPU tensors costs milliseconds)"""
        n = len(self.lens)
        lens = np.asarray(self.lens)
        nll = np.array([1.0 / (d * n) if d != 0 else np.inf for d in self.d], dtype=np.float32)
        kl = (np.float32(beta) / (lens * n)).astype(np.float32)
        both = np.stack([np.repeat(nll, lens), np.repeat(kl, lens)])
        return torch.from_numpy(both).to(self.items.device)


class SvaeEngine:
    """One ``rtx_svae``: the SVAE network's compute state (embedding -> GRU -> VAE head -> decoder, float32), bound to
    the network's parameters and, for training, to gradient buffers and the Adam moments (see :class:`Engine`)."""

    def __init__(self, n_items, embed_size, rnn_size, enc_dims, dec_dims, max_len=256):
        _lib.require_gpu()
        cfg = _lib.SvaeCfg()
        cfg.n_items, cfg.embed_size, cfg.rnn_size = int(n_items), int(embed_size), int(rnn_size)
        if len(enc_dims) - 1 > _lib.MAX_LAYERS or len(dec_dims) - 1 > _lib.MAX_LAYERS:
            raise _lib.RtxError("at most %d layers per encoder/decoder are supported" % _lib.MAX_LAYERS)
        cfg.n_enc, cfg.n_dec = len(enc_dims) - 1, len(dec_dims) - 1
        for i, d in enumerate(enc_dims):
            cfg.enc_dims[i] = int(d)
        for i, d in enumerate(dec_dims):
            cfg.dec_dims[i] = int(d)
        cfg.max_len = int(max_len)
        self.max_len, self.n_items, self.latent = int(max_len), int(n_items), int(enc_dims[-1])
        h = C.c_void_p()
        check(lib().rtx_svae_create(C.byref(cfg), C.byref(h)))
        self.handle = h
        self.n_tensors = lib().rtx_svae_n_tensors(h)
        self._keep = []

    def loss_mailbox(self, enable=True):
        """``rtx_svae_loss_mailbox``: every training step also reports its loss to coherent host memory as soon as it is final"""
        check(lib().rtx_svae_loss_mailbox(self.handle, int(bool(enable))))
        self._mailbox = bool(enable)

    def wait_loss(self, timeout_s=60.0):
        """the loss of the LAST step enqueued, without draining the stream behind it (``rtx_svae_wait_loss``)"""
        out = C.c_float()
        check(lib().rtx_svae_wait_loss(self.handle, C.byref(out), float(timeout_s)))
        return float(out.value)

    def set_option(self, key, value):
        """``"gemm_bf16"``: bf16 operands / float32 accumulate in every matrix product of the model (include/rectorch_hip.h)"""
        check(lib().rtx_svae_set_option(self.handle, key.encode(), int(value)))

    def bind(self, params, grads=None, exp_avg=None, exp_avg_sq=None):
        n = self.n_tensors
        assert len(params) == n, "expected %d parameter tensors, got %d" % (n, len(params))
        rows, cols = C.c_int32(), C.c_int32()
        for t, p in enumerate(params):
            check(lib().rtx_svae_tensor_shape(self.handle, t, C.byref(rows), C.byref(cols)))
            want = (rows.value,) if cols.value == 1 and p.dim() == 1 else (rows.value, cols.value)
            if tuple(p.shape) != want or p.dtype != torch.float32 or not p.is_cuda or not p.is_contiguous():
                raise _lib.RtxError("parameter %d must be a contiguous float32 device tensor of shape %s, got %s %s on %s"
                                    % (t, want, tuple(p.shape), p.dtype, p.device))

        def arr(ts):
            if ts is None:
                return None
            return (C.c_void_p * n)(*[t.data_ptr() for t in ts])
        self._keep = [params, grads, exp_avg, exp_avg_sq]
        check(lib().rtx_svae_bind(self.handle, arr(params), arr(grads), arr(exp_avg), arr(exp_avg_sq)))

    @staticmethod
    def _items(x):
        return x.reshape(-1).to("cuda", torch.int32).contiguous()

    def forward(self, x, noise=None, seed=0, remove_train=False, want_all=True):
        """SVAE_net.forward on one sequence: (logits [T, n_items] or None, last-step logits [n_items], mu, logvar)."""
        items = self._items(x)
        T = int(items.numel())
        dev = items.device
        la = torch.empty((T, self.n_items), dtype=torch.float32, device=dev) if want_all else None
        ll = torch.empty((self.n_items,), dtype=torch.float32, device=dev)
        mu = torch.empty((T, self.latent), dtype=torch.float32, device=dev)
        lv = torch.empty_like(mu)
        if noise is not None:
            noise = noise.to(dev, torch.float32).contiguous()
        check(lib().rtx_svae_forward(self.handle, _ptr(items), T, _ptr(noise), C.c_uint64(int(seed) & (2 ** 64 - 1)), C.c_uint64(0),
                                     int(remove_train), _ptr(la), _ptr(ll), _ptr(mu), _ptr(lv), stream_ptr()))
        return la, ll, mu, lv

    def train_step(self, x, target, step, loss_out, loss_accum=None):
        items = self._items(x)
        T = int(items.numel())
        if isinstance(target, SvaeTarget):
            assert target.n_steps == T, "the target has %d steps, the sequence %d" % (target.n_steps, T)
            ptr, idx, dense = target.indptr, target.indices, None
        else:
            dense = target.reshape(T, -1).to(items.device, torch.float32).contiguous()
            if dense.shape[1] != self.n_items:
                raise _lib.RtxError("the target must be [1, T, %d], got %s" % (self.n_items, tuple(target.shape)))
            ptr = idx = None
        check(lib().rtx_svae_train_step(self.handle, _ptr(items), T, _ptr(ptr), _ptr(idx), _ptr(dense), C.byref(step), _ptr(loss_out),
                                        _ptr(loss_accum), stream_ptr()))

    def train_pack(self, pack, step, beta, loss_out, loss_accum=None):
        """one optimizer step on the users of ``pack`` (:class:`SvaePack`): mean over the pack of the per-user SVAE loss"""
        if pack.n_steps > self.max_len:
            raise _lib.RtxError("the pack holds %d time steps, the engine was sized for %d" % (pack.n_steps, self.max_len))
        sc = pack.row_scales(beta)
        check(lib().rtx_svae_train_pack(self.handle, _ptr(pack.items), pack.n_steps, _ptr(pack.seq_ptr), len(pack), _ptr(sc[0]), _ptr(sc[1]),
                                        _ptr(pack.indptr), _ptr(pack.indices), C.byref(step), _ptr(loss_out), _ptr(loss_accum), stream_ptr()))

    def __del__(self):
        h = getattr(self, "handle", None)
        if h is not None and h.value:
            try:
                lib().rtx_svae_destroy(h)
            except Exception:
                pass
            self.handle = None
