r"""Trainer classes of the Mult-VAE / Mult-DAE path with the reference's API (rectorch/models.py), running
on the MI355X engine.

Same class hierarchy and signatures as the reference: ``RecSysModel`` (models.py:70-161) ->
``TorchNNTrainer`` (:164-322) -> ``AETrainer`` (:325-516) -> ``VAE`` (:519-625) -> ``MultiVAE`` (:709-908), and
``MultiDAE`` (:628-706).  What changes is underneath: ``train_batch`` is ONE call into librectorch_hip
(gather -> forward -> multinomial/KL loss -> backward -> fused Adam, all hand-written HIP), ``predict`` is the
HIP forward, and ``train_epoch`` feeds row numbers of a device-resident CSR instead of dense host batches and
reads the loss back only at the logging boundaries.  ``model.optimizer`` is still a ``torch.optim.Adam``
object whose ``state`` tensors ARE the buffers the fused Adam kernel updates, so checkpoints round-trip with the
reference's layout (``epoch``, ``state_dict``, ``optimizer``, ``gradient_updates``).
"""
import logging
import os
import time

import numpy as np
import torch
import torch.optim as optim

from . import _lib
from .engine import CsrMatrix, EaseSolver, RowBatch, SvaePack, SvaeTarget, multinomial_loss, tagged_rows
from .evaluation import ValidFunc, evaluate
from .samplers import DataSampler

__all__ = ['RecSysModel', 'TorchNNTrainer', 'AETrainer', 'VAE', 'MultiVAE', 'MultiDAE', 'CMultiVAE', 'EASE', 'SVAE']

logger = logging.getLogger(__name__)


class RecSysModel():
    r"""Abstract base class that any Recommendation model must inherit from (reference models.py:70-161)."""
    def train(self, train_data, **kwargs):
        raise NotImplementedError()

    def predict(self, x, *args, **kwargs):
        raise NotImplementedError()

    def save_model(self, filepath, *args, **kwargs):
        raise NotImplementedError()

    def load_model(self, filepath, *args, **kwargs):
        raise NotImplementedError()


class _RtxState:
    """Private per-trainer state of the HIP path (kept in one object with a short repr because the
    reference's ``__str__`` prints every attribute of the model)."""
    def __init__(self):
        self.flat_grads = None
        self.grads = None
        self.layer_ranges = None
        self.tensor_offsets = None
        self.adam_step = 0
        self.loss_buf = None      # [0] = last loss, [1] = running sum since the last read-back
        self.pending_seed = None  # the dropout seed drawn one step early for a batch announced to the engine
        self.reducer = None       # data parallel: rectorch_amd.parallel.GradAllReducer
        self.masters_sharded = False   # sharded optimizer: the float32 master rows of other ranks are stale until gathered
        self.inject = None        # parity tests: (dropout keep-mask, eps) captured from the reference's RNG

    def __repr__(self):
        return "<MI355X engine state>"


class TorchNNTrainer(RecSysModel):
    r"""Abstract class representing a neural network-based model (reference models.py:164-322).

    Parameters
    ----------
    net : :class:`torch.nn.Module`
        The neural network architecture (a :mod:`rectorch_amd.nets` network).
    learning_rate : :obj:`float` [optional]
        The learning rate for the optimizer, by default 1e-3.

    Attributes
    ----------
    network, learning_rate, optimizer, device : as in the reference.  ``device`` is the HIP device: a
        network still on the host is moved to the MI355X at construction (this package has no CPU compute
        path); without any HIP device the object can be built, saved and loaded but not run.
    """
    def __init__(self, net, learning_rate=1e-3):
        self.network = net
        self.learning_rate = learning_rate
        self.optimizer = None #to be initialized in the sub-classes

        if not next(self.network.parameters()).is_cuda and torch.cuda.is_available():
            logger.info("moving the network to the MI355X (rectorch_amd computes only on the HIP device)")
            self.network.to(torch.device("cuda"))
        if next(self.network.parameters()).is_cuda:
            self.device = torch.device("cuda")
        else:
            self.device = torch.device("cpu")

    def loss_function(self, prediction, ground_truth, *args, **kwargs):
        raise NotImplementedError()

    def train(self, train_data, *args, **kwargs):
        raise NotImplementedError()

    def train_epoch(self, epoch, train_data, *args, **kwargs):
        raise NotImplementedError()

    def train_batch(self, epoch, tr_batch, te_batch, *args, **kwargs):
        raise NotImplementedError()

    def predict(self, x, *args, **kwargs):
        raise NotImplementedError()

    def __str__(self):
        # "ClassName(\n  attr = value,\n  ...\n)" with multi-line values indented under their attribute -- the layout of
        # the reference's repr (models.py:313-322), which its users read in logs
        fields = []
        for name, value in vars(self).items():
            first, *rest = str(value).split("\n")
            fields.append("  %s = %s" % (name, "\n".join([first] + ["  " + ln for ln in rest])))
        return "%s(\n%s\n)" % (type(self).__name__, ",\n".join(fields))

    __repr__ = __str__


def _with_next(it):
    """(item, next item or None) over an iterator: one batch of look-ahead for the engine's prefetch"""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class _EpochLog:
    """The reference's progress lines for one epoch (models.py:401-422): every ``max(10, n_batches // 10**verbose)``
    batches ``| epoch e | i/n batches | ms/batch t | loss l |`` with the mean loss of that stretch, at the end
    ``| epoch e | loss L | total time: Ts |`` with the mean over all batches.  One implementation for every trainer."""
    def __init__(self, epoch, n_batches, verbose):
        self.epoch, self.n_batches = epoch, n_batches
        self.every = max(10, n_batches // 10 ** verbose)
        self.t_epoch = self.t_stretch = time.time()
        self.total = 0.0
        self.seen = 0           # batches actually consumed (a sampler's __len__ may be approximate: SVAE packs under shuffle)

    def due(self, n_done):
        return n_done % self.every == 0

    def stretch(self, n_done, loss_sum):
        """close a stretch of ``every`` batches whose losses sum to ``loss_sum``"""
        now = time.time()
        logger.info('| epoch %d | %d/%d batches | ms/batch %.2f | loss %.2f |', self.epoch, n_done, self.n_batches,
                    (now - self.t_stretch) * 1000 / self.every, loss_sum / self.every)
        self.total += loss_sum
        self.seen = n_done
        self.t_stretch = time.time()

    def finish(self, tail_loss_sum, n_done=None):
        n = n_done if n_done is not None else self.n_batches
        logger.info("| epoch %d | loss %.4f | total time: %.2fs |", self.epoch,
                    (self.total + tail_loss_sum) / max(n, 1), time.time() - self.t_epoch)


def _validate(model, epoch, valid_data, valid_metric, valid_func):
    """one validation pass + the reference's log line; returns the mean of the metric (models.py:389-398, 881-889)"""
    assert valid_metric is not None, \
        "In case of validation 'valid_metric' must be provided"
    per_user = valid_func(model, valid_data, valid_metric)
    mean = np.mean(per_user)
    logger.info('| epoch %d | %s %.3f (%.4f) |', epoch, valid_metric, mean, np.std(per_user) / np.sqrt(len(per_user)))
    return mean


class AETrainer(TorchNNTrainer):
    r"""Base class for Autoencoder-based models (reference models.py:325-516).  Holds the Adam optimizer, the
    epoch loop, prediction and checkpointing shared by :class:`MultiDAE` and :class:`MultiVAE`.

    Extra keyword arguments (not in the reference): ``numerics`` = arithmetic of the training step
    ("bf16": bf16 MFMA operands, f32 accumulation, f32 master weights and Adam; "fp32": exact-f32 MFMA) and
    ``predict_numerics`` (default "fp32", the parity mode: logits within 1e-5 of the reference's CPU path).
    """
    _variant = "dae"
    _uses_te = False     # train_batch ignores te_batch (reference models.py:444-445)

    def __init__(self, ae_net, learning_rate=1e-3, numerics="bf16", predict_numerics="fp32"):
        super(AETrainer, self).__init__(ae_net, learning_rate)
        self.optimizer = optim.Adam(self.network.parameters(), lr=learning_rate)
        self.numerics = numerics
        self.predict_numerics = predict_numerics
        # the single-GPU bf16 step runs Adam inside the weight-gradient kernels (dw_adam.hip), so the gradients never
        # reach HBM and p.grad is NOT refreshed; set True to also store them (4 B/param).  float32 numerics and the
        # data-parallel step always store them.
        self.keep_grads = False
        # train_batch returns THIS step's loss as a float (reference models.py:835).  True: the float comes from the engine's host
        # mailbox (rtx_engine_wait_loss: the host spins on a step count in coherent host memory while the step's remaining
        # kernels keep running); False: ``loss.item()``, which drains the stream like the reference does.
        self.loss_mailbox = True
        # train_epoch / callers of _fused_step(next_x=...) announce the next batch to the engine, which gathers it on its side
        # stream under the current step's last weight kernel (rtx_engine_set_next_batch); False: every step gathers for itself
        self.prefetch_batches = True
        self._rtx = _RtxState()

    # ------------------------------------------------------------------------------------------ loss
    def loss_function(self, prediction, ground_truth):
        r"""The reference's vanilla autoencoder uses an MSE loss here (models.py:347-377); that model is not
        part of the Mult-VAE / Mult-DAE path."""
        raise NotImplementedError("the MSE autoencoder is outside the MI355X hot path; use MultiDAE / MultiVAE")

    # ------------------------------------------------------------------------------------- training
    def train(self,
              train_data,
              valid_data=None,
              valid_metric=None,
              valid_func=ValidFunc(evaluate),
              num_epochs=100,
              verbose=1):
        r"""Training of a neural network-based model (reference models.py:379-398): ``num_epochs`` epochs, each followed
        by a validation pass when ``valid_data`` is given; Ctrl-C ends training early with a warning."""
        try:
            for epoch in range(1, num_epochs + 1):
                self.train_epoch(epoch, train_data, verbose)
                if valid_data is not None:
                    self.consolidate()          # (data parallel, sharded optimizer: every rank runs train() and gets here)
                    _validate(self, epoch, valid_data, valid_metric, valid_func)
        except KeyboardInterrupt:
            logger.warning('Handled KeyboardInterrupt: exiting from training early')

    def train_epoch(self, epoch, train_loader, verbose=1):
        r"""Training of a single epoch (reference models.py:401-422): same progress lines.  With a device-resident
        :class:`DataSampler` the batches are row numbers, steps are enqueued back to back and the loss is read from the
        device only where a progress line needs it (the engine keeps a running sum)."""
        self.network.train()
        log = _EpochLog(epoch, len(train_loader), verbose)
        resident = isinstance(train_loader, DataSampler) and train_loader.resident
        pending = 0.0                       # host path: losses of the current stretch
        done = 0
        def shaped(item):
            return item if self._uses_te or item.te is None else RowBatch(item.tr, None, item.rows)
        for done, (item, nxt) in enumerate(_with_next(train_loader.iter_rows()) if resident else ((i, None) for i in train_loader), 1):
            if resident:
                # (the batch after this one is announced to the engine: it is gathered under this step's last weight kernel)
                # (no wait for the engine's side stream between two steps: the next step resolves the join in its first kernel)
                self._fused_step(shaped(item), None, want_loss=False, next_x=None if nxt is None else shaped(nxt), defer_join=True)
            else:
                data, gt = item
                pending += self.train_batch(data, gt)
            if log.due(done):
                log.stretch(done, self._read_loss_sum() if resident else pending)
                pending = 0.0
        log.finish(self._read_loss_sum() if resident else pending, done)

    def train_batch(self, tr_batch, te_batch=None):
        r"""Training of a single batch (reference models.py:424-447): the loss target is the batch itself
        (``te_batch`` is ignored, as in the reference).  Returns the loss as a Python float."""
        return self._fused_step(tr_batch, None, want_loss=True)

    # ---------------------------------------------------------------------------- the fused HIP step
    def _step_scalars(self):
        """(beta, lam) of this update; sub-classes fill them."""
        return 0.0, 0.0

    def _after_step(self):
        pass

    def _ensure_train_state(self):
        st = self._rtx
        params = self.network._param_list()
        if st.grads is None or st.flat_grads.device != params[0].device:
            # one flat gradient buffer (one RCCL all-reduce region per layer); p.grad are views into it
            # a weight matrix's region holds its rows padded to a multiple of 128 (zeros), so that a data-parallel
            # reduce-scatter can cut it into equal, row-aligned blocks for any power-of-two number of ranks
            offs, total = [], 0
            for p in params:
                offs.append(total)
                n = p.numel() if p.dim() == 1 else ((p.shape[0] + 1 + 127) // 128 * 128) * p.shape[1]
                total += (n + 63) // 64 * 64
            st.flat_grads = torch.zeros(total, dtype=torch.float32, device=params[0].device)
            st.grads = [st.flat_grads[o:o + p.numel()].view(p.shape) for o, p in zip(offs, params)]
            st.layer_ranges = [(offs[2 * l], offs[2 * l + 1] + params[2 * l + 1].numel()) for l in range(len(params) // 2)]
            st.tensor_offsets = offs
            for p, g in zip(params, st.grads):
                p.grad = g
            st.loss_buf = torch.zeros(2, dtype=torch.float32, device=params[0].device)
        first = True
        for p in params:
            state = self.optimizer.state[p]
            if len(state) == 0:
                # what torch.optim.Adam._init_group creates lazily on its first step()
                state['step'] = torch.tensor(0.0, dtype=torch.float32)
                state['exp_avg'] = torch.zeros_like(p, memory_format=torch.preserve_format)
                state['exp_avg_sq'] = torch.zeros_like(p, memory_format=torch.preserve_format)
            if first:
                st.adam_step = max(st.adam_step, int(float(state['step'])))
                first = False
        m = [self.optimizer.state[p]['exp_avg'] for p in params]
        v = [self.optimizer.state[p]['exp_avg_sq'] for p in params]
        return st, params, m, v

    def _fused_step(self, x, target, want_loss=True, next_x=None, next_target=None, defer_join=False):
        """``next_x`` (optional, a :class:`RowBatch`): the batch of the NEXT ``_fused_step`` call.  The engine gathers it on its
        side stream under this step's last weight kernel, so that step starts with the first-layer product (single GPU, bf16).
        Its dropout seed is drawn now and kept for that step: the sequence of draws from torch's generator is unchanged.
        ``defer_join`` (epoch loops only): the step does not make the stream wait for the engine's side stream at its end
        (``RTX_STEP_DEFER_JOIN``); the next step resolves the join inside its first kernel.  The caller must not read parameters,
        optimizer state or the loss buffers before its next ``_fused_step`` / engine call or ``self._join()``."""
        _lib.require_gpu()
        st, params, m, v = self._ensure_train_state()
        if not isinstance(x, RowBatch):
            x = self.network._as_input(x)
            if target is not None:
                target = self.network._as_input(target)
        B = len(x) if isinstance(x, RowBatch) else x.shape[0]
        eng = self.network.rtx_engine(self.numerics, B, train_buffers=(st.grads, m, v))
        g = self.optimizer.param_groups[0]
        beta, lam = self._step_scalars()
        st.adam_step += 1
        from .nets import draw_seed
        red = st.reducer
        native = red is not None and getattr(red, "native", False)
        # (python-driven reducer) data parallel, bf16 numerics and bf16 exchange with the optimizer behind each bucket: the weight-gradient kernels write
        # the bf16 images the all-reduce sends directly (no float32 gradient store, no cast pass); p.grad is not filled then
        direct16 = (red is not None and not native and self.numerics == "bf16" and red.flat16 is not None and red.on_device and
                    getattr(red, "bucket_adam", False) and getattr(red, "allow_direct16", True) and not self.keep_grads)
        if red is not None and not native:
            red.direct16 = direct16
            eng.bind_grads16(red.grads16_ptrs() if direct16 else None)
        inj = self._rtx.inject or (None, None)     # (keep-mask uint8 [B, n_items], eps [B, latent]) for parity tests
        seed = st.pending_seed if st.pending_seed is not None else draw_seed()     # (drawn one step early for an announced batch)
        st.pending_seed = None
        if (next_x is not None and (red is None or native) and self.numerics == "bf16" and self._rtx.inject is None and self.prefetch_batches
                and isinstance(next_x, RowBatch)):
            st.pending_seed = draw_seed()
            # (data parallel, engine-scheduled: the gather rides behind bucket A on the engine's side stream; the dropout stream of a rank is
            #  keyed by its rank, as the step itself keys it)
            eng.set_next_batch(next_x, next_target, seed=st.pending_seed, offset=0 if red is None else red.rank)
        step = eng._step(seed=seed, offset=0 if red is None else red.rank, mask=inj[0], noise=inj[1],
                         beta=float(beta), lam=float(lam),
                         # (a rank's slice made by parallel.shard_batch knows the global batch: no collective, no host sync)
                         inv_batch=1.0 / (B if red is None else (getattr(x, "global_len", None) or red.global_batch(B))),
                         lr=float(g['lr']), beta1=float(g['betas'][0]), beta2=float(g['betas'][1]),
                         eps=float(g['eps']), weight_decay=float(g['weight_decay']), step=st.adam_step,
                         flags=(_lib.RTX_STEP_KEEP_GRADS if self.keep_grads else 0) |
                               (_lib.RTX_STEP_DEFER_JOIN if defer_join and (red is None or native) and not want_loss and self.numerics == "bf16" else 0) |
                               (_lib.RTX_STEP_GRADS_BF16 if direct16 else 0) |
                               # data parallel: the (rank-independent) DAE regulariser enters the summed loss once
                               (_lib.RTX_STEP_NO_REG_IN_LOSS if red is not None and red.rank != 0 else 0))
        loss_out, loss_acc = st.loss_buf[0:1], st.loss_buf[1:2]
        mailbox = red is None and want_loss and self.loss_mailbox
        if mailbox and not getattr(eng, "_mailbox", False):
            eng.loss_mailbox(True)
        if red is None:
            eng.train_step(x, target, step, loss_out, loss_acc)
        elif native:
            # the engine schedules the whole data-parallel step (rectorch_amd/parallel.py NativePlan): ONE call
            if getattr(eng, "_dp_plan", None) is not red:
                eng.dp_attach(red)                   # (a rebuilt engine -- a larger batch arrived -- attaches again)
            red.error = None                         # (a stale failure of an earlier step must not be re-raised for this one)
            try:
                eng.train_step_dp(x, target, step, loss_out, loss_acc)
            except _lib.RtxError:
                if getattr(red, "error", None) is not None:
                    raise red.error                  # a collective issued through torch.distributed failed: its own message
                raise
            if eng._dp_any_sharded and red.transport != "emulate":   # what the ENGINE shards, not what the plan asked for
                st.masters_sharded = True
                self.network._rtx_masters_stale = True
        else:
            if getattr(red, "bucket_adam", False):
                g16 = red.grads16_ptrs()
                red.adam = lambda lo, hi: eng.apply_adam_layers(step, lo, hi, g16)
                if getattr(red, "sharded", False):
                    # sharded optimizer: this rank updates only its rows of the big matrices, then the compute copies
                    # are all-gathered in place (rectorch_amd/parallel.py)
                    red.adam_rows = lambda layer, lo, hi: eng.apply_adam_rows(
                        step, layer, lo, hi, True,
                        None if g16 is None else g16[2 * layer], None if g16 is None else g16[2 * layer + 1])
                    red.shadow = eng.shadow_tensor
                    st.masters_sharded = True
                    self.network._rtx_masters_stale = True
            else:
                red.adam = None
            eng.loss_grads(x, target, step, loss_out, loss_acc, layer_cb=red.on_layer)
            red.wait()
            if red.adam is None:
                eng.apply_adam(step)
            red.adam = None
        self.network._rtx_mark_updated(self.numerics)
        self._after_step()
        if want_loss:
            if red is not None:
                return red.reduce_scalar(st.loss_buf[0:1].clone())
            if mailbox:
                # THIS step's loss (reference models.py:835 `return loss.item()`) from the engine's host mailbox: the weight-gradient
                # + Adam kernels behind the loss keep running while the host returns and enqueues the next step
                return eng.wait_loss(st.adam_step)
            return st.loss_buf[0].item()       # the reference's per-step loss.item() sync
        return None

    def _join(self):
        """order the current stream behind everything the engines' side streams still carry (steps run with ``defer_join``)"""
        for eng in getattr(self.network, "_rtx_engines", {}).values():
            eng.join()

    def _read_loss_sum(self):
        self._join()
        st = self._rtx
        if st.reducer is not None:
            s = st.reducer.reduce_scalar(st.loss_buf[1:2].clone())
        else:
            s = st.loss_buf[1].item()
        st.loss_buf[1].zero_()
        return s

    # ----------------------------------------------------------------------------------- prediction
    def _predict_engine(self, n):
        """what every prediction starts with: the checks, eval mode and the engine of the predict numerics for batches of ``n`` rows
        (``evaluate_device`` does this ONCE per loader instead of once per batch)"""
        _lib.require_gpu()
        self._join()         # (an epoch interrupted between two deferred-join steps: order this stream behind the side streams first)
        if self._rtx.masters_sharded and self.predict_numerics != self.numerics:
            # that engine's compute copies come from the float32 masters, of which this rank holds only its rows
            raise _lib.RtxError("sharded optimizer: the float32 master rows of the other ranks are stale on this rank; call "
                                "model.consolidate() on EVERY rank (it is a collective) before predict() in another numerics "
                                "mode than the training one")
        self.network.eval()
        return self.network.rtx_engine(self.predict_numerics, n)

    def _predict_tuple(self, x, remove_train):
        x_in = self.network._as_input(x)
        n = len(x_in) if isinstance(x_in, RowBatch) else x_in.shape[0]
        eng = self._predict_engine(n)
        if not isinstance(x_in, RowBatch) and tagged_rows(x_in) is None:
            # dense input: through the PyTorch-ROCm custom op (torch.ops.rectorch_hip.*, rectorch_amd/ops.py)
            from . import ops  # noqa: F401  (registers the ops)
            if self._variant == "vae":
                return torch.ops.rectorch_hip.mvae_forward(eng.op_handle, x_in, False, bool(remove_train), 0)
            return (torch.ops.rectorch_hip.mdae_forward(eng.op_handle, x_in, False, bool(remove_train), 0), None, None)
        return eng.forward(x_in, training=False, remove_train=remove_train)

    def predict(self, x, remove_train=True):
        r"""Perform the prediction using a trained Autoencoder (reference models.py:449-473).  Returns the
        1-tuple ``(recon_x,)``; with ``remove_train`` the items of ``x`` are scored :math:`-\infty`."""
        recon_x, _, _ = self._predict_tuple(x, remove_train)
        return (recon_x, )

    # -------------------------------------------------------------------------------- checkpointing
    def consolidate(self):
        """Data parallel with the sharded optimizer: every rank holds current float32 master rows (and Adam moments) only for
        its own shard of the big matrices.  COLLECTIVE -- call it on EVERY rank -- before anything reads parameters from the
        host side: checkpoints, ``state_dict()``, ``predict`` in another numerics mode, a larger batch (the engine is rebuilt
        from the masters).  ``train()`` does so before each validation pass; ``save_model`` / ``predict`` raise when called with
        stale rows.  A no-op without the sharded optimizer."""
        self._gather_sharded_state()

    def _gather_sharded_state(self):
        st = self._rtx
        if st.reducer is None or not st.masters_sharded:
            return
        params = self.network._param_list()
        if getattr(st.reducer, "native", False):
            def tensors_n(layer):
                p = params[2 * layer]
                state = self.optimizer.state[p]
                return [p.data, state['exp_avg'], state['exp_avg_sq']]
            torch.cuda.current_stream().synchronize()
            eng = self.network._rtx_engines[self.numerics]
            st.reducer.gather_state(eng, tensors_n, len(params) // 2)
            st.masters_sharded = False
            self.network._rtx_masters_stale = False
            keep = self.network._rtx_shadow_versions.get(self.numerics)
            self.network._rtx_shadow_versions.clear()     # compute copies of the other numerics modes: refresh from the masters
            if keep is not None:
                # p.data was rewritten in place (version counters moved) with the values the training engine's compute copies
                # already hold: that engine's copies stay valid
                self.network._rtx_shadow_versions[self.numerics] = self.network._param_version()
            return

        def tensors(layer):
            p = params[2 * layer]
            state = self.optimizer.state[p]
            return [p.data, state['exp_avg'], state['exp_avg_sq']]
        st.reducer.wait()
        st.reducer.gather_state(tensors)
        st.masters_sharded = False
        self.network._rtx_masters_stale = False
        self.network._rtx_shadow_versions.clear()     # compute copies of the other numerics modes: refresh from the masters

    def _require_consolidated(self, what):
        if self._rtx.masters_sharded:
            raise _lib.RtxError("sharded optimizer: %s() needs the float32 master rows of every rank; call model.consolidate() "
                                "on EVERY rank first (it is a collective -- a rank-0-only checkpoint would deadlock inside it)" % what)

    def _sync_optimizer_state(self):
        """write the step count the fused Adam kernel is at into torch.optim.Adam's state"""
        self._join()         # (checkpoints read parameters and moments through torch: behind the engines' side streams)
        for p in self.network.parameters():
            state = self.optimizer.state.get(p)
            if state is not None and 'step' in state:
                state['step'] = torch.tensor(float(self._rtx.adam_step), dtype=torch.float32)

    def save_model(self, filepath, cur_epoch):
        r"""Save the model to file (reference models.py:475-489): ``epoch``, ``state_dict``, ``optimizer``."""
        self._require_consolidated("save_model")
        self._sync_optimizer_state()
        state = {'epoch': cur_epoch,
                 'state_dict': self.network.state_dict(),
                 'optimizer': self.optimizer.state_dict()
                }
        self._save_checkpoint(filepath, state)

    def _save_checkpoint(self, filepath, state):
        logger.info("Saving model checkpoint to %s...", filepath)
        torch.save(state, filepath)
        logger.info("Model checkpoint saved!")

    def load_model(self, filepath):
        r"""Load the model from file (reference models.py:496-516); returns the checkpoint dictionary."""
        assert os.path.isfile(filepath), "The checkpoint file %s does not exist." %filepath
        logger.info("Loading model checkpoint from %s...", filepath)
        checkpoint = torch.load(filepath, map_location=self.device)
        self.network.load_state_dict(checkpoint['state_dict'])
        self.optimizer.load_state_dict(checkpoint['optimizer'])
        steps = [int(float(s['step'])) for s in self.optimizer.state.values() if 'step' in s]
        self._rtx.adam_step = steps[0] if steps else 0
        logger.info("Model checkpoint loaded!")
        return checkpoint


class VAE(AETrainer):
    r"""Plumbing shared with the reference's ``VAE`` class (models.py:519-625): ``predict`` returning
    ``(recon_x, mu, logvar)``.  The reference's own BCE loss for a sigmoid VAE is not on the hot path."""
    _variant = "vae"

    def loss_function(self, recon_x, x, mu, logvar):
        raise NotImplementedError("the generic BCE VAE is outside the MI355X hot path; use MultiVAE")

    def predict(self, x, remove_train=True):
        r"""Perform the prediction using a trained Variational Autoencoder (reference models.py:594-625).
        Returns ``(recon_x, mu, logvar)``; with ``remove_train`` the items of ``x`` are scored
        :math:`-\infty`."""
        return self._predict_tuple(x, remove_train)


class MultiDAE(AETrainer):
    r"""Denoising Autoencoder with multinomial likelihood for collaborative filtering (reference
    models.py:628-706): Adam with (coupled) ``weight_decay=0.001``; loss = multinomial NLL +
    :math:`\lambda \sum_W \lVert W \rVert_2`.

    Parameters
    ----------
    mdae_net : :class:`rectorch_amd.nets.MultiDAE_net`
    lam : :obj:`float` [optional]
        The regularization hyper-parameter, by default 0.2.
    learning_rate : :obj:`float` [optional]
        By default 1e-3.
    """
    _variant = "dae"

    def __init__(self,
                 mdae_net,
                 lam=0.2,
                 learning_rate=1e-3,
                 numerics="bf16",
                 predict_numerics="fp32"):
        super(MultiDAE, self).__init__(mdae_net, learning_rate, numerics, predict_numerics)
        self.optimizer = optim.Adam(self.network.parameters(),
                                    lr=self.learning_rate,
                                    weight_decay=0.001)
        self.lam = lam

    def loss_function(self, recon_x, x):
        r"""Multinomial likelihood denoising autoencoder loss (reference models.py:662-706), on the HIP
        device; returns a 0-dim tensor."""
        from .engine import sum_l2_norms
        nll = multinomial_loss(recon_x, x)
        return nll + self.lam * sum_l2_norms([p.data for p in self.network.parameters()])

    def _step_scalars(self):
        return 0.0, self.lam


class MultiVAE(VAE):
    r"""Variational Autoencoder for collaborative Filtering (reference models.py:709-908).

    Parameters
    ----------
    mvae_net : :class:`rectorch_amd.nets.MultiVAE_net`
    beta : :obj:`float` [optional]
        The :math:`\beta` hyper-parameter of Multi-VAE, by default 1.0.
    anneal_steps : :obj:`int` [optional]
        Number of annealing steps for reaching the target value ``beta``, by default 0 (no annealing).
    learning_rate : :obj:`float` [optional]
        By default 1e-3.

    Attributes
    ----------
    anneal_steps, annealing, gradient_updates (a float, as in the reference), beta.
    """
    _uses_te = True      # the loss target is te_batch when given (reference models.py:819-822)

    def __init__(self,
                 mvae_net,
                 beta=1.,
                 anneal_steps=0,
                 learning_rate=1e-3,
                 numerics="bf16",
                 predict_numerics="fp32"):
        super(MultiVAE, self).__init__(mvae_net, learning_rate, numerics, predict_numerics)
        self.optimizer = optim.Adam(self.network.parameters(),
                                    lr=learning_rate,
                                    weight_decay=0.0)
        self.anneal_steps = anneal_steps
        self.annealing = anneal_steps > 0
        self.gradient_updates = 0.
        self.beta = beta

    def loss_function(self, recon_x, x, mu, logvar, beta=1.0):
        r"""VAE for collaborative filtering loss function (reference models.py:776-815):
        multinomial NLL + ``beta`` * KL, on the HIP device; returns a 0-dim tensor."""
        return multinomial_loss(recon_x, x, mu, logvar, beta)

    def _step_scalars(self):
        if self.annealing:
            anneal_beta = min(self.beta, 1. * self.gradient_updates / self.anneal_steps)
        else:
            anneal_beta = self.beta
        return anneal_beta, 0.0

    def _after_step(self):
        self.gradient_updates += 1.

    def train_batch(self, tr_batch, te_batch=None):
        r"""Training of a single batch (reference models.py:817-835): the encoder input is ``tr_batch``, the
        loss target ``te_batch`` when given, the KL weight is annealed with ``gradient_updates``."""
        return self._fused_step(tr_batch, te_batch, want_loss=True)

    def train(self,
              train_data,
              valid_data=None,
              valid_metric=None,
              valid_func=ValidFunc(evaluate),
              num_epochs=200,
              best_path="chkpt_best.pth",
              verbose=1):
        r"""Training procedure for Multi-VAE (reference models.py:837-895): per epoch ``train_epoch``, then -- when
        ``valid_data`` is truthy, as in the reference -- validation with ``valid_func`` and a checkpoint to ``best_path``
        whenever the mean of ``valid_metric`` improves on the best so far (metrics are assumed non-negative)."""
        best = -1.
        try:
            for epoch in range(1, num_epochs + 1):
                self.train_epoch(epoch, train_data, verbose)
                if valid_data:
                    self.consolidate()          # (data parallel, sharded optimizer: every rank runs train() and gets here)
                    score = _validate(self, epoch, valid_data, valid_metric, valid_func)
                    if score > best:
                        self.save_model(best_path, epoch)
                        best = score
        except KeyboardInterrupt:
            logger.warning('Handled KeyboardInterrupt: exiting from training early')

    def save_model(self, filepath, cur_epoch):
        r"""Save the model to file (reference models.py:897-903): adds ``gradient_updates``."""
        self._require_consolidated("save_model")
        self._sync_optimizer_state()
        state = {'epoch': cur_epoch,
                 'state_dict': self.network.state_dict(),
                 'optimizer': self.optimizer.state_dict(),
                 'gradient_updates': self.gradient_updates
                }
        self._save_checkpoint(filepath, state)

    def load_model(self, filepath):
        r"""Load the model from file and restore ``gradient_updates`` (reference models.py:905-908)."""
        checkpoint = super().load_model(filepath)
        self.gradient_updates = checkpoint['gradient_updates']
        return checkpoint


class CMultiVAE(MultiVAE):
    r"""Conditioned Variational Autoencoder for collaborative filtering (reference models.py:911-956).

    Same training procedure as :class:`MultiVAE`; the network is a :class:`rectorch_amd.nets.CMultiVAE_net` and the
    sampler a :class:`rectorch_amd.samplers.ConditionedDataSampler`, whose batches are ``[items | condition]`` rows
    with the condition-filtered rows as loss target.  ``predict`` masks the *item* columns of ``x`` only
    (``x[:, :-cond_dim].nonzero()``, reference models.py:952-953) -- the engine bounds the mask at ``n_items``.
    """
    def __init__(self,
                 cmvae_net,
                 beta=1.,
                 anneal_steps=0,
                 learning_rate=1e-3,
                 numerics="bf16",
                 predict_numerics="fp32"):
        super(CMultiVAE, self).__init__(cmvae_net,
                                        beta=beta,
                                        anneal_steps=anneal_steps,
                                        learning_rate=learning_rate,
                                        numerics=numerics,
                                        predict_numerics=predict_numerics)

    def predict(self, x, remove_train=True):
        return self._predict_tuple(x, remove_train)


class EASE(RecSysModel):
    r"""Embarrassingly Shallow AutoEncoder (reference rectorch/models.py:959-1069) solved on the MI355X.

    Same constructor, attributes and methods as the reference.  ``train`` computes the closed form
    :math:`B = P / (-\operatorname{diag} P),\ P = (X^\top X + \lambda I)^{-1},\ B_{ii} = 0` in float64 on the
    device (``rtx_ease_fit``: MFMA Gram matrix, blocked Cholesky, triangular inverse) and keeps ``B`` in HBM;
    ``predict`` multiplies the requested users' training rows by ``B`` on the device instead of looking them up
    in a materialised ``n_users x n_items`` score matrix.  ``model`` -- the reference's score matrix
    ``X B`` -- is materialised on the host only when it is read (``save_model`` does, to keep the file format).

    Parameters
    ----------
    lam : :obj:`float` [optional]
        The regularization hyper-parameter, by default 100.
    """
    def __init__(self, lam=100.):
        self.lam = lam
        self._solver = None
        self._model = None      # host score matrix: only after ``model`` was read or ``load_model``

    @property
    def model(self):
        """The score matrix **S** = X B as a :class:`numpy.ndarray` (``None`` before training)."""
        if self._model is None and self._solver is not None:
            n_users = self._solver.train.shape[0]
            out = np.empty((n_users, self._solver.n_items), dtype=np.float64)
            step = 8192
            for lo in range(0, n_users, step):
                hi = min(lo + step, n_users)
                out[lo:hi] = self._solver.scores(torch.arange(lo, hi, dtype=torch.int32)).cpu().numpy()
            self._model = out
        return self._model

    @model.setter
    def model(self, value):
        self._model = value
        self._solver = None

    def train(self, train_data):
        """Training of the EASE model (reference models.py:1006-1026).

        Parameters
        ----------
        train_data : :class:`scipy.sparse.csr_matrix`
            The training data.
        """
        logger.info("EASE - start tarining (lam=%.4f)", self.lam)
        self._model = None
        self._solver = EaseSolver(train_data, self.lam)
        logger.info("EASE - training complete")

    def predict(self, ids_te_users, test_tr, remove_train=True, as_tensor=False):
        r"""Prediction using the EASE model, :math:`S_{u}=\mathbf{X}_{u,:} \cdot \mathbf{B}` (reference
        models.py:1028-1057).

        Parameters
        ----------
        ids_te_users : array_like
            List of the test user indexes.
        test_tr : :class:`scipy.sparse.csr_matrix`
            Training portion of the test users.
        remove_train : :obj:`bool` [optional]
            Whether to set the scores of the items in ``test_tr`` to :math:`-\infty`, by default True.
        as_tensor : :obj:`bool` [optional]
            Return the float64 device tensor instead of copying it to a numpy array, by default False.

        Returns
        -------
        pred, : :obj:`tuple` with a single element
            The items' score (on the columns) for each user (on the rows).
        """
        if self._solver is None:
            if self._model is None:
                raise RuntimeError("EASE.predict called before train / load_model")
            pred = self._model[ids_te_users, :]         # a loaded score matrix is a pure look-up, as in the reference
            if remove_train:
                pred[test_tr.nonzero()] = -np.inf
            return (torch.from_numpy(pred).to("cuda"), ) if as_tensor else (pred, )
        mask = CsrMatrix(test_tr) if remove_train else None
        pred = self._solver.scores(ids_te_users, mask)
        return (pred, ) if as_tensor else (pred.cpu().numpy(), )

    def save_model(self, filepath):
        state = {'lambda': self.lam,
                 'model': self.model
                }
        logger.info("Saving EASE model to %s...", filepath)
        np.save(filepath, state)
        logger.info("Model saved!")

    def load_model(self, filepath):
        assert os.path.isfile(filepath), "The model file %s does not exist." %filepath
        logger.info("Loading EASE model from %s...", filepath)
        state = np.load(filepath, allow_pickle=True)[()]
        self.lam = state["lambda"]
        self.model = state["model"]
        logger.info("Model loaded!")
        return state

    def __str__(self):
        s = "EASE(lambda=%.4f" % self.lam
        if self._solver is not None:
            s += ", model size=(%d, %d))" % tuple(self._solver.train.shape)
        elif self._model is not None:
            s += ", model size=(%d, %d))" % self._model.shape
        else:
            s += ") - not trained yet!"
        return s

    def __repr__(self):
        return str(self)


class SVAE(MultiVAE):
    r"""Sequential Variational Autoencoders for Collaborative Filtering (reference models.py:1581-1635).

    One user sequence per optimizer step, as in the reference (several with ``SVAE_Sampler(pack=N)``, which is not: one
    Adam step for the mean loss of a pack of users): ``train_batch(x, y)`` with ``x`` the LongTensor
    ``[1, T]`` of the user's items and ``y`` the multi-hot targets ``[1, T, n_items]`` yielded by
    :class:`rectorch_amd.samplers.SVAE_Sampler` (or its compact :class:`rectorch_amd.engine.SvaeTarget`).  The whole
    step -- embedding, GRU, VAE head, decoder, loss, back-propagation through time, Adam with ``weight_decay=5e-3`` --
    is one call into librectorch_hip (``rtx_svae_train_step``), in float32.

    Parameters
    ----------
    svae_net : :class:`rectorch_amd.nets.SVAE_net`
    beta, anneal_steps, learning_rate
        As in :class:`MultiVAE`.
    numerics : "fp32" (default: the reference's arithmetic) or "bf16" -- every matrix product of the model with bf16 operands and
        float32 accumulation on the bf16 MFMA (what ``BASELINE.json`` ``configs[4]`` names); the GRU recurrences, the losses,
        the master weights and Adam stay float32.  Not in the reference's signature.
    """
    def __init__(self,
                 svae_net,
                 beta=1.,
                 anneal_steps=0,
                 learning_rate=1e-3,
                 numerics="fp32"):
        if numerics not in ("fp32", "bf16"):
            raise ValueError("numerics must be 'fp32' or 'bf16', got %r" % (numerics,))
        if getattr(svae_net, "_svae_engine", None) is not None and getattr(svae_net, "svae_numerics", "fp32") != numerics:
            svae_net._svae_engine = None          # built for the other arithmetic: rebuilt on first use
        svae_net.svae_numerics = numerics
        super(SVAE, self).__init__(svae_net,
                                   beta=beta,
                                   anneal_steps=anneal_steps,
                                   learning_rate=learning_rate,
                                   numerics="fp32",
                                   predict_numerics="fp32")
        self.optimizer = optim.Adam(self.network.parameters(),
                                    lr=learning_rate,
                                    weight_decay=5e-3)

    def loss_function(self, recon_x, x, mu, logvar, beta=1.0):
        r"""SVAE loss (reference models.py:1622-1626): :math:`\sum_t \mathrm{NLL}_t / d + \beta \cdot
        \mathrm{mean}_t \mathrm{KL}_t` with :math:`d` = ``sum(x[0, :n_items])`` -- all the ones of a ``[1, T, n_items]``
        target, but only those of the first time step when the target arrives flattened, as it does from
        ``train_batch``.  Computed by the HIP multinomial-loss kernel on the ``T`` rows."""
        T, n_items = recon_x.shape[1], recon_x.shape[2]
        d = float(torch.sum(x[0, :n_items]))
        y = x.reshape(T, n_items).to(recon_x.device, torch.float32)
        scale = T / d
        return scale * multinomial_loss(recon_x.reshape(T, n_items), y, mu, logvar, beta / scale)

    def train_batch(self, tr_batch, te_batch=None):
        r"""Training of a single sequence (reference models.py:817-835 with the SVAE loss and optimizer)."""
        _lib.require_gpu()
        if te_batch is None:
            raise ValueError("SVAE.train_batch needs the target sequence (SVAE_Sampler yields it)")
        st, params, m, v = self._ensure_train_state()
        pack = tr_batch if isinstance(tr_batch, SvaePack) else None
        T = pack.n_steps if pack is not None else int(tr_batch.numel())
        eng = self.network.svae_engine(T, train_buffers=(st.grads, m, v))
        if self.loss_mailbox and not getattr(eng, "_mailbox", False):
            eng.loss_mailbox(True)
        if pack is not None:
            d = 1.0           # every user's own normaliser travels with the pack's rows
        elif isinstance(te_batch, SvaeTarget):
            d = te_batch.d
        else:
            # the reference flattens the target to [1, T * n_items] before loss_function reads x[0, :n_items]
            d = float(te_batch.reshape(te_batch.shape[0], -1)[0, :eng.n_items].sum())
        if self.annealing:
            anneal_beta = min(self.beta, 1. * self.gradient_updates / self.anneal_steps)
        else:
            anneal_beta = self.beta
        g = self.optimizer.param_groups[0]
        st.adam_step += 1
        from .nets import draw_seed
        noise = self._rtx.inject[1] if self._rtx.inject else None     # parity tests: eps captured from the reference's RNG
        step = _lib.Step()
        step.beta, step.lam = float(anneal_beta), 0.0
        step.inv_batch = (1.0 / d) if d != 0 else float("inf")
        step.lr, step.beta1, step.beta2 = float(g['lr']), float(g['betas'][0]), float(g['betas'][1])
        step.eps, step.weight_decay = float(g['eps']), float(g['weight_decay'])
        step.step, step.flags = st.adam_step, 0
        step.seed, step.offset = draw_seed() & (2 ** 64 - 1), 0
        step.dropout_mask = None
        step.eps_noise = None if noise is None else noise.data_ptr()
        if pack is not None:
            # SVAE_Sampler(pack=N): ONE optimizer step for the mean of the pack's per-user losses (not in the reference)
            eng.train_pack(pack, step, anneal_beta, st.loss_buf[0:1], st.loss_buf[1:2])
        else:
            eng.train_step(tr_batch, te_batch, step, st.loss_buf[0:1], st.loss_buf[1:2])
        self.gradient_updates += 1.
        if getattr(eng, "_mailbox", False):
            return eng.wait_loss()        # this step's loss (models.py:835) from the host mailbox: the backward half of the step is still running
        return st.loss_buf[0].item()

    def predict(self, x, remove_train=True):
        r"""Scores of the step after the sequence ``x`` (reference models.py:1628-1635): ``(recon_x[:, -1, :], mu,
        logvar)``; with ``remove_train`` the items of ``x`` are scored :math:`-\infty`.  The latent code is sampled
        (the reference's ``VAE_net._reparameterize`` has no eval branch): seed torch's generator for repeatable scores."""
        _lib.require_gpu()
        self.network.eval()
        from .nets import draw_seed
        noise = self._rtx.inject[1] if self._rtx.inject else None
        eng = self.network.svae_engine(int(x.numel()))
        _, last, mu, logvar = eng.forward(x, noise=noise, seed=draw_seed(), remove_train=remove_train, want_all=False)
        return last.view(1, -1), mu, logvar
