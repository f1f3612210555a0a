"""Network classes of the Mult-VAE / Mult-DAE path with the reference's API (rectorch/nets.py), computed
by librectorch_hip on an MI355X.

The classes are ``torch.nn.Module`` s holding ordinary ``nn.Linear`` parameters (same ``state_dict`` keys
and shapes as the reference: ``enc_layers.{i}.weight/bias``, ``dec_layers.{i}.weight/bias``, so reference
checkpoints load), but ``encode`` / ``decode`` / ``forward`` do not run torch kernels: they hand the
parameters' device pointers to the HIP engine.  These methods are forward-only (no autograd graph); training
goes through :class:`rectorch_amd.models.MultiVAE` / ``MultiDAE`` which use the engine's fused
forward+loss+backward+Adam step.

Reference: AE_net nets.py:22-96, MultiDAE_net :175-247, VAE_net :250-353, MultiVAE_net :356-417.
"""
import logging

import torch
from torch import nn
from torch.nn.init import xavier_uniform_ as xavier_init
from torch.nn.init import normal_ as normal_init

from . import _lib
from .engine import Engine

__all__ = ['AE_net', 'MultiDAE_net', 'VAE_net', 'MultiVAE_net', 'CMultiVAE_net', 'SVAE_net']

logger = logging.getLogger(__name__)

DEFAULT_MAX_BATCH = 512


def draw_seed():
    """One 63-bit Philox seed per stochastic call, drawn from torch's global CPU generator: like the
    reference, ``torch.manual_seed(s)`` makes dropout masks and eps reproducible (tests/test_nets.py:56-74:
    encode and forward under the same seed give the same mu/logvar)."""
    return int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item())


class AE_net(nn.Module):
    r"""Abstract Autoencoder network (reference nets.py:22-96).

    Parameters
    ----------
    dec_dims : list of int
        Decoder dimensions, ``dec_dims[0]`` latent size ... ``dec_dims[-1]`` input size.
    enc_dims : list of int or None
        Encoder dimensions; if it evaluates to False, ``enc_dims = dec_dims[::-1]``.
    """
    _variant = None

    def __init__(self, dec_dims, enc_dims=None):
        super(AE_net, self).__init__()
        if enc_dims:
            self.enc_dims = enc_dims
        else:
            self.enc_dims = dec_dims[::-1]
        self.dec_dims = dec_dims
        self._rtx_engines = {}
        self._rtx_shadow_versions = {}
        self._rtx_masters_stale = False     # data parallel, sharded optimizer: this rank's float32 masters hold only ITS rows

    def encode(self, x):
        raise NotImplementedError()

    def decode(self, z):
        raise NotImplementedError()

    def forward(self, x):
        z = self.encode(x)
        return self.decode(z)

    def init_weights(self):
        raise NotImplementedError()

    # ---- HIP engine plumbing (not part of the reference API) -------------------------------------
    def _param_list(self):
        ps = []
        for layer in list(self.enc_layers) + list(self.dec_layers):
            ps += [layer.weight, layer.bias]
        return ps

    def _param_version(self):
        return tuple((p.data_ptr(), p._version) for p in self._param_list())

    def _device(self):
        p = next(self.parameters())
        if not p.is_cuda:
            _lib.require_gpu()
            raise _lib.RtxError("the network's parameters are on %s; move it to the MI355X first (net.to('cuda')): "
                                "rectorch_amd has no CPU compute path" % p.device)
        return p.device

    def rtx_engine(self, numerics="fp32", max_batch=DEFAULT_MAX_BATCH, train_buffers=None):
        """The engine of this network for a numerics mode ("fp32": exact-f32 MFMA, parity mode; "bf16":
        bf16 MFMA with f32 accumulation).  Rebuilt when a larger batch arrives, re-bound when the parameter
        storage changes, shadows refreshed when the parameters were modified by anyone but the engine."""
        self._device()
        eng = self._rtx_engines.get(numerics)
        if getattr(self, "_rtx_masters_stale", False) and (
                eng is None or eng.max_batch < max_batch or self._rtx_shadow_versions.get(numerics) != self._param_version()):
            # building an engine or refreshing its compute copies reads the float32 masters, of which this rank holds current
            # values only for its own rows of the sharded matrices: wrong weights with no error otherwise
            raise _lib.RtxError("sharded optimizer: an engine (%s numerics, batch %d) would be rebuilt from float32 masters whose "
                                "rows of other ranks are stale; call model.consolidate() on EVERY rank first" % (numerics, max_batch))
        if eng is None or eng.max_batch < max_batch:
            mb = max(max_batch, DEFAULT_MAX_BATCH if eng is None else eng.max_batch)
            eng = Engine(self.enc_dims, self.dec_dims, self._variant, self.dropout.p, numerics, mb,
                         cond_dim=getattr(self, "cond_dim", 0))
            self._rtx_engines[numerics] = eng
            self._rtx_shadow_versions.pop(numerics, None)
        params = [p.data for p in self._param_list()]
        pkey = tuple(t.data_ptr() for t in params)
        tkey = None if train_buffers is None else tuple(t.data_ptr() for ts in train_buffers for t in ts)
        if getattr(eng, "param_key", None) != pkey or (tkey is not None and getattr(eng, "train_key", None) != tkey):
            if train_buffers is not None:
                eng.bind(params, *train_buffers)
            else:
                eng.bind(params)
            eng.param_key, eng.train_key = pkey, tkey
            self._rtx_shadow_versions.pop(numerics, None)
        ver = self._param_version()
        if self._rtx_shadow_versions.get(numerics) != ver:
            eng.sync_shadows()
            self._rtx_shadow_versions[numerics] = ver
        return eng

    def _rtx_mark_updated(self, numerics):
        """Called by the trainer after an engine-side Adam step: that engine's shadows are fresh, every other
        engine's are stale (the master parameters changed under them)."""
        ver = self._param_version()
        for k in list(self._rtx_shadow_versions):
            if k != numerics:
                self._rtx_shadow_versions.pop(k)
        self._rtx_shadow_versions[numerics] = ver

    def _as_input(self, x):
        from .engine import RowBatch, tagged_rows
        if isinstance(x, RowBatch) or tagged_rows(x) is not None:
            return x
        dev = self._device()
        return x.reshape(x.shape[0], -1).to(dev, torch.float32).contiguous()

    def _init_linear(self):
        for layer in self.enc_layers:
            xavier_init(layer.weight)
            normal_init(layer.bias)
        for layer in self.dec_layers:
            xavier_init(layer.weight)
            normal_init(layer.bias)


class MultiDAE_net(AE_net):
    r"""Denoising Autoencoder network for collaborative filtering (reference nets.py:175-247):
    L2-normalise -> dropout -> tanh(Linear) on every encoder layer; decoder tanh on all but the last layer.
    """
    _variant = "dae"

    def __init__(self, dec_dims, enc_dims=None, dropout=0.5):
        super(MultiDAE_net, self).__init__(dec_dims, enc_dims)
        self.dropout = nn.Dropout(dropout)
        self.enc_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(self.enc_dims[:-1], self.enc_dims[1:])])
        self.dec_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(self.dec_dims[:-1], self.dec_dims[1:])])
        self.init_weights()

    def encode(self, x):
        x = self._as_input(x)
        eng = self.rtx_engine("fp32", x.shape[0])
        h, _ = eng.encode(x, training=self.training, seed=draw_seed() if self.training else 0)
        return h

    def decode(self, z):
        eng = self.rtx_engine("fp32", z.shape[0])
        return eng.decode(z.to(self._device()))

    def forward(self, x):
        x = self._as_input(x)
        eng = self.rtx_engine("fp32", x.shape[0])
        logits, _, _ = eng.forward(x, training=self.training, seed=draw_seed() if self.training else 0)
        return logits

    def init_weights(self):
        r"""xavier_uniform weights, N(0,1) biases (reference nets.py:235-247)."""
        self._init_linear()


class VAE_net(AE_net):
    r"""Variational Autoencoder network skeleton (reference nets.py:250-353): builds the layers (the last
    encoder layer emits mean and log-variance) and initialises them.  The generic sigmoid/BCE VAE forward
    of the reference is outside the MI355X hot path; :class:`MultiVAE_net` provides encode/decode."""
    _variant = "vae"

    def __init__(self, dec_dims, enc_dims=None):
        super(VAE_net, self).__init__(dec_dims, enc_dims)
        temp_dims = self.enc_dims[:-1] + [self.enc_dims[-1] * 2]
        self.enc_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(temp_dims[:-1], temp_dims[1:])])
        self.dec_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(self.dec_dims[:-1], self.dec_dims[1:])])
        self.init_weights()

    def encode(self, x):
        raise NotImplementedError("the generic (sigmoid/BCE) VAE_net is not on the MI355X hot path; use MultiVAE_net")

    def decode(self, z):
        raise NotImplementedError("the generic (sigmoid/BCE) VAE_net is not on the MI355X hot path; use MultiVAE_net")

    def forward(self, x):
        raise NotImplementedError("the generic (sigmoid/BCE) VAE_net is not on the MI355X hot path; use MultiVAE_net")

    def init_weights(self):
        r"""xavier_uniform weights, N(0,1) biases (reference nets.py:341-353)."""
        self._init_linear()


class MultiVAE_net(VAE_net):
    r"""Variational Autoencoder network for collaborative filtering (reference nets.py:356-417):
    ``encode`` = L2-normalise -> dropout (training) -> tanh(Linear) ... -> Linear -> (mu, logvar);
    ``_reparameterize`` samples in training, returns mu in eval; ``decode`` = tanh(Linear) ... -> Linear.
    """

    def __init__(self, dec_dims, enc_dims=None, dropout=0.5):
        super(MultiVAE_net, self).__init__(dec_dims, enc_dims)
        self.dropout = nn.Dropout(dropout)

    def encode(self, x):
        x = self._as_input(x)
        eng = self.rtx_engine("fp32", x.shape[0])
        return eng.encode(x, training=self.training, seed=draw_seed() if self.training else 0)

    def _reparameterize(self, mu, logvar):
        if self.training:
            # same Philox family as the fused path, element-wise on a [B, latent] tensor: plumbing-sized
            raise NotImplementedError("sampling outside forward() is not exposed; call net(x) in training mode")
        return mu

    def decode(self, z):
        eng = self.rtx_engine("fp32", z.shape[0])
        return eng.decode(z.to(self._device()))

    def forward(self, x):
        x = self._as_input(x)
        eng = self.rtx_engine("fp32", x.shape[0])
        return eng.forward(x, training=self.training, seed=draw_seed() if self.training else 0)


class CMultiVAE_net(MultiVAE_net):
    r"""Conditioned Variational Autoencoder network for collaborative filtering (reference nets.py:420-480).

    The input rows are ``[items | condition one-hot]``: the item part is L2-normalised and dropped out, the
    ``cond_dim`` condition columns are concatenated raw, so the first encoder layer is
    ``Linear(n_items + cond_dim, ...)``.  On the device this is the same engine with ``rtx_cfg.cond_dim`` set: the
    gather kernel leaves the trailing columns unscaled and the first GEMM simply has a longer K.
    """

    def __init__(self, cond_dim, dec_dims, enc_dims=None, dropout=0.5):
        super(CMultiVAE_net, self).__init__(dec_dims, enc_dims, dropout)
        self.cond_dim = cond_dim

        temp_dims = self.enc_dims[:-1] + [self.enc_dims[-1] * 2]
        temp_dims[0] += self.cond_dim
        self.enc_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(temp_dims[:-1], temp_dims[1:])])

        self.dec_layers = nn.ModuleList(
            [nn.Linear(d_in, d_out) for d_in, d_out in zip(self.dec_dims[:-1], self.dec_dims[1:])])
        self.init_weights()


class SVAE_net(VAE_net):
    """Sequential Variational Autoencoder network (reference nets.py:624-693): item embedding -> one-layer GRU
    (batch_first) -> the VAE head of :class:`VAE_net` (tanh hidden layers, linear ``mu | logvar``; **always** sampled,
    in eval too, because ``_reparameterize`` is the base class's) -> tanh MLP decoder with a linear output.

    The ``nn.Embedding`` / ``nn.GRU`` / ``nn.Linear`` modules hold the parameters (same ``state_dict`` keys as the
    reference: ``enc_layers.*``, ``dec_layers.*``, ``item_embed.weight``, ``gru.weight_ih_l0`` ...); the computation is
    the ``rtx_svae`` engine of librectorch_hip.

    Parameters
    ----------
    n_items : :obj:`int`
        Number of items.
    embed_size : :obj:`int`
        Size of the embedding for the items.
    rnn_size : :obj:`int`
        Size of the recurrent layer of the GRU part of the network.
    dec_dims, enc_dims : :obj:`list` of :obj:`int`
        See :class:`AE_net` (``enc_dims[0]`` must be ``rnn_size``).
    """
    def __init__(self, n_items, embed_size, rnn_size, dec_dims, enc_dims):
        super(SVAE_net, self).__init__(dec_dims, enc_dims)
        self.enc_dims = enc_dims
        self.dec_dims = dec_dims
        self.n_items = n_items
        self.embed_size = embed_size
        self.rnn_size = rnn_size
        self.item_embed = nn.Embedding(n_items, embed_size)
        self.gru = nn.GRU(embed_size, rnn_size, batch_first=True, num_layers=1)
        self.init_weights()
        self._svae_engine = None

    def init_weights(self):
        r"""xavier_normal weights for the encoder / decoder layers, biases left at their defaults (reference
        nets.py:689-693)."""
        for layer in self.enc_layers:
            nn.init.xavier_normal_(layer.weight)
        for layer in self.dec_layers:
            nn.init.xavier_normal_(layer.weight)

    def _param_list(self):
        # the order of SVAE_net.parameters(): the MLPs are registered by VAE_net.__init__, then embedding and GRU
        ps = []
        for layer in list(self.enc_layers) + list(self.dec_layers):
            ps += [layer.weight, layer.bias]
        return ps + [self.item_embed.weight, self.gru.weight_ih_l0, self.gru.weight_hh_l0, self.gru.bias_ih_l0,
                     self.gru.bias_hh_l0]

    def svae_engine(self, seq_len=1, train_buffers=None):
        """The ``rtx_svae`` engine of this network, grown when a longer sequence arrives and re-bound when the parameter
        (or training-buffer) storage changes.  There are no compute copies: the kernels read the parameters in place."""
        from .engine import SvaeEngine
        self._device()
        eng = self._svae_engine
        if eng is None or eng.max_len < seq_len:
            eng = SvaeEngine(self.n_items, self.embed_size, self.rnn_size, self.enc_dims, self.dec_dims,
                             max_len=max(256, 2 * int(seq_len)))
            if getattr(self, "svae_numerics", "fp32") == "bf16":     # set by SVAE(..., numerics="bf16")
                eng.set_option("gemm_bf16", 1)
            self._svae_engine = eng
        params = [p.data for p in self._param_list()]
        pkey = tuple(t.data_ptr() for t in params)
        tkey = None if train_buffers is None else tuple(t.data_ptr() for ts in train_buffers for t in ts)
        if getattr(eng, "param_key", None) != pkey or (tkey is not None and getattr(eng, "train_key", None) != tkey):
            if train_buffers is not None:
                eng.bind(params, *train_buffers)
            else:
                eng.bind(params)
            eng.param_key, eng.train_key = pkey, tkey
        return eng

    def _rtx_mark_updated(self, numerics):
        pass

    def forward(self, x):
        """``x``: LongTensor [1, T].  Returns ``(dec_out [1, T, n_items], mu [T, latent], logvar [T, latent])``."""
        la, _, mu, logvar = self.svae_engine(x.numel()).forward(x, seed=draw_seed())
        return la.view(x.shape[0], x.shape[1], -1), mu, logvar
