"""torch-CPU restatement of the reference trainer's op sequence -- TEST INFRASTRUCTURE ONLY.

This is the CPU baseline ("port") timed next to the HIP path in bench.py: it executes the same ATen
CPU op sequence the reference executes per step (the reference's arithmetic lives in torch,
SURVEY.md §8c), written from the math in SURVEY.md Appendix B:

    DataSampler.__iter__      rectorch/samplers.py:91-107   scipy row gather -> toarray -> FloatTensor
    MultiVAE_net.forward      rectorch/nets.py:394-417      normalize, dropout, Linear/tanh, reparam
    MultiVAE.loss_function    rectorch/models.py:813-815    log_softmax NLL + beta*KL
    MultiVAE.train_batch      rectorch/models.py:817-835    zero_grad, forward, backward, Adam.step
    MultiDAE (variant="dae")  rectorch/nets.py:219-233, models.py:657-659, 701-706

It is validated against the imported reference (bit-exact under the same torch seed) in
tests/test_oracle_golden.py via the golden vectors.
"""
import numpy as np
import torch
import torch.nn.functional as F


class CpuNet(torch.nn.Module):
    def __init__(self, enc_dims, dec_dims, variant="vae", dropout=0.5):
        super().__init__()
        self.variant, self.p = variant, dropout
        self.latent = enc_dims[-1]
        enc_out = list(enc_dims[1:])
        if variant == "vae":
            enc_out[-1] *= 2
        self.enc = torch.nn.ModuleList([torch.nn.Linear(i, o) for i, o in zip(enc_dims[:-1], enc_out)])
        self.dec = torch.nn.ModuleList([torch.nn.Linear(i, o) for i, o in zip(dec_dims[:-1], dec_dims[1:])])

    def load_numpy(self, params):
        with torch.no_grad():
            for p, a in zip(self.parameters(), params):
                p.copy_(torch.from_numpy(np.asarray(a, dtype=np.float32)))

    def forward(self, x):
        h = F.normalize(x)
        if self.training:
            h = F.dropout(h, self.p, True)
        n = len(self.enc)
        for i, layer in enumerate(self.enc):
            h = layer(h)
            if self.variant == "dae" or i != n - 1:
                h = torch.tanh(h)
        mu = logvar = None
        if self.variant == "vae":
            mu, logvar = h[:, :self.latent], h[:, self.latent:]
            if self.training:
                std = torch.exp(0.5 * logvar)
                h = mu + torch.randn_like(std) * std
            else:
                h = mu
        for i, layer in enumerate(self.dec):
            h = layer(h)
            if i != len(self.dec) - 1:
                h = torch.tanh(h)
        return h, mu, logvar


class CpuTrainer:
    def __init__(self, net, beta=1.0, anneal_steps=0, lam=0.2, lr=1e-3):
        self.net = net
        wd = 0.0 if net.variant == "vae" else 0.001
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, weight_decay=wd)
        self.beta, self.anneal_steps, self.lam = beta, anneal_steps, lam
        self.gradient_updates = 0.0

    def loss(self, y, gt, mu, logvar, beta):
        nll = -torch.mean(torch.sum(F.log_softmax(y, 1) * gt, -1))
        if self.net.variant == "vae":
            kld = -0.5 * torch.mean(torch.sum(1 + logvar - mu.pow(2) - logvar.exp(), dim=1))
            return nll + beta * kld
        reg = 0
        for w in self.net.parameters():
            reg = reg + w.norm(2)
        return nll + self.lam * reg

    def train_batch(self, x, gt=None):
        gt = x if gt is None else gt
        beta = self.beta
        if self.net.variant == "vae" and self.anneal_steps > 0:
            beta = min(self.beta, self.gradient_updates / self.anneal_steps)
        self.net.train()
        self.opt.zero_grad()
        y, mu, logvar = self.net(x)
        loss = self.loss(y, gt, mu, logvar, beta)
        loss.backward()
        self.opt.step()
        self.gradient_updates += 1.0
        return loss.item()

    def predict(self, x, remove_train=True):
        self.net.eval()
        with torch.no_grad():
            y, mu, logvar = self.net(x)
            if remove_train:
                y[tuple(x.nonzero().t())] = -np.inf
        return (y, mu, logvar) if self.net.variant == "vae" else (y,)


def densify_batch(csr, idx):
    """The reference DataSampler's per-batch work (samplers.py:99-100)."""
    return torch.FloatTensor(csr[idx].toarray())
