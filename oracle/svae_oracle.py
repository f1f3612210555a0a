"""TEST INFRASTRUCTURE ONLY -- numpy (float64) restatement of the reference's SVAE (rectorch/nets.py:624-693 SVAE_net,
rectorch/models.py:1609-1635 SVAE loss / optimizer / predict, torch.nn.GRU's cell equations, torch.optim.Adam with coupled
weight decay).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import it.

Pinned by tests/golden/g12_svae_*.npz (outputs of the reference's own classes run by tests/golden/make_golden.py)."""
import numpy as np


def _sig(x):
    return 1.0 / (1.0 + np.exp(-x))


class SvaeOracle:
    """params: dict with the reference's state-dict keys (enc_layers.i.weight ... item_embed.weight, gru.weight_ih_l0 ...)."""

    def __init__(self, sd, n_enc, n_dec, lr=1e-3, weight_decay=5e-3, beta=1.0, anneal_steps=0):
        self.p = {k: np.array(v, dtype=np.float64) for k, v in sd.items()}
        self.n_enc, self.n_dec = n_enc, n_dec
        self.keys = ["%s.%d.%s" % (part, i, kind) for part, n in (("enc_layers", n_enc), ("dec_layers", n_dec))
                     for i in range(n) for kind in ("weight", "bias")]
        self.keys += ["item_embed.weight", "gru.weight_ih_l0", "gru.weight_hh_l0", "gru.bias_ih_l0", "gru.bias_hh_l0"]
        self.m = {k: np.zeros_like(self.p[k]) for k in self.keys}
        self.v = {k: np.zeros_like(self.p[k]) for k in self.keys}
        self.lr, self.wd, self.beta, self.anneal_steps = lr, weight_decay, beta, anneal_steps
        self.step = 0
        self.gradient_updates = 0.0

    # nets.py:666-676
    def forward(self, items, eps):
        p = self.p
        X = p["item_embed.weight"][items]                                   # [T, E]
        T = len(items)
        R = p["gru.weight_hh_l0"].shape[1]
        gi = X @ p["gru.weight_ih_l0"].T + p["gru.bias_ih_l0"]
        H = np.zeros((T + 1, R))
        r_, z_, n_, hn_ = (np.zeros((T, R)) for _ in range(4))
        for t in range(T):
            gh = p["gru.weight_hh_l0"] @ H[t] + p["gru.bias_hh_l0"]
            r = _sig(gi[t, :R] + gh[:R])
            z = _sig(gi[t, R:2 * R] + gh[R:2 * R])
            hn = gh[2 * R:]
            n = np.tanh(gi[t, 2 * R:] + r * hn)
            H[t + 1] = (1 - z) * n + z * H[t]
            r_[t], z_[t], n_[t], hn_[t] = r, z, n, hn
        acts = [H[1:]]
        h = H[1:]
        for i in range(self.n_enc):                                          # VAE_net.encode, nets.py:287-295
            h = h @ p["enc_layers.%d.weight" % i].T + p["enc_layers.%d.bias" % i]
            if i != self.n_enc - 1:
                h = np.tanh(h)
            acts.append(h)
        Z = h.shape[1] // 2
        mu, lv = h[:, :Z], h[:, Z:]
        z = mu + eps * np.exp(0.5 * lv)                                      # always sampled, nets.py:316-319
        acts.append(z)
        h = z
        for i in range(self.n_dec):                                          # SVAE_net.decode, nets.py:683-687
            h = h @ p["dec_layers.%d.weight" % i].T + p["dec_layers.%d.bias" % i]
            if i != self.n_dec - 1:
                h = np.tanh(h)
            acts.append(h)
        cache = dict(items=np.asarray(items), X=X, gi=gi, H=H, r=r_, z=z_, n=n_, hn=hn_, acts=acts, mu=mu, lv=lv, eps=eps)
        return h, mu, lv, cache

    # models.py:1622-1626
    def loss_and_grads(self, items, y, eps, beta, likelihood_d=None):
        """``likelihood_d``: the normaliser of the likelihood.  ``loss_function`` computes it as
        ``sum(x[0, :n_items])`` (models.py:1623); through ``train_batch`` x is the target flattened to [1, T * n_items]
        (models.py:822), so it is the number of ones of the FIRST time step.  None -> all ones of the target."""
        p = self.p
        logits, mu, lv, c = self.forward(items, eps)
        T = logits.shape[0]
        mx = logits.max(1, keepdims=True)
        lse = mx + np.log(np.exp(logits - mx).sum(1, keepdims=True))
        d = float(y.sum()) if likelihood_d is None else float(likelihood_d)
        nll = -((logits - lse) * y).sum()
        kld = -0.5 * np.mean(np.sum(1 + lv - mu ** 2 - np.exp(lv), axis=1))
        loss = nll / d + beta * kld
        g = {}
        dpre = (y.sum(1, keepdims=True) * np.exp(logits - lse) - y) / d     # d loss / d logits
        acts = c["acts"]
        # acts: [rnn_out, enc_0 .. enc_{ne-1}, z, dec_0 .. dec_{nd-1}]
        for i in range(self.n_dec - 1, -1, -1):
            inp = acts[self.n_enc + 1 + i]
            g["dec_layers.%d.weight" % i] = dpre.T @ inp
            g["dec_layers.%d.bias" % i] = dpre.sum(0)
            dinp = dpre @ p["dec_layers.%d.weight" % i]
            if i > 0:
                dpre = dinp * (1 - inp ** 2)
        dz = dinp
        Z = mu.shape[1]
        dhead = np.concatenate([dz + beta * mu / T, dz * eps * 0.5 * np.exp(0.5 * lv) + beta * 0.5 * (np.exp(lv) - 1) / T], axis=1)
        dpre = dhead
        for i in range(self.n_enc - 1, -1, -1):
            inp = acts[i]
            g["enc_layers.%d.weight" % i] = dpre.T @ inp
            g["enc_layers.%d.bias" % i] = dpre.sum(0)
            dinp = dpre @ p["enc_layers.%d.weight" % i]
            if i > 0:
                dpre = dinp * (1 - inp ** 2)
        dH_out = dinp
        R = c["H"].shape[1]
        Whh = p["gru.weight_hh_l0"]
        dgi = np.zeros((T, 3 * R))
        dgh = np.zeros((T, 3 * R))
        dh = np.zeros(R)
        for t in range(T - 1, -1, -1):
            dd = dh + dH_out[t]
            r, z, n, hn, hp = c["r"][t], c["z"][t], c["n"][t], c["hn"][t], c["H"][t]
            dn = dd * (1 - z)
            dzp = dd * (hp - n) * z * (1 - z)
            dnp = dn * (1 - n ** 2)
            drp = dnp * hn * r * (1 - r)
            dgi[t] = np.concatenate([drp, dzp, dnp])
            dgh[t] = np.concatenate([drp, dzp, dnp * r])
            dh = dd * z + Whh.T @ dgh[t]
        g["gru.weight_hh_l0"] = dgh.T @ c["H"][:-1]
        g["gru.bias_hh_l0"] = dgh.sum(0)
        g["gru.weight_ih_l0"] = dgi.T @ c["X"]
        g["gru.bias_ih_l0"] = dgi.sum(0)
        dX = dgi @ p["gru.weight_ih_l0"]
        ge = np.zeros_like(p["item_embed.weight"])
        np.add.at(ge, c["items"], dX)
        g["item_embed.weight"] = ge
        return loss, g, logits, mu, lv

    def anneal_beta(self):
        if self.anneal_steps > 0:
            return min(self.beta, self.gradient_updates / self.anneal_steps)
        return self.beta

    # MultiVAE.train_batch (models.py:817-835) with torch.optim.Adam(weight_decay=5e-3) (models.py:1618-1620)
    def train_batch(self, items, y, eps, beta1=0.9, beta2=0.999, adam_eps=1e-8, likelihood_d=None):
        if likelihood_d is None:
            likelihood_d = float(np.asarray(y)[0].sum())          # what train_batch's flattening produces
        loss, g, _, _, _ = self.loss_and_grads(items, y, eps, self.anneal_beta(), likelihood_d)
        self.step += 1
        bc1, bc2 = 1 - beta1 ** self.step, 1 - beta2 ** self.step
        for k in self.keys:
            gk = g[k] + self.wd * self.p[k]
            self.m[k] = beta1 * self.m[k] + (1 - beta1) * gk
            self.v[k] = beta2 * self.v[k] + (1 - beta2) * gk * gk
            self.p[k] = self.p[k] - (self.lr / bc1) * self.m[k] / (np.sqrt(self.v[k]) / np.sqrt(bc2) + adam_eps)
        self.gradient_updates += 1.0
        self.last_grads = g
        return loss

    def train_pack(self, users, beta1=0.9, beta2=0.999, adam_eps=1e-8):
        """NOT in the reference: one Adam step for the MEAN over ``users`` (list of (items, y, eps)) of the per-user loss --
        the gradients of the reference's per-user objective accumulated over the pack, then the reference's optimizer."""
        beta = self.anneal_beta()
        loss, g = 0.0, None
        for items, y, eps in users:
            d = float(np.asarray(y)[0].sum())
            l, gu, _, _, _ = self.loss_and_grads(items, y, eps, beta, d)
            loss += l / len(users)
            g = {k: gu[k] / len(users) for k in self.keys} if g is None else {k: g[k] + gu[k] / len(users) for k in self.keys}
        self.step += 1
        bc1, bc2 = 1 - beta1 ** self.step, 1 - beta2 ** self.step
        for k in self.keys:
            gk = g[k] + self.wd * self.p[k]
            self.m[k] = beta1 * self.m[k] + (1 - beta1) * gk
            self.v[k] = beta2 * self.v[k] + (1 - beta2) * gk * gk
            self.p[k] = self.p[k] - (self.lr / bc1) * self.m[k] / (np.sqrt(self.v[k]) / np.sqrt(bc2) + adam_eps)
        self.gradient_updates += 1.0
        self.last_grads = g
        return loss

    # models.py:1628-1635
    def predict(self, items, eps, remove_train=True):
        logits, mu, lv, _ = self.forward(items, eps)
        last = logits[-1].copy()
        if remove_train:
            last[np.asarray(items)] = -np.inf
        return last, mu, lv
