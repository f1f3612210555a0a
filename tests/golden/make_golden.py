#!/usr/bin/env python
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference (makgyver/rectorch)
from /root/reference in the build container.  The reference never travels to the GPU box; these
small .npz files (inputs + expected outputs) and this script do.

Run from the repo root:   python tests/golden/make_golden.py

Vector ids follow SURVEY.md §8c (G1..G9).  RNG-dependent quantities (dropout mask, eps) are captured
by seeded replay: `torch.manual_seed(s)` then `F.dropout(ones)` then `randn` reproduces exactly what
`MultiVAE_net.forward` consumes (dropout first, then randn_like; reference nets.py:394-411,
tests/test_nets.py:56-74).
"""
import os
import sys

sys.dont_write_bytecode = True          # keep /root/reference pristine
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tools", "ref_standins"))
sys.path.insert(0, "/root/reference")
sys.path.insert(0, ROOT)

import tempfile                          # noqa: E402
import numpy as np                       # noqa: E402
import torch                             # noqa: E402
import torch.nn.functional as F          # noqa: E402
from scipy.sparse import csr_matrix      # noqa: E402

from rectorch.nets import MultiVAE_net, MultiDAE_net, CMultiVAE_net, SVAE_net  # noqa: E402
from rectorch.models import MultiVAE, MultiDAE, EASE, CMultiVAE, SVAE  # noqa: E402
from rectorch.samplers import DataSampler, ConditionedDataSampler, BalancedConditionedDataSampler, \
    EmptyConditionedDataSampler, SVAE_Sampler                   # noqa: E402
from rectorch.evaluation import evaluate                        # noqa: E402
from rectorch.metrics import Metrics                            # noqa: E402

import importlib.util                    # noqa: E402


def _load(name, rel):
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


hashinit = _load("hashinit", "rectorch_amd/utils/hashinit.py")
synth = _load("synth", "rectorch_amd/utils/synth.py")

torch.set_num_threads(8)


def sd_np(net):
    return {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}


def load_hash(net, enc_dims, dec_dims, variant, seed, bias_std=1.0):
    sd = hashinit.hash_state_dict(enc_dims, dec_dims, variant, seed, bias_std)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    return sd


def replay_rng(seed, B, I, L, p):
    """dropout keep-mask (uint8) and eps exactly as the reference's forward would draw them."""
    torch.manual_seed(seed)
    m = F.dropout(torch.ones(B, I), p, True)
    eps = torch.randn(B, L) if L else None
    return (m != 0).numpy().astype(np.uint8), (eps.numpy() if eps is not None else None)


def small_x(B, I, seed):
    rng = np.random.default_rng(seed)
    x = (rng.random((B, I)) < 0.2).astype(np.float32)
    x[1, :] = 0.0                       # all-zero row
    x[2, :] = 0.0
    x[2, 7] = 1.0                       # single-item row
    return x


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs)
    print("wrote", path, os.path.getsize(path), "bytes")


def flat(prefix, d):
    return {prefix + k.replace(".", "__"): v for k, v in d.items()}


# ---------------------------------------------------------------- G1 / G7
def g1_g7():
    I, H, L, B = 64, 16, 8, 5
    net = MultiVAE_net([L, H, I], dropout=0.5)
    sd = load_hash(net, [I, H, L], [L, H, I], "vae", 11)
    x = small_x(B, I, 3)
    model = MultiVAE(net, beta=0.2, anneal_steps=0)
    net.eval()
    with torch.no_grad():
        y, mu, logvar = net(torch.from_numpy(x))
        loss = model.loss_function(y, torch.from_numpy(x), mu, logvar, 0.2)
    pred, pmu, plv = model.predict(torch.from_numpy(x), remove_train=True)
    pred_keep = model.predict(torch.from_numpy(x), remove_train=False)[0]
    save("g1_mvae_fwd_eval_small", x=x, logits=y.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
         loss=np.float32(loss.item()), beta=np.float32(0.2),
         dims=np.array([I, H, L]), **flat("sd__", sd))
    save("g7_predict_remove_train", x=x, pred=pred.numpy(), pred_keep=pred_keep.numpy(),
         mu=pmu.numpy(), logvar=plv.numpy(), n_neg_inf=np.int64(np.isneginf(pred.numpy()).sum()),
         dims=np.array([I, H, L]), **flat("sd__", sd))


# ---------------------------------------------------------------- G2
def train_steps_vae(name, enc_dims, dec_dims, B, xs, gts, beta, anneal_steps, p, seeds, hseed,
                    lr=1e-3, extra=None, cond_dim=0):
    I, L = enc_dims[0], enc_dims[-1]
    if cond_dim:
        net = CMultiVAE_net(cond_dim, list(dec_dims), list(enc_dims), dropout=p)
        sd0 = load_hash(net, [I + cond_dim] + list(enc_dims[1:]), dec_dims, "vae", hseed)
        model = CMultiVAE(net, beta=beta, anneal_steps=anneal_steps, learning_rate=lr)
    else:
        net = MultiVAE_net(list(dec_dims), list(enc_dims), dropout=p)
        sd0 = load_hash(net, enc_dims, dec_dims, "vae", hseed)
        model = MultiVAE(net, beta=beta, anneal_steps=anneal_steps, learning_rate=lr)
    out = dict(flat("sd0__", sd0))
    names = [k for k, _ in net.named_parameters()]
    for t, seed in enumerate(seeds):
        x = torch.from_numpy(xs[t])
        gt = None if gts is None else torch.from_numpy(gts[t])
        mask, eps = replay_rng(seed, x.shape[0], I, L, p)
        out["mask_%d" % t] = mask
        out["eps_%d" % t] = eps
        # gradients of this step (before the update) via a side computation with the same RNG
        net.train()
        torch.manual_seed(seed)
        ab = min(beta, model.gradient_updates / anneal_steps) if anneal_steps > 0 else beta
        y, mu, logvar = net(x)
        l_side = model.loss_function(y, x if gt is None else gt, mu, logvar, ab)
        grads = torch.autograd.grad(l_side, list(net.parameters()))
        for n, g in zip(names, grads):
            out["grad_%d__%s" % (t, n.replace(".", "__"))] = g.numpy().copy()
        out["logits_%d" % t] = y.detach().numpy().copy()
        out["mu_%d" % t] = mu.detach().numpy().copy()
        out["logvar_%d" % t] = logvar.detach().numpy().copy()
        out["anneal_beta_%d" % t] = np.float32(ab)
        # the real step
        torch.manual_seed(seed)
        loss = model.train_batch(x, gt)
        assert abs(loss - l_side.item()) <= 1e-6 * max(1.0, abs(loss)), (loss, l_side.item())
        out["loss_%d" % t] = np.float32(loss)
        out.update(flat("sd_%d__" % t, sd_np(net)))
        for i, prm in enumerate(net.parameters()):
            st = model.optimizer.state[prm]
            out["exp_avg_%d__%s" % (t, names[i].replace(".", "__"))] = st["exp_avg"].numpy().copy()
            out["exp_avg_sq_%d__%s" % (t, names[i].replace(".", "__"))] = st["exp_avg_sq"].numpy().copy()
    out["gradient_updates"] = np.float64(model.gradient_updates)
    out["xs"] = np.stack(xs)
    if gts is not None:
        out["gts"] = np.stack(gts)
    out["meta"] = np.array([beta, anneal_steps, p, lr], dtype=np.float64)
    out["enc_dims"] = np.array(enc_dims)
    out["dec_dims"] = np.array(dec_dims)
    if extra:
        out.update(extra)
    save(name, **out)


def g2():
    I, H, L, B = 64, 16, 8, 5
    xs = [small_x(B, I, 3), small_x(B, I, 4), small_x(B, I, 5)]
    train_steps_vae("g2_mvae_train_step_small", [I, H, L], [L, H, I], B, xs, None,
                    beta=0.2, anneal_steps=2, p=0.5, seeds=[1001, 1002, 1003], hseed=11)
    # te_batch != None: loss target differs from encoder input (reference models.py:819-822)
    gts = [small_x(B, I, 13), small_x(B, I, 14)]
    train_steps_vae("g2b_mvae_train_step_te", [I, H, L], [L, H, I], B, xs[:2], gts,
                    beta=1.0, anneal_steps=0, p=0.5, seeds=[2001, 2002], hseed=12)
    # odd sizes + deeper nets (nothing a multiple of any tile) + non-binary values
    I, B = 77, 7
    rng = np.random.default_rng(9)
    xs = [((rng.random((B, I)) < 0.15) * rng.integers(1, 4, (B, I))).astype(np.float32) for _ in range(2)]
    train_steps_vae("g2c_mvae_train_step_deep", [I, 21, 13, 5], [5, 9, I], B, xs, None,
                    beta=0.3, anneal_steps=0, p=0.3, seeds=[3001, 3002], hseed=13)


# ---------------------------------------------------------------- G3
def g3():
    I, H, L, B = 20108, 600, 200, 8
    X = synth.synth_interactions(4096, I, seed=777)
    rows = np.array([5, 17, 100, 1000, 2047, 3000, 4000, 4095])
    x = np.asarray(X[rows].toarray(), dtype=np.float32)
    net = MultiVAE_net([L, H, I], dropout=0.5)
    load_hash(net, [I, H, L], [L, H, I], "vae", 4242, bias_std=1.0)
    model = MultiVAE(net, beta=0.2, anneal_steps=0)
    net.eval()
    xt = torch.from_numpy(x)
    with torch.no_grad():
        y, mu, logvar = net(xt)
        loss = model.loss_function(y, xt, mu, logvar, 0.2)
        lse = torch.logsumexp(y, 1)
    # one training-mode step with injected RNG at the real K: loss + gradient slices
    seed = 5005
    mask, eps = replay_rng(seed, B, I, L, 0.5)
    net.train()
    torch.manual_seed(seed)
    yt, mut, lvt = net(xt)
    lt = model.loss_function(yt, xt, mut, lvt, 0.2)
    grads = torch.autograd.grad(lt, list(net.parameters()))
    names = [k for k, _ in net.named_parameters()]
    g = dict(zip(names, grads))
    save("g3_mvae_fwd_ml20m_slice",
         synth_seed=np.int64(777), synth_users=np.int64(4096), rows=rows, hash_seed=np.int64(4242),
         dims=np.array([I, H, L]), beta=np.float32(0.2),
         logits_s257=y.numpy()[:, ::257].copy(), lse=lse.numpy(), mu=mu.numpy(), logvar=logvar.numpy(),
         loss=np.float32(loss.item()),
         train_seed=np.int64(seed), mask_bits=np.packbits(mask, axis=1), eps=eps,
         train_loss=np.float32(lt.item()), train_logits_s257=yt.detach().numpy()[:, ::257].copy(),
         train_mu=mut.detach().numpy(), train_logvar=lvt.detach().numpy(),
         gW1_s=g["enc_layers.0.weight"].numpy()[::37, ::211].copy(),
         gb1=g["enc_layers.0.bias"].numpy(),
         gW2_s=g["enc_layers.1.weight"].numpy()[::7, ::11].copy(),
         gb2=g["enc_layers.1.bias"].numpy(),
         gW3_s=g["dec_layers.0.weight"].numpy()[::11, ::7].copy(),
         gb3=g["dec_layers.0.bias"].numpy(),
         gW4_s=g["dec_layers.1.weight"].numpy()[::211, ::37].copy(),
         gb4_s=g["dec_layers.1.bias"].numpy()[::101].copy())


# ---------------------------------------------------------------- G4
def g4():
    I, H, L, B = 64, 16, 8, 5
    p, lam = 0.5, 0.2
    net = MultiDAE_net([L, H, I], dropout=p)
    sd0 = load_hash(net, [I, H, L], [L, H, I], "dae", 21)
    model = MultiDAE(net, lam=lam)
    xs = [small_x(B, I, 3), small_x(B, I, 4), small_x(B, I, 5)]
    names = [k for k, _ in net.named_parameters()]
    out = dict(flat("sd0__", sd0))
    for t, seed in enumerate([4001, 4002, 4003]):
        x = torch.from_numpy(xs[t])
        mask, _ = replay_rng(seed, B, I, 0, p)
        out["mask_%d" % t] = mask
        net.train()
        torch.manual_seed(seed)
        y = net(x)
        l_side = model.loss_function(y, x)
        grads = torch.autograd.grad(l_side, list(net.parameters()))
        for n, g in zip(names, grads):
            out["grad_%d__%s" % (t, n.replace(".", "__"))] = g.numpy().copy()
        out["logits_%d" % t] = y.detach().numpy().copy()
        torch.manual_seed(seed)
        loss = model.train_batch(x, None)
        assert abs(loss - l_side.item()) <= 1e-6 * max(1.0, abs(loss))
        out["loss_%d" % t] = np.float32(loss)
        out.update(flat("sd_%d__" % t, sd_np(net)))
        for i, prm in enumerate(net.parameters()):
            st = model.optimizer.state[prm]
            out["exp_avg_%d__%s" % (t, names[i].replace(".", "__"))] = st["exp_avg"].numpy().copy()
            out["exp_avg_sq_%d__%s" % (t, names[i].replace(".", "__"))] = st["exp_avg_sq"].numpy().copy()
    net.eval()
    pred = model.predict(torch.from_numpy(xs[0]), remove_train=True)
    assert len(pred) == 1
    out["pred_after"] = pred[0].numpy()
    out["xs"] = np.stack(xs)
    out["meta"] = np.array([lam, 0.001, p, 1e-3], dtype=np.float64)
    out["enc_dims"] = np.array([I, H, L])
    out["dec_dims"] = np.array([L, H, I])
    save("g4_mdae_train_step_small", **out)


# ---------------------------------------------------------------- G5
def g5():
    rng = np.random.default_rng(5)
    dense = (rng.random((37, 23)) < 0.2).astype(np.float64)
    dense_te = (rng.random((37, 23)) < 0.1).astype(np.float64)
    tr, te = csr_matrix(dense), csr_matrix(dense_te)
    out = dict(dense_tr=dense, dense_te=dense_te)
    for tag, (shuffle, with_te) in {"ns": (False, False), "s": (True, False), "ste": (True, True)}.items():
        np.random.seed(424242)
        smp = DataSampler(tr, te if with_te else None, batch_size=8, shuffle=shuffle)
        out["len_" + tag] = np.int64(len(smp))
        for e in range(2):                      # two epochs: a fresh permutation each iter()
            for b, (dtr, dte) in enumerate(smp):
                out["%s_e%d_b%d_tr" % (tag, e, b)] = dtr.numpy()
                if dte is not None:
                    out["%s_e%d_b%d_te" % (tag, e, b)] = dte.numpy()
                else:
                    assert not with_te
    out["np_seed"] = np.int64(424242)
    save("g5_sampler_batches", **out)


# ---------------------------------------------------------------- G6
def g6():
    rng = np.random.default_rng(6)
    scores = rng.standard_normal((16, 300)).astype(np.float32)
    train = rng.random((16, 300)) < 0.1
    scores[train] = -np.inf
    heldout = ((rng.random((16, 300)) < 0.05) & ~train).astype(np.float32)
    heldout[0, :] = 0; heldout[0, 5] = 1          # single relevant item
    heldout[3, ~train[3]] = 1                    # more relevant items than k
    mets = ["ndcg@100", "ndcg@10", "recall@50", "recall@20", "hit@5", "mrr@10", "ndcg@1000"]
    res = Metrics.compute(scores, heldout, mets)
    out = dict(scores=scores, heldout=heldout)
    for m in mets:
        out["res__" + m.replace("@", "_at_")] = np.asarray(res[m], dtype=np.float64)
    # the reference's own KATs (tests/test_metrics.py:14-61), outputs as produced by the reference
    s = np.array([[4., 3., 2., 1.]]); gt1 = np.array([[1., 1., 0., 0.]]); gt2 = np.array([[0., 0., 1., 1.]])
    out["kat_ndcg2_a"] = Metrics.ndcg_at_k(s, gt1, 2); out["kat_ndcg2_b"] = Metrics.ndcg_at_k(s, gt2, 2)
    out["kat_ndcg3_b"] = Metrics.ndcg_at_k(s, gt2, 3)
    out["kat_recall3_a"] = Metrics.recall_at_k(np.array([[4., 3., 2., 1., 0.]]), np.array([[1., 1., 0., 0., 1.]]), 3)
    save("g6_metrics", **out)


# ---------------------------------------------------------------- G8
def g8():
    U, I, H, L, B = 512, 128, 32, 8, 64
    rng = np.random.default_rng(8)
    # low-rank synthetic preferences
    P, Q = rng.standard_normal((U + 64, 4)), rng.standard_normal((I, 4))
    S = P @ Q.T + 0.5 * rng.standard_normal((U + 64, I))
    dense = (S > np.quantile(S, 0.85, axis=1, keepdims=True)).astype(np.float64)
    train = csr_matrix(dense[:U])
    held = csr_matrix(dense[U:])
    val_tr, val_te = synth.split_heldout(held, 0.2, seed=1)
    p = 0.5
    net = MultiVAE_net([L, H, I], dropout=p)
    sd0 = load_hash(net, [I, H, L], [L, H, I], "vae", 88, bias_std=0.1)
    model = MultiVAE(net, beta=0.2, anneal_steps=20)
    n_epochs = 5
    nb = int(np.ceil(U / B))
    masks = np.zeros((n_epochs, nb, B, I), dtype=np.uint8)
    epss = np.zeros((n_epochs, nb, B, L), dtype=np.float32)
    losses = np.zeros((n_epochs, nb), dtype=np.float64)
    ndcg = np.zeros(n_epochs); recall = np.zeros(n_epochs)
    perms = np.zeros((n_epochs, U), dtype=np.int64)
    for e in range(n_epochs):
        np.random.seed(8000 + e)
        perm = list(range(U)); np.random.shuffle(perm); perms[e] = perm
        np.random.seed(8000 + e)
        smp = DataSampler(train, batch_size=B, shuffle=True)
        net.train()
        for b, (dtr, _) in enumerate(smp):
            assert np.array_equal(dtr.numpy(), dense[:U][perm[b * B:(b + 1) * B]].astype(np.float32))
            seed = 80000 + e * 100 + b
            m, eps = replay_rng(seed, dtr.shape[0], I, L, p)
            masks[e, b], epss[e, b] = m, eps
            torch.manual_seed(seed)
            losses[e, b] = model.train_batch(dtr, None)
        vs = DataSampler(val_tr, val_te, batch_size=32, shuffle=False)
        res = evaluate(model, vs, ["ndcg@100", "recall@50"])
        ndcg[e], recall[e] = np.mean(res["ndcg@100"]), np.mean(res["recall@50"])
        if e == n_epochs - 1:
            last = res
    save("g8_epoch_curve", dense_train=dense[:U].astype(np.uint8),
         val_tr=np.asarray(val_tr.toarray(), dtype=np.uint8), val_te=np.asarray(val_te.toarray(), dtype=np.uint8),
         mask_bits=np.packbits(masks, axis=-1), eps=epss, losses=losses, ndcg100=ndcg, recall50=recall,
         perms=perms, ndcg100_users=last["ndcg@100"], recall50_users=last["recall@50"],
         dims=np.array([I, H, L]), meta=np.array([0.2, 20, p, 1e-3, B]), hash_seed=np.int64(88),
         **flat("sd_final__", sd_np(net)))


# ---------------------------------------------------------------- G9
def g9():
    I, H, L = 12, 6, 3
    torch.manual_seed(90)                # the reference's own random initialisation, reproducibly
    net = MultiVAE_net([L, H, I], dropout=0.5)
    model = MultiVAE(net, beta=0.2, anneal_steps=5)
    torch.manual_seed(9)
    model.train_batch(torch.from_numpy(small_x(4, I, 9)), None)
    tmp = tempfile.NamedTemporaryFile(suffix=".pth")
    model.save_model(tmp.name, 3)
    ck = torch.load(tmp.name, weights_only=False)
    keys = sorted(ck.keys())
    sd_keys = list(ck["state_dict"].keys())
    sd_shapes = [tuple(v.shape) for v in ck["state_dict"].values()]
    opt = ck["optimizer"]
    pg = {k: v for k, v in opt["param_groups"][0].items() if k != "params"}
    st0 = opt["state"][0]
    save("g9_checkpoint_layout", top_keys=np.array(keys), sd_keys=np.array(sd_keys),
         sd_shapes=np.array([str(s) for s in sd_shapes]),
         sd_dtypes=np.array([str(v.dtype) for v in ck["state_dict"].values()]),
         pg_keys=np.array(sorted(pg.keys())), pg_vals=np.array([str(pg[k]) for k in sorted(pg.keys())]),
         pg_params=np.array(opt["param_groups"][0]["params"]),
         state_keys=np.array(sorted(st0.keys())), state_step=np.float64(float(st0["step"])),
         state_step_type=np.array(str(type(st0["step"]))),
         epoch=np.int64(ck["epoch"]), gradient_updates=np.float64(ck["gradient_updates"]))
    # a full reference-written checkpoint the build must be able to LOAD (binary data file)
    import shutil
    shutil.copy(tmp.name, os.path.join(HERE, "g9_reference_checkpoint.pth"))
    x = small_x(4, I, 10)
    pred = model.predict(torch.from_numpy(x), remove_train=True)[0].numpy()
    save("g9_checkpoint_predict", x=x, pred=pred, dims=np.array([I, H, L]))
    # resume semantics (reference models.py:496-516, 905-908): a FRESH reference model loads the checkpoint file and takes
    # one more training step -- the loaded Adam moments, step count and gradient_updates (-> annealed beta) all enter it
    net2 = MultiVAE_net([L, H, I], dropout=0.5)
    model2 = MultiVAE(net2, beta=0.2, anneal_steps=5)
    ck2 = model2.load_model(os.path.join(HERE, "g9_reference_checkpoint.pth"))
    xr = small_x(4, I, 11)
    mask, eps = replay_rng(21, 4, I, L, 0.5)
    torch.manual_seed(21)
    loss = model2.train_batch(torch.from_numpy(xr), None)
    opt2 = model2.optimizer.state_dict()
    save("g9_resume_step", x=xr, mask=mask, eps=eps, loss=np.float64(loss), epoch=np.int64(ck2["epoch"]),
         gradient_updates_after=np.float64(model2.gradient_updates),
         step_after=np.float64(float(opt2["state"][0]["step"])),
         **flat("sd__", sd_np(net2)),
         **{"exp_avg_%d" % k: v["exp_avg"].numpy().copy() for k, v in opt2["state"].items()},
         **{"exp_avg_sq_%d" % k: v["exp_avg_sq"].numpy().copy() for k, v in opt2["state"].items()})


def g10():
    """EASE (reference models.py:1003-1069): closed-form fit, predict with/without remove_train, saved model file."""
    rng = np.random.RandomState(10)
    # (a) implicit feedback, fewer users than items (rank-deficient Gram matrix, lam makes it SPD); 3 blocks of 128
    U, I = 200, 300
    Xa = (rng.rand(U, I) < 0.08).astype(np.float64)
    ease = EASE(200.)
    ease.train(csr_matrix(Xa))
    ids = np.array([3, 17, 17, 199, 0, 42])
    te = Xa[ids].copy()
    te[1, :] = 0                      # a user with an empty fold-in row
    pr_rm = ease.predict(ids, csr_matrix(te))[0].copy()
    pr_keep = ease.predict(ids, csr_matrix(te), remove_train=False)[0].copy()
    tmp = tempfile.NamedTemporaryFile()
    ease.save_model(tmp.name)
    import shutil
    shutil.copy(tmp.name + ".npy", os.path.join(HERE, "g10_reference_ease_model.npy"))
    os.remove(tmp.name + ".npy")
    save("g10_ease_binary", X=Xa.astype(np.uint8), lam=np.float64(200.), model=ease.model, ids=ids, te=te.astype(np.uint8),
         pred_remove=pr_rm, pred_keep=pr_keep, str_trained=np.array(str(ease)), str_new=np.array(str(EASE(200.))))
    # (b) explicit ratings 1..5 (f64 Gram path), one block
    U, I = 90, 70
    Xb = (rng.rand(U, I) < 0.2) * rng.randint(1, 6, size=(U, I))
    Xb = Xb.astype(np.float64) * 0.5          # half stars: not integers -> no exact bf16 path
    ease = EASE(50.)
    ease.train(csr_matrix(Xb))
    save("g10_ease_ratings", X=Xb, lam=np.float64(50.), model=ease.model)


def g11():
    """CMultiVAE (reference nets.py:420-480, models.py:911-956) and the conditioned samplers (samplers.py:108-419)."""
    I, C, H, L, B = 64, 4, 16, 8, 6
    rng = np.random.default_rng(11)
    # (a) eval forward + predict
    net = CMultiVAE_net(C, [L, H, I], dropout=0.5)
    sd = load_hash(net, [I + C, H, L], [L, H, I], "vae", 21)
    x = small_x(B, I, 5)
    cond = np.zeros((B, C), dtype=np.float32)
    cond[0, 1] = cond[3, 0] = cond[4, 3] = cond[2, 2] = 1.0        # rows 1 and 5 unconditioned
    xc = np.concatenate([x, cond], axis=1)
    model = CMultiVAE(net, beta=0.3)
    net.eval()
    with torch.no_grad():
        y, mu, logvar = net(torch.from_numpy(xc))
    pred = model.predict(torch.from_numpy(xc), remove_train=True)[0].numpy()
    save("g11_cmvae_fwd_eval", x=xc, logits=y.numpy(), mu=mu.numpy(), logvar=logvar.numpy(), pred=pred,
         dims=np.array([I, H, L]), cond_dim=np.int64(C), **flat("sd__", sd))
    # (b) three training steps with annealing, filtered targets
    xs, gts = [], []
    for t in range(3):
        xi = small_x(B, I, 30 + t)
        ci = np.zeros((B, C), dtype=np.float32)
        ci[np.arange(B), rng.integers(0, C, size=B)] = 1.0
        ci[t, :] = 0.0
        gi = xi * (rng.random((B, I)) < 0.7)
        gi[0, 5] = 1.0
        xs.append(np.concatenate([xi, ci], axis=1))
        gts.append(gi.astype(np.float32))
    train_steps_vae("g11_cmvae_train_steps", [I, H, L], [L, H, I], B, xs, gts, beta=0.3, anneal_steps=2, p=0.5,
                    seeds=[41, 42, 43], hseed=22, cond_dim=C)
    # (c) samplers
    U, NI, NC = 23, 17, 5
    tr = csr_matrix((rng.random((U, NI)) < 0.25).astype(np.float64))
    for u in range(U):
        if tr[u].nnz == 0:
            tr[u, u % NI] = 1.0
    tr = csr_matrix(tr)
    te = csr_matrix((rng.random((U, NI)) < 0.3).astype(np.float64))
    iid2cids = {i: sorted(set(rng.integers(0, NC, size=1 + i % 3).tolist())) for i in range(NI)}
    out = {"tr": tr.toarray(), "te": te.toarray(), "n_cond": np.int64(NC),
           "iid2cids_items": np.array([i for i in iid2cids for _ in iid2cids[i]]),
           "iid2cids_conds": np.array([c for i in iid2cids for c in iid2cids[i]])}
    s0 = ConditionedDataSampler(iid2cids, NC, tr, te, batch_size=7, shuffle=False)
    out["cds_examples"] = s0.examples.copy()
    out["cds_len"] = np.int64(len(s0))
    np.random.seed(5)
    s1 = ConditionedDataSampler(iid2cids, NC, tr, te, batch_size=7, shuffle=True)
    for i, (a, b) in enumerate(s1):
        out["cds_tr_%d" % i] = a.numpy()
        out["cds_te_%d" % i] = b.numpy()
    out["cds_n_batches"] = np.int64(i + 1)
    np.random.seed(6)
    s2 = BalancedConditionedDataSampler(iid2cids, NC, tr, None, batch_size=9, subsample=0.3)
    out["bal_examples"] = s2.examples.copy()
    out["bal_len"] = np.int64(len(s2))
    np.random.seed(7)
    for i, (a, b) in enumerate(s2):
        out["bal_tr_%d" % i] = a.numpy()
        out["bal_te_%d" % i] = b.numpy()
    out["bal_n_batches"] = np.int64(i + 1)
    np.random.seed(8)
    s3 = EmptyConditionedDataSampler(NC, tr, te, batch_size=10, shuffle=True)
    for i, (a, b) in enumerate(s3):
        out["emp_tr_%d" % i] = a.numpy()
        out["emp_te_%d" % i] = b.numpy()
    out["emp_n_batches"] = np.int64(i + 1)
    save("g11_conditioned_samplers", **out)


def g12():
    """SVAE (reference nets.py:624-693, models.py:1581-1635, samplers.py:446-571): forward / predict with replayed
    noise, three train_batch steps (loss, first-step gradients, parameters, Adam moments), sampler targets."""
    I, E, R, H, L = 40, 6, 10, 8, 4
    torch.manual_seed(120)
    net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, 12, I], enc_dims=[R, H, L])
    sd0 = sd_np(net)
    model = SVAE(net, beta=0.4, anneal_steps=2)
    out = dict(flat("sd0__", sd0))
    out["dims"] = np.array([I, E, R, H, L, 12])
    out["meta"] = np.array([0.4, 2, 1e-3, 5e-3], dtype=np.float64)
    out["param_names"] = np.array([k for k, _ in net.named_parameters()])
    rng = np.random.default_rng(12)
    seqs = {0: rng.integers(0, I, size=9).tolist(), 1: rng.integers(0, I, size=14).tolist(),
            2: [3, 7, 3, 11, 7, 2]}                                   # repeated items: embedding-gradient accumulation
    sampler = SVAE_Sampler(num_items=I, dict_data_tr=seqs, dict_data_te=None, pred_type="next_k", k=2, shuffle=False,
                           is_training=True)
    batches = [(x.clone(), y.clone()) for x, y in sampler]
    # predict before training, noise replayed from the seed (the VAE head samples in eval too)
    xq = torch.LongTensor([[5, 1, 30, 5]])
    torch.manual_seed(7)
    eps_q = torch.randn(4, L).numpy()
    torch.manual_seed(7)
    pr, pmu, plv = model.predict(xq, remove_train=True)
    out.update(pred_x=xq.numpy(), pred_eps=eps_q, pred=pr.numpy(), pred_mu=pmu.numpy(), pred_logvar=plv.numpy())
    names = [k for k, _ in net.named_parameters()]
    for t, (x, y) in enumerate(batches):
        T = x.shape[1]
        seed = 200 + t
        torch.manual_seed(seed)
        eps = torch.randn(T, L).numpy()
        out["x_%d" % t] = x.numpy()
        out["y_%d" % t] = y.numpy()
        out["eps_%d" % t] = eps
        ab = min(model.beta, model.gradient_updates / model.anneal_steps)
        torch.manual_seed(seed)
        net.train()
        recon, mu, logvar = net(x)
        # train_batch hands the target flattened to [1, T * n_items] (models.py:822), so likelihood_d (models.py:1623)
        # counts the ones of the FIRST time step only
        l_side = model.loss_function(recon, y.view(1, -1), mu, logvar, ab)
        grads = torch.autograd.grad(l_side, list(net.parameters()))
        for n, g in zip(names, grads):
            out["grad_%d__%s" % (t, n.replace(".", "__"))] = g.numpy().copy()
        out["logits_%d" % t] = recon.detach().numpy().copy()
        out["likelihood_d_%d" % t] = np.float64(float(torch.sum(y.view(1, -1)[0, :I])))
        torch.manual_seed(seed)
        loss = model.train_batch(x, y)
        assert abs(loss - l_side.item()) <= 1e-6 * max(1.0, abs(loss)), (loss, l_side.item())
        out["loss_%d" % t] = np.float32(loss)
        out["anneal_beta_%d" % t] = np.float32(ab)
        out.update(flat("sd_%d__" % t, sd_np(net)))
    out["n_steps"] = np.int64(len(batches))
    save("g12_svae_steps", **out)
    # sampler targets for the three prediction types and the test mode
    so = {}
    for pt in ("next", "next_k", "postfix"):
        smp = SVAE_Sampler(num_items=I, dict_data_tr=seqs, dict_data_te=None, pred_type=pt, k=3, shuffle=False, is_training=True)
        for u, (x, y) in enumerate(smp):
            so["%s_x_%d" % (pt, u)] = x.numpy()
            so["%s_y_%d" % (pt, u)] = y.numpy().astype(np.uint8)
    te = {0: [1, 2], 1: [39], 2: [0, 5, 9]}
    smp = SVAE_Sampler(num_items=I, dict_data_tr=seqs, dict_data_te=te, pred_type="next_k", k=1, shuffle=False, is_training=False)
    for u, (x, y) in enumerate(smp):
        so["test_x_%d" % u] = x.numpy()
        so["test_y_%d" % u] = y.numpy().astype(np.uint8)
    np.random.seed(3)
    smp = SVAE_Sampler(num_items=I, dict_data_tr=seqs, dict_data_te=None, pred_type="next", shuffle=True, is_training=True)
    so["shuffled_first_items"] = np.array([int(x[0, 0]) for x, _ in smp])
    for u in seqs:
        so["seq_%d" % u] = np.array(seqs[u])
    save("g12_svae_sampler", **so)


def g13():
    """On-disk formats: the preprocessed files the reference's DataProcessing.process writes (data.py:199-219:
    train.csv, {validation,test}_{tr,te}.csv with header uid,iid[,rating,timestamp], unique_{iid,uid}.txt) read back by the
    reference's own DataReader / DatasetManager (data.py:312-557).  DataProcessing itself does not run with the pandas of
    this image, so the small fixture files are written here, in that layout."""
    import json
    import pandas as pd
    from rectorch.data import DataReader, DatasetManager
    rng = np.random.RandomState(13)
    NI = 40
    pdir = os.path.join(HERE, "g13_preproc")
    os.makedirs(pdir, exist_ok=True)

    def users(lo, hi):
        rows = []
        for u in range(lo, hi):
            n = rng.randint(4, 15)
            for i in rng.choice(NI, size=n, replace=False):
                rows.append((u, int(i), float(rng.randint(1, 6)), int(rng.randint(1e6, 2e6))))
        return pd.DataFrame(rows, columns=["uid", "iid", "rating", "timestamp"])

    def split(df):
        tr, te = [], []
        for _, g in df.groupby("uid"):
            k = max(int(0.2 * len(g)), 1)
            idx = np.zeros(len(g), bool)
            idx[rng.choice(len(g), k, replace=False)] = True
            tr.append(g[~idx])
            te.append(g[idx])
        return pd.concat(tr), pd.concat(te)

    users(0, 44).to_csv(os.path.join(pdir, "train.csv"), index=False)
    va = users(44, 52)
    va = va[va["uid"] != 47]                                  # a user id missing from the validation block
    a, b = split(va)
    a.to_csv(os.path.join(pdir, "validation_tr.csv"), index=False)
    b.to_csv(os.path.join(pdir, "validation_te.csv"), index=False)
    a, b = split(users(52, 60))
    a.to_csv(os.path.join(pdir, "test_tr.csv"), index=False)
    b.to_csv(os.path.join(pdir, "test_te.csv"), index=False)
    with open(os.path.join(pdir, "unique_iid.txt"), "w") as f:
        f.write("".join("%d\n" % (500 + i) for i in range(NI)))
    with open(os.path.join(pdir, "unique_uid.txt"), "w") as f:
        f.write("".join("%d\n" % (100 + u) for u in range(60)))
    out = {}
    for topn in (1, 0):
        cfgp = os.path.join(tempfile.gettempdir(), "g13_cfg%d.json" % topn)
        json.dump({"proc_path": pdir, "topn": topn, "seed": 98765, "test_prop": 0.2}, open(cfgp, "w"))
        r = DataReader(cfgp)
        k = "topn%d_" % topn
        out[k + "n_items"] = np.int64(r.n_items)
        out[k + "train"] = r.load_data("train").toarray()
        for dt in ("validation", "test"):
            x, y = r.load_data(dt)
            out[k + dt + "_tr"] = x.toarray()
            out[k + dt + "_te"] = y.toarray()
        out[k + "full"] = r.load_data("full").toarray()
        dm = DatasetManager(cfgp)
        x, y = dm.get_train_and_test()
        out[k + "tt_tr"] = x.toarray()
        out[k + "tt_te"] = y.toarray()
        if topn:
            def pack(d):
                keys = sorted(d)
                return np.array(keys), np.array([len(d[u]) for u in keys]), np.array([i for u in keys for i in d[u]])
            for name, d in (("dict_train", r.load_data_as_dict("train")), ("dict_full", r.load_data_as_dict("full"))):
                out[name + "_keys"], out[name + "_lens"], out[name + "_items"] = pack(d)
            for dt in ("validation", "test"):
                d1, d2 = r.load_data_as_dict(dt)
                out["dict_%s_tr_keys" % dt], out["dict_%s_tr_lens" % dt], out["dict_%s_tr_items" % dt] = pack(d1)
                out["dict_%s_te_keys" % dt], out["dict_%s_te_lens" % dt], out["dict_%s_te_items" % dt] = pack(d2)
    save("g13_data_reader", **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g8", "g9", "g10", "g11", "g12", "g13"]
    for w in which:
        {"g1": g1_g7, "g2": g2, "g3": g3, "g4": g4, "g5": g5, "g6": g6, "g8": g8, "g9": g9, "g10": g10, "g11": g11, "g12": g12, "g13": g13}[w]()
