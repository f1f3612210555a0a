"""GPU parity tests (run with -m gpu on the MI355X box): the HIP path, called through the Python mirror of the
reference API and the C ABI, against (a) golden vectors generated from the reference itself and (b) the CPU
oracle on the same seeded inputs, plus size-independent properties at the full ml-20m shape.

Tolerances.  fp32 ("parity") mode: logits within 1e-5 relative to max|logits| (the north star's criterion),
gradients 2e-4, parameters after Adam 2e-6 absolute.  bf16 mode: operands rounded to 8 bits of mantissa with
f32 accumulation -> logits ~1e-2 relative; gated by nDCG@100 / Recall@50 parity instead.
"""
import os
import random
import subprocess
import tempfile

import numpy as np
import pytest
import torch
from scipy.sparse import csr_matrix

from conftest import ROOT, load_golden, sd_from, params_in_order

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b)) / max(1e-30, np.max(np.abs(b))))


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", dtype)


@pytest.fixture(scope="module", autouse=True)
def _require_native_path():
    from rectorch_amd import _lib
    assert torch.cuda.is_available(), "these tests need the MI355X"
    _lib.lib()          # raises if librectorch_hip.so is missing: no fallback


def make_vae(enc, dec, p, sd, **kw):
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.models import MultiVAE
    net = MultiVAE_net(list(dec), list(enc), dropout=p)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net, MultiVAE(net, **kw)


def make_dae(enc, dec, p, sd, **kw):
    from rectorch_amd.nets import MultiDAE_net
    from rectorch_amd.models import MultiDAE
    net = MultiDAE_net(list(dec), list(enc), dropout=p)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net, MultiDAE(net, **kw)


# ---------------------------------------------------------------------------------------------- native drivers
@pytest.mark.parametrize("exe", ["test_gemm", "test_spmm", "test_engine", "test_potf2"])
def test_native_driver(exe):
    """the no-Python drivers: MFMA GEMM vs host double loops; the sparse first layer vs a host loop over the same stored
    entries; the whole engine vs the C oracle via the C ABI; the EASE solver's leaf (both versions) vs a host Cholesky + inverse"""
    path = os.path.join(ROOT, "build", "native", exe)
    if not os.path.exists(path):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "tests", "native")])
    out = subprocess.run([path], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-2000:]
    assert "PASSED" in out.stdout


# ---------------------------------------------------------------------------------------------- golden: forward
def test_g1_eval_forward_logits_1e5():
    g = load_golden("g1_mvae_fwd_eval_small")
    I, H, L = [int(v) for v in g["dims"]]
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd_from(g, "sd__"), beta=0.2)
    net.eval()
    y, mu, logvar = net(torch.from_numpy(g["x"]))
    assert y.is_cuda and y.shape == (5, I) and mu.shape == (5, L)
    assert rel(y.cpu(), g["logits"]) < 1e-5
    assert rel(mu.cpu(), g["mu"]) < 1e-5 and rel(logvar.cpu(), g["logvar"]) < 1e-5
    mu2, lv2 = net.encode(torch.from_numpy(g["x"]))
    assert torch.equal(mu, mu2) and torch.equal(logvar, lv2)
    assert rel(net.decode(mu).cpu(), g["logits"]) < 1e-5
    loss = model.loss_function(y, dev(g["x"]), mu, logvar, float(g["beta"]))
    assert loss.dim() == 0 and abs(loss.item() - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


def test_g7_predict_remove_train():
    g = load_golden("g7_predict_remove_train")
    I, H, L = [int(v) for v in g["dims"]]
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd_from(g, "sd__"))
    x = torch.from_numpy(g["x"])
    pred, mu, logvar = model.predict(x, remove_train=True)
    p = pred.cpu().numpy()
    assert np.array_equal(np.isneginf(p), np.isneginf(g["pred"]))
    assert int(np.isneginf(p).sum()) == int(g["n_neg_inf"])
    fin = np.isfinite(g["pred"])
    assert rel(p[fin], g["pred"][fin]) < 1e-5
    assert rel(model.predict(x, remove_train=False)[0].cpu(), g["pred_keep"]) < 1e-5
    assert np.array_equal(x.numpy(), g["x"])        # the caller's x is not modified
    assert not net.training                         # predict() leaves the net in eval mode, like the reference


# ---------------------------------------------------------------------------------------------- golden: training
@pytest.mark.parametrize("name", ["g2_mvae_train_step_small", "g2b_mvae_train_step_te", "g2c_mvae_train_step_deep"])
def test_g2_train_steps_fp32(name):
    """3 Adam steps with the reference's own dropout masks / eps injected: loss, grads, params, moments"""
    g = load_golden(name)
    enc, dec = [int(v) for v in g["enc_dims"]], [int(v) for v in g["dec_dims"]]
    beta, anneal, p, lr = [float(v) for v in g["meta"]]
    net, model = make_vae(enc, dec, p, sd_from(g, "sd0__"), beta=beta, anneal_steps=int(anneal), learning_rate=lr,
                          numerics="fp32")
    _, keys = params_in_order(sd_from(g, "sd0__"))
    for t in range(g["xs"].shape[0]):
        model._rtx.inject = (dev(g["mask_%d" % t], torch.uint8), dev(g["eps_%d" % t]))
        gt = torch.from_numpy(g["gts"][t]) if "gts" in g else None
        loss = model.train_batch(torch.from_numpy(g["xs"][t]), gt)
        assert isinstance(loss, float)
        assert abs(loss - float(g["loss_%d" % t])) < 1e-5 * abs(float(g["loss_%d" % t])), (t, loss)
        sd_t, _ = params_in_order(sd_from(g, "sd_%d__" % t))
        for k, prm, ref in zip(keys, net._param_list(), sd_t):
            gref = g["grad_%d__%s" % (t, k.replace(".", "__"))]
            assert rel(prm.grad.cpu(), gref) < 2e-4, (t, k)
            assert float(np.max(np.abs(prm.detach().cpu().numpy() - ref))) < 5e-6, (t, k)
            st = model.optimizer.state[prm]
            assert rel(st["exp_avg"].cpu(), g["exp_avg_%d__%s" % (t, k.replace(".", "__"))]) < 2e-4
            assert rel(st["exp_avg_sq"].cpu(), g["exp_avg_sq_%d__%s" % (t, k.replace(".", "__"))]) < 4e-4
    assert model.gradient_updates == float(g["gradient_updates"])


def test_g4_dae_train_steps_fp32():
    g = load_golden("g4_mdae_train_step_small")
    enc, dec = [int(v) for v in g["enc_dims"]], [int(v) for v in g["dec_dims"]]
    lam, wd, p, lr = [float(v) for v in g["meta"]]
    net, model = make_dae(enc, dec, p, sd_from(g, "sd0__"), lam=lam, learning_rate=lr, numerics="fp32")
    assert model.optimizer.param_groups[0]["weight_decay"] == wd
    _, keys = params_in_order(sd_from(g, "sd0__"))
    for t in range(3):
        model._rtx.inject = (dev(g["mask_%d" % t], torch.uint8), None)
        loss = model.train_batch(torch.from_numpy(g["xs"][t]))
        assert abs(loss - float(g["loss_%d" % t])) < 1e-5 * abs(float(g["loss_%d" % t]))
        sd_t, _ = params_in_order(sd_from(g, "sd_%d__" % t))
        for k, prm, ref in zip(keys, net._param_list(), sd_t):
            assert float(np.max(np.abs(prm.detach().cpu().numpy() - ref))) < 5e-6, (t, k)
    pred = model.predict(torch.from_numpy(g["xs"][0]), True)
    assert isinstance(pred, tuple) and len(pred) == 1
    fin = np.isfinite(g["pred_after"])
    assert rel(pred[0].cpu().numpy()[fin], g["pred_after"][fin]) < 1e-4
    x0 = dev(g["xs"][0])
    lf = model.loss_function(model.predict(x0, False)[0], x0)
    assert lf.dim() == 0 and np.isfinite(lf.item())


# ---------------------------------------------------------------------------------------------- golden: real K = 20108
def _g3_setup(numerics):
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    g = load_golden("g3_mvae_fwd_ml20m_slice")
    I, H, L = [int(v) for v in g["dims"]]
    X = synth_interactions(int(g["synth_users"]), I, seed=int(g["synth_seed"]))
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", int(g["hash_seed"]))
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=float(g["beta"]), numerics=numerics, predict_numerics=numerics)
    x = torch.from_numpy(np.asarray(X[g["rows"]].toarray(), dtype=np.float32))
    return g, I, H, L, net, model, x


def test_g3_full_size_forward_fp32_1e5():
    """the 1e-5 logits criterion at the real contraction length (K = 20108 and K = 600)"""
    g, I, H, L, net, model, x = _g3_setup("fp32")
    net.eval()
    y, mu, logvar = net(x)
    assert rel(y.cpu().numpy()[:, ::257], g["logits_s257"]) < 1e-5
    assert rel(mu.cpu(), g["mu"]) < 1e-5 and rel(logvar.cpu(), g["logvar"]) < 1e-5
    lse = torch.logsumexp(y.double(), 1).cpu().numpy()
    assert np.max(np.abs(lse - g["lse"])) < 2e-5
    loss = model.loss_function(y, x.cuda(), mu, logvar, float(g["beta"])).item()
    assert abs(loss - float(g["loss"])) < 1e-5 * abs(float(g["loss"]))


def test_g3_full_size_training_step_fp32():
    g, I, H, L, net, model, x = _g3_setup("fp32")
    mask = np.unpackbits(g["mask_bits"], axis=1)[:, :I]
    model._rtx.inject = (dev(mask, torch.uint8), dev(g["eps"]))
    model.keep_grads = True            # W1/W4 take the fused dW+Adam path: ask for their gradients too
    loss = model.train_batch(x)
    assert abs(loss - float(g["train_loss"])) < 1e-5 * abs(float(g["train_loss"]))
    ps = net._param_list()
    assert rel(ps[0].grad.cpu().numpy()[::37, ::211], g["gW1_s"]) < 2e-4
    assert rel(ps[1].grad.cpu(), g["gb1"]) < 2e-4
    assert rel(ps[2].grad.cpu().numpy()[::7, ::11], g["gW2_s"]) < 2e-4
    assert rel(ps[3].grad.cpu(), g["gb2"]) < 2e-4
    assert rel(ps[4].grad.cpu().numpy()[::11, ::7], g["gW3_s"]) < 2e-4
    assert rel(ps[5].grad.cpu(), g["gb3"]) < 2e-4
    assert rel(ps[6].grad.cpu().numpy()[::211, ::37], g["gW4_s"]) < 2e-4
    assert rel(ps[7].grad.cpu().numpy()[::101], g["gb4_s"]) < 2e-4


def test_g3_full_size_bf16_within_stated_tolerance():
    g, I, H, L, net, model, x = _g3_setup("bf16")
    net.eval()
    eng = net.rtx_engine("bf16", x.shape[0])
    y, mu, logvar = eng.forward(net._as_input(x))
    e_logits = rel(y.cpu().numpy()[:, ::257], g["logits_s257"])
    print("bf16 logits rel err at K=20108: %.3e" % e_logits)
    assert e_logits < 2e-3          # achieved on MI355X: 5.0e-4 (round 2 asserted 3e-2)
    mask = np.unpackbits(g["mask_bits"], axis=1)[:, :I]
    model._rtx.inject = (dev(mask, torch.uint8), dev(g["eps"]))
    model.keep_grads = True
    loss = model.train_batch(x)
    ps = net._param_list()
    e_loss = abs(loss - float(g["train_loss"])) / abs(float(g["train_loss"]))
    e_w4, e_w1 = rel(ps[6].grad.cpu().numpy()[::211, ::37], g["gW4_s"]), rel(ps[0].grad.cpu().numpy()[::37, ::211], g["gW1_s"])
    print("bf16 train step at K=20108: loss rel %.2e, gW4 rel %.2e, gW1 rel %.2e" % (e_loss, e_w4, e_w1))
    # achieved on MI355X (round 3): loss 8.9e-7, gW4 4.8e-3, gW1 2.8e-3 relative (round 2 asserted 5e-3 / 5e-2 / 5e-2)
    assert e_loss < 1e-5
    assert e_w4 < 1.5e-2
    assert e_w1 < 1e-2


# ---------------------------------------------------------------------------------------------- sampler on the device
def test_g5_resident_sampler_matches_reference_batches():
    from rectorch_amd.samplers import DataSampler
    g = load_golden("g5_sampler_batches")
    tr, te = csr_matrix(g["dense_tr"]), csr_matrix(g["dense_te"])
    for tag, (shuffle, with_te) in {"ns": (False, False), "s": (True, False), "ste": (True, True)}.items():
        np.random.seed(int(g["np_seed"]))
        smp = DataSampler(tr, te if with_te else None, batch_size=8, shuffle=shuffle)
        assert smp.resident and len(smp) == 5
        for e in range(2):
            for b, (dtr, dte) in enumerate(smp):
                assert dtr.is_cuda and dtr.dtype == torch.float32
                assert np.array_equal(dtr.cpu().numpy(), g["%s_e%d_b%d_tr" % (tag, e, b)])
                if with_te:
                    assert np.array_equal(dte.cpu().numpy(), g["%s_e%d_b%d_te" % (tag, e, b)])
                else:
                    assert dte is None
    # weighted (non-binary) values survive the upload
    w = csr_matrix(np.array([[0, 2.5, 0, 1.0], [3.0, 0, 0, 0]]))
    (d, _), = list(DataSampler(w, batch_size=2, shuffle=False))
    assert np.array_equal(d.cpu().numpy(), w.toarray().astype(np.float32))


# ---------------------------------------------------------------------------------------------- end-to-end trajectory
# bounds = a few times the deviations achieved on MI355X (round 3: fp32 loss 1.6e-7 / metrics 0; bf16 loss 1.7e-5 / metrics 9.2e-4;
# round 2 asserted 2e-2 / 3e-2 for bf16)
@pytest.mark.parametrize("numerics,tol_loss,tol_metric", [("fp32", 2e-6, 2e-4), ("bf16", 6e-5, 2e-3)])
def test_g8_epoch_curve(numerics, tol_loss, tol_metric):
    """5 epochs (8 batches each) on 512x128 synthetic data with the reference's RNG draws injected: per-batch
    loss trajectory, and nDCG@100 / Recall@50 per epoch through evaluate()."""
    from rectorch_amd.utils import hash_state_dict
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.evaluation import evaluate
    g = load_golden("g8_epoch_curve")
    I, H, L = [int(v) for v in g["dims"]]
    beta, anneal, p, lr, B = [float(v) for v in g["meta"]]
    B = int(B)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", int(g["hash_seed"]), bias_std=0.1)
    net, model = make_vae([I, H, L], [L, H, I], p, sd, beta=beta, anneal_steps=int(anneal), learning_rate=lr,
                          numerics=numerics)
    train = csr_matrix(g["dense_train"].astype(np.float64))
    val_tr, val_te = csr_matrix(g["val_tr"].astype(np.float64)), csr_matrix(g["val_te"].astype(np.float64))
    masks = np.unpackbits(g["mask_bits"], axis=-1)[..., :I]
    n_epochs = g["losses"].shape[0]
    worst_loss, worst_metric = 0.0, 0.0
    for e in range(n_epochs):
        np.random.seed(8000 + e)
        smp = DataSampler(train, batch_size=B, shuffle=True)
        net.train()
        for b, rb in enumerate(smp.iter_rows()):
            assert np.array_equal(rb.rows.cpu().numpy(), g["perms"][e][b * B:(b + 1) * B])
            model._rtx.inject = (dev(masks[e, b], torch.uint8), dev(g["eps"][e, b]))
            loss = model._fused_step(rb, None, want_loss=True)
            worst_loss = max(worst_loss, abs(loss - g["losses"][e, b]) / abs(g["losses"][e, b]))
            assert abs(loss - g["losses"][e, b]) < tol_loss * abs(g["losses"][e, b]), (e, b, loss, g["losses"][e, b])
        res = evaluate(model, DataSampler(val_tr, val_te, batch_size=32, shuffle=False), ["ndcg@100", "recall@50"])
        worst_metric = max(worst_metric, abs(np.mean(res["ndcg@100"]) - g["ndcg100"][e]), abs(np.mean(res["recall@50"]) - g["recall50"][e]))
        assert abs(np.mean(res["ndcg@100"]) - g["ndcg100"][e]) < tol_metric
        assert abs(np.mean(res["recall@50"]) - g["recall50"][e]) < tol_metric
    print("g8 %s: worst loss deviation %.2e relative, worst |nDCG@100 / Recall@50 - reference| %.2e (bounds %.0e / %.0e)"
          % (numerics, worst_loss, worst_metric, tol_loss, tol_metric))
    if numerics == "fp32":
        assert np.max(np.abs(res["ndcg@100"] - g["ndcg100_users"])) < 2e-2    # per-user, rank flips on near-ties only
        sdf = sd_from(g, "sd_final__")
        for k, v in net.state_dict().items():
            assert float(np.max(np.abs(v.cpu().numpy() - sdf[k]))) < 2e-4, k


# ---------------------------------------------------------------------------------------------- the reference's own model tests
def test_reference_test_multivae_semantics():
    """restates reference tests/test_models.py:213-283 (test_MultiVAE) on the HIP path"""
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.models import MultiVAE
    from rectorch_amd.samplers import DataSampler
    net = MultiVAE_net([1, 2], [2, 1], .1)
    model = MultiVAE(net)
    assert model.device == torch.device("cuda") and isinstance(model.optimizer, torch.optim.Adam)
    gt = torch.FloatTensor([[1, 1], [2, 1]])
    pred = torch.sigmoid(torch.FloatTensor([[1, 1], [1, 1]]))
    torch.manual_seed(12345)
    mu, logvar = model.network.encode(gt)
    assert model.loss_function(pred, gt, mu, logvar).item() != 0.0
    train = csr_matrix((np.ones(3), (np.array([0, 0, 1]), np.array([0, 1, 1]))))
    sampler = DataSampler(train, batch_size=1, shuffle=False)
    x = torch.FloatTensor([[1, 1], [2, 2]])
    model.predict(x, True)
    out_1 = model.predict(x, False)[0]
    model.train(sampler, num_epochs=10, verbose=4)
    out_2 = model.predict(x, False)[0]
    assert not torch.all(out_1.eq(out_2)), "the outputs should be different after training"
    tmp = tempfile.NamedTemporaryFile()
    model.save_model(tmp.name, 1)
    model2 = MultiVAE(MultiVAE_net([1, 2], [2, 1], .1))
    model2.load_model(tmp.name)
    assert torch.all(model.predict(x, False)[0].eq(model2.predict(x, False)[0])), "the outputs should be the same"
    sampler = DataSampler(train, train, batch_size=1, shuffle=False)
    tmp2 = tempfile.NamedTemporaryFile()
    model = MultiVAE(MultiVAE_net([1, 2], [2, 1], .1), 1., 5)
    model.train(sampler, valid_data=sampler, valid_metric="ndcg@1", num_epochs=10, best_path=tmp2.name)
    model2 = MultiVAE(MultiVAE_net([1, 2], [2, 1], .1), 1., 5)
    assert model2.gradient_updates == 0
    model2.load_model(tmp2.name)
    assert model2.gradient_updates > 0
    with pytest.raises(AssertionError):
        model.train(sampler, valid_data=sampler, num_epochs=1)          # valid_metric is required


def test_reference_test_multidae_and_nets_semantics():
    """restates reference tests/test_models.py:159-211 (test_MultiDAE) and tests/test_nets.py:27-75"""
    from rectorch_amd.nets import MultiDAE_net, MultiVAE_net
    from rectorch_amd.models import MultiDAE
    from rectorch_amd.samplers import DataSampler
    net = MultiDAE_net([1, 2], [2, 1], dropout=.1)
    model = MultiDAE(net)
    x = torch.FloatTensor([[1, 1], [2, 2]])
    y = net(x)
    assert y.shape == x.shape and y.dtype == torch.float32
    train = csr_matrix((np.ones(3), (np.array([0, 0, 1]), np.array([0, 1, 1]))))
    sampler = DataSampler(train, batch_size=1, shuffle=False)
    out_1 = model.predict(x, False)[0]
    model.train(sampler, num_epochs=10, verbose=4)
    out_2 = model.predict(x, False)[0]
    assert not torch.all(out_1.eq(out_2))
    tmp = tempfile.NamedTemporaryFile()
    model.save_model(tmp.name, 1)
    model2 = MultiDAE(MultiDAE_net([1, 2], [2, 1], dropout=.1))
    model2.load_model(tmp.name)
    assert torch.all(model.predict(x, False)[0].eq(model2.predict(x, False)[0]))
    # MultiVAE_net: same seed -> encode() and forward() agree on mu/logvar in training mode (RNG order)
    vnet = MultiVAE_net([1, 2], [2, 1], .1).cuda()
    vnet.train()
    torch.manual_seed(98765)
    mu, logvar = vnet.encode(x)
    torch.manual_seed(98765)
    y, mu2, logvar2 = vnet(x)
    assert mu.equal(mu2) and logvar.equal(logvar2) and y.shape == x.shape


# ---------------------------------------------------------------------------------------------- oracle parity at scale
@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_ndcg_recall_parity_vs_oracle_ml20m_items(numerics):
    """nDCG@100 / Recall@50 of predict() on 192 held-out users at I = 20108 vs the CPU oracle, same weights"""
    from oracle import c_oracle
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.utils.synth import split_heldout
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.evaluation import evaluate
    from rectorch_amd.metrics import Metrics
    I, H, L = 20108, 600, 200
    X = synth_interactions(192, I, seed=99)
    val_tr, val_te = split_heldout(X, 0.2, seed=3)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 777, bias_std=0.05)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, predict_numerics=numerics)
    res = evaluate(model, DataSampler(val_tr, val_te, batch_size=64, shuffle=False), ["ndcg@100", "recall@50"])
    params, _ = params_in_order(sd)
    xo = np.asarray(val_tr.toarray(), dtype=np.float32)
    lo = c_oracle.predict([I, H, L], [L, H, I], params, xo, True)[0]
    ro = Metrics.compute(lo, np.asarray(val_te.toarray(), dtype=np.float32), ["ndcg@100", "recall@50"])
    d_ndcg = abs(np.mean(res["ndcg@100"]) - np.mean(ro["ndcg@100"]))
    d_rec = abs(np.mean(res["recall@50"]) - np.mean(ro["recall@50"]))
    print("%s: nDCG@100 hip %.5f oracle %.5f | Recall@50 hip %.5f oracle %.5f" % (
        numerics, np.mean(res["ndcg@100"]), np.mean(ro["ndcg@100"]), np.mean(res["recall@50"]), np.mean(ro["recall@50"])))
    # RELATIVE bounds (the means are ~2.5e-3 at these random weights: chance level, where bf16 rounding reorders near-ties)
    tol = 1e-3 if numerics == "fp32" else 5e-2
    assert d_ndcg < tol * np.mean(ro["ndcg@100"]) and d_rec < tol * np.mean(ro["recall@50"])
    if numerics == "fp32":
        assert np.max(np.abs(res["ndcg@100"] - ro["ndcg@100"])) < 1e-3
    else:
        # what bf16 must preserve is the ranking itself: overlap of the two top-100 lists per user
        hip = np.concatenate([model.predict(torch.from_numpy(xo[i:i + 64]).cuda())[0].cpu().numpy() for i in range(0, 192, 64)])
        top_h = np.argpartition(-hip, 100, axis=1)[:, :100]
        top_o = np.argpartition(-lo, 100, axis=1)[:, :100]
        overlap = np.array([len(set(a) & set(b)) for a, b in zip(top_h, top_o)]) / 100.0
        print("bf16 top-100 overlap with the oracle: mean %.3f min %.2f" % (overlap.mean(), overlap.min()))
        assert overlap.mean() > 0.85 and overlap.min() > 0.60


# ---------------------------------------------------------------------------------------------- properties at the full shape
def test_full_size_properties_b500():
    """ml-20m shape, B = 500, bf16: properties that need no reference at this size."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.engine import RowBatch
    I, H, L, B = 20108, 600, 200, 500
    X = synth_interactions(2000, I, seed=5)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 31, bias_std=0.05)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=1000, numerics="bf16")
    smp = DataSampler(X, batch_size=B, shuffle=False)
    rbs = list(smp.iter_rows())
    assert len(rbs) == 4 and len(rbs[0]) == B
    # (1) predict: exactly nnz(x) scores are -inf, everything else finite
    pred = model.predict(smp._csr_tr.gather_dense(rbs[0].rows))[0]
    assert int(torch.isinf(pred).sum().item()) == int(X[:B].nnz) and not torch.isnan(pred).any()
    # (2) data-parallel identity: gradients of the full batch == sum of the two half-batches' gradients at the
    #     same 1/B scale (what the RCCL all-reduce computes), with the same dropout mask / eps
    st, params, m, v = model._ensure_train_state()
    eng = net.rtx_engine("bf16", B, train_buffers=(st.grads, m, v))
    gen = torch.Generator().manual_seed(1)
    mask = (torch.rand(B, I, generator=gen) >= 0.5).to(torch.uint8).cuda()
    eps = torch.randn(B, L, generator=gen).cuda()
    kw = dict(beta=0.1, lam=0.0, inv_batch=1.0 / B, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1)
    loss = torch.zeros(2, device="cuda")
    eng.loss_grads(rbs[0], None, eng._step(mask=mask, noise=eps, **kw), loss[0:1])
    full = st.flat_grads.clone()
    full_loss = loss[0].item()
    acc = torch.zeros_like(full)
    half_loss = 0.0
    for lo, hi in ((0, 250), (250, 500)):
        rb = RowBatch(rbs[0].tr, None, rbs[0].rows[lo:hi].contiguous())
        eng.loss_grads(rb, None, eng._step(mask=mask[lo:hi].contiguous(), noise=eps[lo:hi].contiguous(), **kw), loss[0:1])
        acc += st.flat_grads
        half_loss += loss[0].item()
    assert abs(half_loss - full_loss) < 1e-4 * abs(full_loss)
    assert float((acc - full).abs().max() / full.abs().max()) < 2e-3
    # (3) sum_i d loss / d b_out[i] = sum_b (s_b * sum_i softmax_bi - s_b) / B = 0 (softmax rows sum to one)
    gb_out = net._param_list()[-1].grad
    assert abs(float(gb_out.sum())) < 2e-2 * float(gb_out.abs().sum())
    # (4) training drives the loss down and keeps the replicas' invariants: finite params, Adam step count
    torch.manual_seed(0)
    losses = [model._fused_step(rbs[i % 4], None, want_loss=True) for i in range(24)]
    assert np.isfinite(losses).all() and np.mean(losses[-4:]) < np.mean(losses[:4])
    assert model._rtx.adam_step == 24 and model.gradient_updates == 24.0
    assert all(torch.isfinite(p).all() for p in net.parameters())
    # (5) determinism: same torch seed -> bit-identical loss sequence from the Philox path
    net2, model2 = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=1000, numerics="bf16")
    torch.manual_seed(0)
    l2 = [model2._fused_step(rbs[i % 4], None, want_loss=True) for i in range(3)]
    assert l2 == losses[:3]


def test_edge_cases():
    """empty rows, single-item rows, batch of 1, ragged last batch, dense vs resident input agree"""
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.utils import hash_state_dict
    I, H, L = 300, 40, 10
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5, bias_std=0.2)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, numerics="fp32")
    dense = np.zeros((7, I), dtype=np.float64)
    dense[0, [3, 7, 299]] = 1
    dense[2, 5] = 1                      # rows 1, 3 are empty
    dense[4, :] = 1                      # a user who has every item
    dense[5, 10:20] = 2.0
    dense[6, 0] = 1
    X = csr_matrix(dense)
    smp = DataSampler(X, batch_size=3, shuffle=False)
    batches = list(smp)
    assert [b[0].shape[0] for b in batches] == [3, 3, 1]
    for (d, _), lo in zip(batches, (0, 3, 6)):
        p_res = model.predict(d)[0]                                   # resident rows (tensor carries its row ids)
        p_dense = model.predict(torch.from_numpy(dense[lo:lo + d.shape[0]].astype(np.float32)))[0]   # dense drop-in
        assert torch.equal(torch.isinf(p_res), torch.isinf(p_dense))
        fin = torch.isfinite(p_res)
        assert torch.allclose(p_res[fin], p_dense[fin], rtol=0, atol=1e-6)
        assert not torch.isnan(p_res).any()
    assert bool(torch.isinf(model.predict(batches[1][0])[0][1]).all())           # all items removed for user 4
    torch.manual_seed(3)
    l1 = model.train_batch(batches[2][0])                                         # batch of one
    assert np.isfinite(l1)
    empty_only = torch.zeros(2, I)
    assert np.isfinite(model.train_batch(empty_only))                              # all-zero rows: loss is just KL
    with pytest.raises(Exception):
        model.predict(torch.zeros(2, I + 1))                                      # wrong width fails loudly


def test_custom_ops_match_the_direct_calls():
    """torch.ops.rectorch_hip.* (rectorch_amd/ops.py) are thin adapters over the same C entry points"""
    from rectorch_amd import ops  # noqa: F401
    from rectorch_amd.engine import CsrMatrix
    from rectorch_amd.utils import hash_state_dict
    I, H, L = 200, 32, 8
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 9, bias_std=0.2)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, numerics="fp32")
    rng = np.random.default_rng(1)
    dense = (rng.random((12, I)) < 0.1).astype(np.float32)
    x = dev(dense)
    eng = net.rtx_engine("fp32", 12)
    a = eng.forward(x, remove_train=True)
    b = torch.ops.rectorch_hip.mvae_forward(eng.op_handle, x, False, True, 0)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    assert torch.equal(model.predict(x)[0], a[0])                  # predict() goes through the op
    loss_op = torch.ops.rectorch_hip.multinomial_loss(a[0].nan_to_num(neginf=0.0), x, a[1], a[2], 0.3)
    loss_fn = model.loss_function(a[0].nan_to_num(neginf=0.0), x, a[1], a[2], 0.3)
    assert torch.equal(loss_op, loss_fn)
    csr = CsrMatrix(csr_matrix(dense.astype(np.float64)))
    rows = torch.tensor([3, 0, 11], dtype=torch.int32, device="cuda")
    assert np.array_equal(torch.ops.rectorch_hip.csr_gather_dense(csr.op_handle, rows).cpu().numpy(), dense[[3, 0, 11]])


def test_config0_multidae_ml100k_shape_vs_oracle():
    """BASELINE.json configs[0]: MultiDAE on ml-100k-shaped data (740 x 1450, ~60 nnz/user), dims [I,600,200],
    lam 0.2, lr 1e-3, batch 250 (config/config_dae.json) and 100: 6 steps with injected dropout masks vs the CPU oracle
    (loss curve + parameters), then nDCG@100 of predict() on held-out users vs the oracle's predict."""
    from oracle import c_oracle
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.metrics import Metrics
    U, I, H, L = 740, 1450, 600, 200
    X = synth_interactions(U, I, mu=3.8, sigma=0.7, dmin=3, dmax=400, seed=100)
    dense = np.asarray(X.toarray(), dtype=np.float32)
    for B in (250, 100):
        sd = hash_state_dict([I, H, L], [L, H, I], "dae", 55, bias_std=0.1)
        net, model = make_dae([I, H, L], [L, H, I], 0.5, sd, lam=0.2, numerics="fp32")
        params, keys = params_in_order(sd)
        ref = c_oracle.OracleTrainer([I, H, L], [L, H, I], params, "dae", 0.5, lam=0.2, lr=1e-3)
        rng = np.random.default_rng(B)
        for t in range(6):
            rows = rng.choice(U, B, replace=False)
            mask = (rng.random((B, I)) >= 0.5).astype(np.uint8)
            model._rtx.inject = (dev(mask, torch.uint8), None)
            loss = model.train_batch(torch.from_numpy(dense[rows]))
            ref_loss = ref.train_batch(dense[rows], None, mask, None)
            assert abs(loss - ref_loss) < 2e-5 * abs(ref_loss), (B, t, loss, ref_loss)
        for p, r, k in zip(net._param_list(), ref.params, keys):
            # 6 Adam steps.  The update is lr*g/(|g|+eps): an entry whose gradient is ~1e-8 moves by O(lr) on a 1e-9
            # gradient difference, so a handful of entries may differ by a fraction of one lr-sized step while all the
            # others agree to float rounding
            d = np.abs(p.detach().cpu().numpy() - r)
            assert float(d.max()) < 1e-3 and float(np.mean(d > 2e-5)) < 1e-4, (B, k, float(d.max()), float(np.mean(d > 2e-5)))
        held = dense[:100].copy()
        te = np.zeros_like(held)
        for u in range(100):                      # hold out every 5th item of each user
            nz = np.flatnonzero(held[u])[::5]
            te[u, nz] = 1
            held[u, nz] = 0
        pred = model.predict(torch.from_numpy(held))[0].cpu().numpy()
        (pref,) = ref.predict(held, True)
        a = Metrics.ndcg_at_k(pred, te, 100)
        b = Metrics.ndcg_at_k(pref, te, 100)
        assert abs(np.nanmean(a) - np.nanmean(b)) < 1e-4


def test_config3_netflix_shape_step_is_finite_and_consistent():
    """BASELINE.json configs[3] shape on one GPU: I = 17769, 512 users per GPU, bf16; the fp32 and bf16 engines see
    the same batch and must agree on the loss to bf16 accuracy; gradients all finite."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 17769, 600, 200, 512
    X = synth_interactions(1024, I, mu=4.3, sigma=1.0, dmax=5000, seed=17)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 3, bias_std=0.05)
    losses = {}
    for numerics in ("fp32", "bf16"):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, numerics=numerics)
        (rb, _) = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
        gen = torch.Generator().manual_seed(5)
        mask = (torch.rand(B, I, generator=gen) >= 0.5).to(torch.uint8).cuda()
        eps = torch.randn(B, L, generator=gen).cuda()
        model._rtx.inject = (mask, eps)
        losses[numerics] = model._fused_step(rb, None, want_loss=True)
        assert all(torch.isfinite(p.grad).all() for p in net._param_list())
    assert abs(losses["bf16"] - losses["fp32"]) < 3e-3 * abs(losses["fp32"]), losses


def test_evaluate_device_equals_host_evaluate():
    """nDCG@k / Recall@k from the device top-k kernel equal Metrics.* on the same scores (float64),
    including users with a single held-out item, more held-out items than k, and an empty held-out row (nan)."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.utils.synth import split_heldout
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.evaluation import evaluate, evaluate_host, evaluate_device, ValidFunc
    from rectorch_amd.engine import CsrMatrix, topk_metrics
    from rectorch_amd.metrics import Metrics
    # (a) raw kernel vs Metrics on golden G6 scores (contains -inf and ties-free random scores)
    g = load_golden("g6_metrics")
    scores, held = g["scores"], g["heldout"]
    hm = CsrMatrix(csr_matrix(held.astype(np.float64)))
    rows = torch.arange(scores.shape[0], dtype=torch.int32, device="cuda")
    ks = [10, 100, 20, 50, 300]
    ndcg, recall, topk = topk_metrics(dev(scores), hm, rows, ks, want_topk=True)
    for q, k in enumerate(ks):
        ref_n = Metrics.ndcg_at_k(scores, held, k)
        ref_r = Metrics.recall_at_k(scores, held, k)
        assert np.allclose(ndcg[q].cpu().numpy(), ref_n, rtol=1e-12, atol=0, equal_nan=True), k
        assert np.allclose(recall[q].cpu().numpy(), ref_r, rtol=1e-12, atol=0, equal_nan=True), k
    assert np.allclose(ndcg[1].cpu().numpy(), g["res__ndcg_at_100"], rtol=1e-12, equal_nan=True)       # the reference's own output
    assert np.allclose(recall[3].cpu().numpy(), g["res__recall_at_50"], rtol=1e-12, equal_nan=True)
    order = np.argsort(-scores, axis=1, kind="stable")[:, :50]
    fin = np.take_along_axis(scores, order, 1) > -np.inf
    assert np.array_equal(topk.cpu().numpy()[:, :50][fin], order[fin])          # sorted ids, wherever scores are finite
    # (b) end to end at I = 20108: evaluate_device == evaluate through the same model
    I, H, L = 20108, 600, 200
    X = synth_interactions(300, I, seed=123)
    val_tr, val_te = split_heldout(X, 0.2, seed=4)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 777, bias_std=0.05)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd)
    mets = ["ndcg@100", "ndcg@10", "recall@50", "recall@20"]
    smp = DataSampler(val_tr, val_te, batch_size=128, shuffle=False)
    dev_res = evaluate_device(model, smp, mets)
    host_res = evaluate_host(model, smp, mets)       # the reference's loop: scores and held-out rows to the host, numpy metrics
    auto_res = evaluate(model, smp, mets)            # round 5: evaluate() itself takes the device route on a resident sampler
    for m in mets:
        assert np.array_equal(auto_res[m], dev_res[m])
    model.device_metrics = False
    off_res = evaluate(model, smp, mets)
    model.device_metrics = True
    for m in mets:
        assert np.array_equal(off_res[m], host_res[m])
    for m in mets:
        assert dev_res[m].shape == host_res[m].shape == (300,)
        assert np.allclose(dev_res[m], host_res[m], rtol=1e-12, atol=0, equal_nan=True), m
    vf = ValidFunc(evaluate_device)
    assert np.allclose(vf(model, smp, "ndcg@100"), host_res["ndcg@100"], equal_nan=True)
    # unsupported metric -> falls back to the host path and still answers
    mixed = evaluate_device(model, smp, ["mrr@10", "ndcg@10"])
    assert set(mixed) == {"mrr@10", "ndcg@10"} and np.allclose(mixed["ndcg@10"], host_res["ndcg@10"], equal_nan=True)


def test_topk_kernel_order_statistics_stress():
    """Round 6: the selection kernel ranks by counting instead of sorting.  Against numpy's stable order (score descending, item id
    ascending among equal scores -- the kernel's documented tie rule) on the cases the counting has to get right: heavy ties
    (scores quantised to a few levels), rows that are entirely -inf, more than 1024 elements tied above the bound (the radix
    fall-back), K from 1 to 1000 (1 .. 4 maxima per thread), widths that are not a multiple of 4, rows shorter than K, and
    held-out rows longer than the 512 entries the kernel parks in LDS.  Metrics against rectorch's own formulas (metrics.py:136-147,
    187-196) evaluated with that same order."""
    from rectorch_amd.engine import CsrMatrix, topk_metrics
    rng = np.random.RandomState(11)

    def ref(scores, held, ks):
        B, I = scores.shape
        order = np.lexsort((np.broadcast_to(np.arange(I), (B, I)), -scores.astype(np.float64)), axis=1)   # primary: score desc, then id asc
        nd, rc = np.empty((len(ks), B)), np.empty((len(ks), B))
        for q, k in enumerate(ks):
            kk = min(k, I)
            for b in range(B):
                relv = held[b, order[b, :kk]]
                disc = 1.0 / np.log2(np.arange(2, kk + 2))
                dcg = float((relv * disc).sum())
                n = int(held[b].sum())
                idcg = float(disc[:min(n, kk)].sum())
                with np.errstate(divide="ignore", invalid="ignore"):
                    nd[q, b] = np.float64(dcg) / np.float64(idcg)
                    rc[q, b] = np.float64(np.float32((relv > 0).sum())) / np.float64(min(kk, int((held[b] > 0).sum())))
        return order, nd, rc

    cases = []
    for (B, I, levels, ks, held_n) in [(9, 20108, 7, [100, 50], 30), (5, 20108, 0, [1000, 300, 1], 700), (6, 4099, 3, [257, 10], 20),
                                        (4, 90, 0, [100, 5], 10), (7, 20108, 2000, [512, 100], 5), (3, 1030, 1, [10], 3)]:
        sc = rng.randn(B, I).astype(np.float32)
        if levels:
            sc = np.round(sc * levels / 3.0).astype(np.float32) * (3.0 / levels)     # `levels` distinct values per unit: ties everywhere
        sc[0, :] = -np.inf                                                           # a fully masked row
        if B > 2:
            sc[1, rng.rand(I) < 0.3] = -np.inf
            sc[2, :] = 0.25                                                          # every element tied: > 1024 candidates at the bound
        held = np.zeros((B, I), np.float64)
        for b in range(B):
            n = held_n if b != B - 1 else 0                                          # the last user has an empty held-out row (nan metrics)
            held[b, rng.choice(I, size=min(n, I), replace=False)] = 1.0
        cases.append((sc, held, ks))
    n_unamb = 0
    for sc, held, ks in cases:
        B, I = sc.shape
        hm = CsrMatrix(csr_matrix(held))
        rows = torch.arange(B, dtype=torch.int32, device="cuda")
        ndcg, recall, topk = topk_metrics(dev(sc), hm, rows, ks, want_topk=True)
        order, nd, rc = ref(sc, held, ks)
        kmax = min(max(ks), I)
        got = topk.cpu().numpy()
        want = order[:, :kmax]
        fin = np.take_along_axis(sc, want, 1) > -np.inf
        # Which of the elements TIED AT THE K-TH PLACE make the list is arbitrary when the radix fall-back runs (as in the reference's
        # argpartition): ids and metrics are asserted on the rows without such a tie, the scores of the ranked items on every row.
        unamb = np.zeros(B, bool)
        for b in range(B):
            srt = np.sort(sc[b])[::-1]
            unamb[b] = srt[kmax - 1] > -np.inf and (kmax >= I or (sc[b] >= srt[kmax - 1]).sum() == kmax)
            assert np.array_equal(sc[b][got[b][fin[b]]], sc[b][want[b][fin[b]]]), (I, ks, b)
            if unamb[b]:
                assert np.array_equal(got[b][fin[b]], want[b][fin[b]]), (I, ks, b)
        n_unamb += int(unamb.sum())
        for q in range(len(ks)):
            assert np.allclose(ndcg[q].cpu().numpy()[unamb], nd[q][unamb], rtol=1e-12, atol=0, equal_nan=True), (I, ks, q)
            assert np.allclose(recall[q].cpu().numpy()[unamb], rc[q][unamb], rtol=1e-12, atol=0, equal_nan=True), (I, ks, q)
    assert n_unamb >= 6      # (the cases above hold enough rows whose ranked list is unique)


# ------------------------------------------------------------------------------------------------ EASE
# Tolerance: the reference computes in float64 with LAPACK's LU inverse, the device in float64 with a Cholesky
# inverse; both are backward stable on the SPD matrix X^T X + lam I (condition number <= (|X|_2^2 + lam) / lam), so
# the score matrices agree to ~1e-12 absolute on these fixtures.  Asserted: 1e-10.
def test_ease_binary_g10():
    from rectorch_amd.models import EASE
    g = load_golden("g10_ease_binary")
    X = g["X"].astype(np.float64)
    ease = EASE(float(g["lam"]))
    assert ease.model is None and str(ease) == str(g["str_new"])
    ease.train(csr_matrix(X))
    assert str(ease) == str(g["str_trained"]) and repr(ease) == str(ease)
    pr = ease.predict(g["ids"], csr_matrix(g["te"].astype(np.float64)))[0]
    assert pr.shape == g["pred_remove"].shape and pr.dtype == np.float64
    assert np.array_equal(np.isneginf(pr), np.isneginf(g["pred_remove"]))
    fin = np.isfinite(pr)
    np.testing.assert_allclose(pr[fin], g["pred_remove"][fin], rtol=0, atol=1e-10)
    pk = ease.predict(g["ids"], csr_matrix(g["te"].astype(np.float64)), remove_train=False)[0]
    np.testing.assert_allclose(pk, g["pred_keep"], rtol=0, atol=1e-10)
    assert isinstance(ease.model, np.ndarray)
    np.testing.assert_allclose(ease.model, g["model"], rtol=0, atol=1e-10)
    B = ease._solver.weights().cpu().numpy()
    assert np.all(np.diag(B) == 0)


def test_ease_ratings_f64_gram_g10():
    from rectorch_amd.models import EASE
    g = load_golden("g10_ease_ratings")
    ease = EASE(float(g["lam"]))
    ease.train(csr_matrix(g["X"]))
    np.testing.assert_allclose(ease.model, g["model"], rtol=0, atol=1e-10)


def test_ease_reference_api_and_model_file():
    """the reference's own test_EASE (tests/test_models.py:359-381) + loading a model file the reference wrote"""
    from rectorch_amd.models import EASE
    ease = EASE(200.)
    assert hasattr(ease, "lam") and hasattr(ease, "model")
    assert ease.lam == 200 and ease.model is None and repr(ease) == str(ease)
    X = csr_matrix(np.random.RandomState(0).randint(2, size=(10, 5)), dtype="float64")
    ease.train(X)
    assert isinstance(ease.model, np.ndarray)
    pr = ease.predict([2, 4, 5], X[[2, 4, 5]])[0]
    assert pr.shape == (3, 5)
    tmp = tempfile.NamedTemporaryFile()
    ease.save_model(tmp.name)
    ease2 = EASE(200.)
    ease2.load_model(tmp.name + ".npy")
    assert np.all(ease2.model == ease.model)
    os.remove(tmp.name + ".npy")
    g = load_golden("g10_ease_binary")
    ease3 = EASE()
    state = ease3.load_model(os.path.join(ROOT, "tests", "golden", "g10_reference_ease_model.npy"))
    assert ease3.lam == 200. and set(state.keys()) == {"lambda", "model"}
    pr = ease3.predict(g["ids"], csr_matrix(g["te"].astype(np.float64)))[0]
    assert np.array_equal(pr, g["pred_remove"])


@pytest.mark.parametrize("U,I,lam,density", [(3000, 1000, 100., 0.03), (700, 1500, 10., 0.01), (129, 129, 1., 0.3)])
def test_ease_vs_oracle_mid_size(U, I, lam, density):
    from oracle.ease_oracle import ease_fit
    from rectorch_amd.engine import EaseSolver
    rng = np.random.RandomState(U + I)
    X = (rng.rand(U, I) < density).astype(np.float64)
    s = EaseSolver(csr_matrix(X), lam)
    B = s.weights().cpu().numpy()
    Bo = ease_fit(X, lam)
    assert np.all(np.diag(B) == 0)
    assert np.max(np.abs(B - Bo)) <= 1e-10 * max(1.0, np.max(np.abs(Bo)))
    ids = rng.randint(0, U, size=50)
    sc = s.scores(ids).cpu().numpy()
    np.testing.assert_allclose(sc, X[ids] @ Bo, rtol=0, atol=1e-9)
    # size-independent property: (G + lam I) P = I  <=>  G B_j + lam B_j = -e_j / P_jj off the constraint; check the
    # KKT condition of the EASE problem instead: (G + lam I) B has constant columns off the diagonal = 0, i.e.
    # R = (G + lam I) B + diag(1/P_jj) - G  vanishes ... restated: (G + lam I)(I - B) is diagonal.
    G = X.T @ X + lam * np.eye(I)
    R = G @ (np.eye(I) - B)
    off = R - np.diag(np.diag(R))
    assert np.max(np.abs(off)) <= 1e-8 * np.max(np.abs(np.diag(R)))


def test_ease_not_positive_definite_is_an_error():
    from rectorch_amd._lib import RtxError
    from rectorch_amd.engine import EaseSolver
    X = np.zeros((4, 6))
    X[0, 0] = X[1, 1] = 1.0
    with pytest.raises(RtxError, match="positive definite"):
        EaseSolver(csr_matrix(X), 0.0)


@pytest.mark.parametrize("vmax,U,I", [(5, 900, 300), (200, 300, 260), (16, 400, 200)])
def test_ease_integer_valued_gram_paths(vmax, U, I):
    """integer ratings: the Gram matrix runs on fp8 MFMA (|v| <= 16) or bf16 MFMA (|v| <= 256) and must still be exact"""
    from oracle.ease_oracle import ease_fit
    from rectorch_amd.engine import EaseSolver
    rng = np.random.RandomState(vmax)
    X = ((rng.rand(U, I) < 0.1) * rng.randint(1, vmax + 1, size=(U, I))).astype(np.float64)
    X[0, 0] = vmax
    B = EaseSolver(csr_matrix(X), 30.0).weights().cpu().numpy()
    Bo = ease_fit(X, 30.0)
    assert np.max(np.abs(B - Bo)) <= 1e-10 * max(1.0, np.max(np.abs(Bo)))


# ------------------------------------------------------------------------------------------------ CMultiVAE
def make_cvae(cond, enc, dec, p, sd, **kw):
    from rectorch_amd.nets import CMultiVAE_net
    from rectorch_amd.models import CMultiVAE
    net = CMultiVAE_net(cond, list(dec), list(enc), dropout=p)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
    return net, CMultiVAE(net, **kw)


def test_g11_cmvae_eval_forward_and_predict():
    g = load_golden("g11_cmvae_fwd_eval")
    I, H, L = [int(v) for v in g["dims"]]
    C_ = int(g["cond_dim"])
    net, model = make_cvae(C_, [I, H, L], [L, H, I], 0.5, sd_from(g, "sd__"), beta=0.3)
    assert net.enc_layers[0].weight.shape == (H, I + C_)
    net.eval()
    y, mu, logvar = net(torch.from_numpy(g["x"]))
    assert y.shape == (g["x"].shape[0], I)
    assert rel(y.cpu(), g["logits"]) < 1e-5 and rel(mu.cpu(), g["mu"]) < 1e-5 and rel(logvar.cpu(), g["logvar"]) < 1e-5
    pred = model.predict(torch.from_numpy(g["x"]), remove_train=True)[0].cpu().numpy()
    assert np.array_equal(np.isneginf(pred), np.isneginf(g["pred"]))
    fin = np.isfinite(pred)
    assert rel(pred[fin], g["pred"][fin]) < 1e-5
    with pytest.raises(Exception):
        model.train_batch(torch.from_numpy(g["x"]))          # a conditioned row cannot be its own target


def test_g11_cmvae_train_steps_fp32():
    g = load_golden("g11_cmvae_train_steps")
    enc, dec = [int(v) for v in g["enc_dims"]], [int(v) for v in g["dec_dims"]]
    beta, anneal, p, lr = [float(v) for v in g["meta"]]
    C_ = g["xs"].shape[2] - enc[0]
    net, model = make_cvae(C_, enc, dec, p, sd_from(g, "sd0__"), beta=beta, anneal_steps=int(anneal), learning_rate=lr,
                           numerics="fp32")
    _, keys = params_in_order(sd_from(g, "sd0__"))
    for t in range(g["xs"].shape[0]):
        model._rtx.inject = (dev(g["mask_%d" % t], torch.uint8), dev(g["eps_%d" % t]))
        loss = model.train_batch(torch.from_numpy(g["xs"][t]), torch.from_numpy(g["gts"][t]))
        assert abs(loss - float(g["loss_%d" % t])) < 1e-5 * abs(float(g["loss_%d" % t])), (t, loss)
        sd_t, _ = params_in_order(sd_from(g, "sd_%d__" % t))
        for k, prm, ref in zip(keys, net._param_list(), sd_t):
            gref = g["grad_%d__%s" % (t, k.replace(".", "__"))]
            assert rel(prm.grad.cpu(), gref) < 2e-4, (t, k)
            assert float(np.max(np.abs(prm.detach().cpu().numpy() - ref))) < 5e-6, (t, k)
    assert model.gradient_updates == float(g["gradient_updates"])


def test_cmvae_sparse_sampler_batches_equal_dense_batches():
    """the conditioned sampler's per-batch CSR form (sparse=True) drives the engine to the same step as its dense form"""
    from rectorch_amd.samplers import ConditionedDataSampler
    from rectorch_amd.utils.hashinit import hash_state_dict
    rng = np.random.RandomState(4)
    U, I, C_, H, L = 40, 96, 3, 24, 8
    tr = csr_matrix((rng.rand(U, I) < 0.15).astype(np.float32))
    tr = csr_matrix(tr + csr_matrix((np.ones(U), (np.arange(U), np.arange(U) % I)), shape=(U, I)))
    tr.data[:] = 1.0
    iid2cids = {i: sorted({int(i % C_), int((i * 7) % C_)}) for i in range(I)}
    sd = hash_state_dict([I + C_, H, L], [L, H, I], "vae", 5, 1.0)
    losses = {}
    for sparse in (False, True):
        torch.manual_seed(123)                 # the per-step Philox seed is drawn from torch's CPU generator
        net, model = make_cvae(C_, [I, H, L], [L, H, I], 0.0, sd, beta=0.1, numerics="fp32")
        sampler = ConditionedDataSampler(iid2cids, C_, tr, None, batch_size=16, shuffle=False, sparse=sparse)
        out = []
        for item in sampler:
            data, gt = item                    # every sampler yields a PAIR, the sparse form included
            out.append(model.train_batch(data, gt))
        losses[sparse] = out
    assert len(losses[True]) == len(losses[False]) > 3
    np.testing.assert_allclose(losses[True], losses[False], rtol=1e-6)


def test_cmvae_sparse_samplers_through_the_public_consumers():
    """train() / evaluate() / one_plus_random() unpack ``(data, target)`` from the sparse=True samplers like from the
    dense ones and give the same numbers (reference consumers: models.py:401-422, evaluation.py:100-103, 157-161)"""
    from rectorch_amd.samplers import ConditionedDataSampler, EmptyConditionedDataSampler
    from rectorch_amd.evaluation import evaluate, one_plus_random
    from rectorch_amd.utils.hashinit import hash_state_dict
    rng = np.random.RandomState(5)
    U, I, C_, H, L = 48, 96, 3, 24, 8
    tr = csr_matrix((rng.rand(U, I) < 0.15).astype(np.float32))
    tr = csr_matrix(tr + csr_matrix((np.ones(U), (np.arange(U), np.arange(U) % I)), shape=(U, I)))
    tr.data[:] = 1.0
    te = csr_matrix((rng.rand(U, I) < 0.1).astype(np.float32))
    te = csr_matrix(te + csr_matrix((np.ones(U), (np.arange(U), (np.arange(U) * 5 + 1) % I)), shape=(U, I)))
    te.data[:] = 1.0
    iid2cids = {i: sorted({int(i % C_), int((i * 7) % C_)}) for i in range(I)}
    sd = hash_state_dict([I + C_, H, L], [L, H, I], "vae", 5, 1.0)
    res = {}
    for sparse in (False, True):
        torch.manual_seed(7)
        net, model = make_cvae(C_, [I, H, L], [L, H, I], 0.0, sd, beta=0.1, numerics="fp32")
        train_s = ConditionedDataSampler(iid2cids, C_, tr, None, batch_size=16, shuffle=False, sparse=sparse)
        valid_s = EmptyConditionedDataSampler(C_, tr, te, batch_size=16, shuffle=False, sparse=sparse)
        model.train(train_s, valid_s, "ndcg@10", num_epochs=2, best_path=os.path.join(tempfile.gettempdir(), "cmvae_sparse_%d.pth" % sparse),
                    verbose=1)
        ev = evaluate(model, valid_s, ["ndcg@10", "recall@5"])
        random.seed(3)
        opr = one_plus_random(model, valid_s, ["hit@3"], r=20)
        res[sparse] = (ev, opr, [p.detach().cpu().numpy().copy() for p in net.parameters()])
    for k in res[False][0]:
        assert res[True][0][k].shape == (U,)
        np.testing.assert_allclose(res[True][0][k], res[False][0][k], rtol=1e-6, atol=1e-7)
    for k in res[False][1]:
        np.testing.assert_allclose(res[True][1][k], res[False][1][k], rtol=1e-6, atol=1e-7)
    for a, b in zip(res[True][2], res[False][2]):
        np.testing.assert_allclose(a, b, rtol=1e-5, atol=1e-7)


def test_resident_sampler_tensor_edited_in_place_is_honoured():
    """a tensor yielded by the device DataSampler carries its CSR rows as a shortcut; once the caller edits the tensor in
    place the shortcut is stale and the CONTENTS must be used (reference models.py:619-624 uses x itself)"""
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.utils.hashinit import hash_state_dict
    rng = np.random.RandomState(11)
    U, I, H, L = 20, 130, 24, 8
    X = csr_matrix((rng.rand(U, I) < 0.2).astype(np.float32))
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 3, 1.0)
    net, model = make_vae([I, H, L], [L, H, I], 0.0, sd, beta=0.1)
    smp = DataSampler(X, None, batch_size=U, shuffle=False)
    x, _ = next(iter(smp))
    p0 = model.predict(x)[0].cpu().numpy()
    dense = torch.from_numpy(X.toarray()).cuda()
    np.testing.assert_array_equal(np.isneginf(p0), X.toarray() != 0)
    x[:, :40] = 0                                   # in-place edit: cold-start the first 40 items
    dense[:, :40] = 0
    p1 = model.predict(x)[0].cpu().numpy()
    p2 = model.predict(dense)[0].cpu().numpy()
    np.testing.assert_array_equal(np.isneginf(p1), dense.cpu().numpy() != 0)
    fin = np.isfinite(p2)
    np.testing.assert_allclose(p1[fin], p2[fin], rtol=1e-6, atol=1e-7)
    assert not np.allclose(p0[:, 60:][np.isfinite(p0[:, 60:]) & np.isfinite(p1[:, 60:])],
                           p1[:, 60:][np.isfinite(p0[:, 60:]) & np.isfinite(p1[:, 60:])])


def test_dp_path_world1_rccl():
    """the data-parallel step (RCCL all-reduce per bucket on a side stream, per-bucket Adam on a third stream, float32
    and bf16 exchange) against the single-GPU step, in its own process with a one-rank RCCL group"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([os.sys.executable, os.path.join(ROOT, "tests", "dp_world1_check.py"), str(port)],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "DP_WORLD1_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


# ------------------------------------------------------------------------------------------------ SVAE
def make_svae(g, **kw):
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    I, E, R, H, L, D = [int(v) for v in g["dims"]]
    net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_from(g, "sd0__").items()})
    net.to("cuda")
    return net, SVAE(net, **kw)


def test_g12_svae_predict_and_train_steps():
    """float32 throughout: logits within 1e-5 of the reference, parameters after Adam (weight decay 5e-3) within 5e-6"""
    g = load_golden("g12_svae_steps")
    beta, anneal, lr, wd = [float(v) for v in g["meta"]]
    net, model = make_svae(g, beta=beta, anneal_steps=int(anneal), learning_rate=lr)
    assert [k for k, _ in net.named_parameters()] == list(g["param_names"])
    assert model.optimizer.param_groups[0]["weight_decay"] == wd
    model._rtx.inject = (None, dev(g["pred_eps"]))
    pr, mu, lv = model.predict(torch.from_numpy(g["pred_x"]), remove_train=True)
    assert pr.shape == (1, int(g["dims"][0])) and mu.shape == g["pred_mu"].shape
    pr = pr.cpu().numpy()
    assert np.array_equal(np.isneginf(pr), np.isneginf(g["pred"]))
    fin = np.isfinite(pr)
    assert rel(pr[fin], g["pred"][fin]) < 1e-5 and rel(mu.cpu(), g["pred_mu"]) < 1e-5 and rel(lv.cpu(), g["pred_logvar"]) < 1e-5
    names = list(g["param_names"])
    for t in range(int(g["n_steps"])):
        model._rtx.inject = (None, dev(g["eps_%d" % t]))
        loss = model.train_batch(torch.from_numpy(g["x_%d" % t]), torch.from_numpy(g["y_%d" % t]))
        assert abs(loss - float(g["loss_%d" % t])) < 1e-5 * abs(float(g["loss_%d" % t])), (t, loss, float(g["loss_%d" % t]))
        for k, prm in zip(names, net._param_list()):
            gref = g["grad_%d__%s" % (t, k.replace(".", "__"))]
            assert rel(prm.grad.cpu(), gref) < 2e-4, (t, k)
            ref = g["sd_%d__%s" % (t, k.replace(".", "__"))]
            assert float(np.max(np.abs(prm.detach().cpu().numpy() - ref))) < 5e-6, (t, k)
    assert model.gradient_updates == float(g["n_steps"])
    model._rtx.inject = None


def test_svae_reference_api_train_save_load():
    """the reference's own test_SVAE (tests/test_models.py:491-546), with the network on the MI355X, plus the compact
    target form of the sampler"""
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    from rectorch_amd.samplers import SVAE_Sampler
    total_items = 7
    net = SVAE_net(n_items=total_items, embed_size=2, rnn_size=2, dec_dims=[2, total_items], enc_dims=[2, 2]).to("cuda")
    model = SVAE(net)
    assert model.learning_rate == 1e-3 and model.network == net and isinstance(model.optimizer, torch.optim.Adam)
    assert str(model) == repr(model)
    tr = {0: [0, 1, 2, 3, 4, 5, 6], 1: [6, 5, 4, 3, 2, 1, 0], 2: [2, 1, 6, 0, 3]}
    sampler = SVAE_Sampler(num_items=total_items, dict_data_tr=tr, dict_data_te=None, pred_type="next", k=2, shuffle=False,
                           is_training=True)
    x = torch.LongTensor([[1, 2, 5]])
    pr = model.predict(x, True)[0]
    assert pr.shape == (1, total_items) and torch.isneginf(pr[0, [1, 2, 5]]).all()
    torch.manual_seed(12345)
    out_1 = model.predict(x, False)[0]
    model.train(sampler, num_epochs=10, verbose=4)
    torch.manual_seed(12345)
    out_2 = model.predict(x, False)[0]
    assert not torch.all(out_1.eq(out_2))
    tmp = tempfile.NamedTemporaryFile()
    model.save_model(tmp.name, 1)
    net2 = SVAE_net(n_items=total_items, embed_size=2, rnn_size=2, dec_dims=[2, total_items], enc_dims=[2, 2]).to("cuda")
    model2 = SVAE(net2)
    model2.load_model(tmp.name)
    torch.manual_seed(12345)
    out_1 = model.predict(x, False)[0]
    torch.manual_seed(12345)
    out_2 = model2.predict(x, False)[0]
    assert torch.all(out_1.eq(out_2))
    # compact targets drive the same step as the dense ones
    losses = {}
    for sparse in (False, True):
        torch.manual_seed(5)
        n = SVAE_net(n_items=total_items, embed_size=2, rnn_size=2, dec_dims=[2, total_items], enc_dims=[2, 2])
        n.load_state_dict(net2.state_dict())
        m = SVAE(n.to("cuda"))
        smp = SVAE_Sampler(total_items, tr, None, pred_type="next_k", k=2, shuffle=False, sparse=sparse)
        losses[sparse] = [m.train_batch(a, b) for a, b in smp]
    np.testing.assert_allclose(losses[True], losses[False], rtol=1e-6)


@pytest.mark.parametrize("R", [96, 163])
def test_svae_vs_oracle_longer_sequences(R):
    """ml-1m-like widths (embedding 64, latent 32) and sequences of 1..300 steps against the numpy oracle: GRU 96 (the whole-row
    recurrence kernel) and GRU 163 -- the K-sliced kernels with a last K slice and last row groups that are only partly real"""
    from oracle.svae_oracle import SvaeOracle
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    torch.manual_seed(3)
    I, E, H, L, D = 500, 64, 80, 32, 72
    net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=0)
    orc = SvaeOracle(sd, n_enc=2, n_dec=2, beta=0.2)
    rng = np.random.RandomState(9)
    for T in (1, 37, 300):
        items = rng.randint(0, I, size=T)
        y = np.zeros((T, I), dtype=np.float32)
        for t in range(T):
            y[t, rng.choice(I, size=3, replace=False)] = 1.0
        eps = rng.randn(T, L).astype(np.float32)
        model._rtx.inject = (None, dev(eps))
        loss = model.train_batch(torch.from_numpy(items[None, :]), torch.from_numpy(y[None]))
        lo = orc.train_batch(items, y.astype(np.float64), eps.astype(np.float64))
        assert abs(loss - lo) < 2e-5 * abs(lo), (T, loss, lo)
        for k, prm in zip(orc.keys, net._param_list()):
            assert rel(prm.grad.cpu(), orc.last_grads[k]) < 5e-4, (T, k)
            # Adam's normalised step turns a gradient that is round-off noise around zero into a +-lr move, so single
            # elements may differ by a fraction of lr = 1e-3; everything else agrees to float32 round-off
            dlt = np.abs(prm.detach().cpu().numpy() - orc.p[k])
            assert float(dlt.max()) < 1e-3 and float(np.mean(dlt > 2e-5)) < 1e-4, (T, k, float(dlt.max()))


def test_svae_loss_mailbox_returns_this_steps_loss():
    """Round 6: SVAE.train_batch takes its return value (reference models.py:835 `return loss.item()`) from a host mailbox the loss
    kernel writes mid-step, so the host does not wait for the backward half.  Same losses and the same parameters (to float32 round-off) as
    with the draining read-back, over sequences of very different lengths back to back (a short step behind a long one: the ticket
    of the LAST step is the one waited for)."""
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    I, E, R, H, L, D = 400, 48, 96, 64, 24, 56
    rng = np.random.RandomState(4)
    seqs = []
    for T in (210, 3, 1, 97, 2, 300, 5):
        items = rng.randint(0, I, size=T)
        y = np.zeros((T, I), dtype=np.float32)
        for t in range(T):
            y[t, rng.choice(I, size=3, replace=False)] = 1.0
        seqs.append((items, y, rng.randn(T, L).astype(np.float32)))
    runs = {}
    for mailbox in (True, False):
        torch.manual_seed(21)
        net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
        model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=0)
        model.loss_mailbox = mailbox
        losses = []
        for items, y, eps in seqs:
            model._rtx.inject = (None, dev(eps))
            losses.append(model.train_batch(torch.from_numpy(items[None, :]), torch.from_numpy(y[None])))
        eng = net._svae_engine
        assert bool(getattr(eng, "_mailbox", False)) == mailbox
        torch.cuda.synchronize()
        runs[mailbox] = (losses, [p.detach().cpu().numpy().copy() for p in net._param_list()])
    # (not bit for bit: the bias and embedding gradients are atomic sums -- k_sv_colsum1, k_sv_embed_grad -- whose order varies from run to run)
    assert np.allclose(runs[True][0], runs[False][0], rtol=2e-6, atol=0), (runs[True][0], runs[False][0])
    for a, b in zip(runs[True][1], runs[False][1]):
        assert np.allclose(a, b, rtol=0, atol=2e-6)


@pytest.mark.parametrize("widths,lens", [((300, 48, 40, 36, 16, 28), (2, 9, 41, 17, 130, 3, 66)),
                                         # the benchmarked GRU width: the K-sliced recurrence kernels, one workgroup per user of the pack
                                         ((800, 256, 200, 150, 64, 150), (2, 9, 41, 17, 400, 3, 66))])
def test_svae_pack_of_users_vs_oracle(widths, lens):
    """SVAE_Sampler(pack=N) (not in the reference): ONE Adam step for the mean of the per-user losses of a pack of users with
    different lengths -- concatenated rows, one recurrence workgroup per user, per-row loss factors -- against the numpy
    oracle's gradient accumulation (loss, every gradient, parameters after two packs); a pack of ONE user computes what the
    reference's per-user step computes."""
    from oracle.svae_oracle import SvaeOracle
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    from rectorch_amd.engine import SvaePack
    from rectorch_amd.samplers import SVAE_Sampler
    torch.manual_seed(5)
    I, E, R, H, L, D = widths
    net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    model = SVAE(net.to("cuda"), beta=0.3, anneal_steps=0)
    orc = SvaeOracle(sd, n_enc=2, n_dec=2, beta=0.3)
    rng = np.random.RandomState(11)
    data = {u: rng.randint(0, I, size=n).tolist() for u, n in enumerate(lens)}
    smp = SVAE_Sampler(I, data, None, pred_type="next_k", k=3, shuffle=False, sparse=True, pack=4)
    packs = [p for p, _ in smp]
    assert len(packs) == len(smp) == 2 and sorted(u for p in packs for u in p.users) == list(range(7))
    assert [len(data[u]) for u in packs[0].users] == sorted(len(data[u]) for u in packs[0].users)     # grouped by length
    for p in packs:
        eps = rng.randn(p.n_steps, L).astype(np.float32)
        model._rtx.inject = (None, dev(eps))
        loss = model.train_batch(p, p)
        users, o = [], 0
        for u, n in zip(p.users, p.lens):
            rows = smp._target_rows(u)
            y = np.zeros((n, I))
            for t, r in enumerate(rows):
                y[t, r] = 1.0
            users.append((np.array(data[u][:-1]), y, eps[o:o + n].astype(np.float64)))
            o += n
        lo = orc.train_pack(users)
        assert abs(loss - lo) < 2e-5 * abs(lo), (loss, lo)
        for k, prm in zip(orc.keys, net._param_list()):
            assert rel(prm.grad.cpu(), orc.last_grads[k]) < 5e-4, k
            dlt = np.abs(prm.detach().cpu().numpy() - orc.p[k])
            assert float(dlt.max()) < 1e-3 and float(np.mean(dlt > 2e-5)) < 1e-4, (k, float(dlt.max()))
    # a pack of one user == the per-user step (same objective, same kernels; only the loss factors travel per row)
    net1 = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    net2 = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    net1.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    net2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
    m1, m2 = SVAE(net1.to("cuda"), beta=0.3, anneal_steps=0), SVAE(net2.to("cuda"), beta=0.3, anneal_steps=0)
    one = SVAE_Sampler(I, {0: data[4]}, None, pred_type="next_k", k=3, shuffle=False, sparse=True)
    (x, y), = list(one)
    eps = dev(rng.randn(x.numel(), L).astype(np.float32))
    m1._rtx.inject = m2._rtx.inject = (None, eps)
    l1 = m1.train_batch(x, y)
    pk = SvaePack([data[4][:-1]], [one._target_rows(0)])
    l2 = m2.train_batch(pk, pk)
    assert abs(l1 - l2) < 1e-6 * abs(l1)
    for a, b in zip(net1.parameters(), net2.parameters()):
        assert float((a.detach() - b.detach()).abs().max()) < 1e-6


def test_dp_world2_on_one_gpu():
    """two data-parallel ranks (gloo over device tensors) on the one GPU of the box: row sharding, bucketed exchange
    (float32 and bf16), per-bucket Adam -- three steps must land on the reference's parameters on both ranks"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [os.sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_world2_onegpu_check.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "DP_WORLD2_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_dp_world8_on_one_gpu():
    """EIGHT data-parallel ranks (gloo over device tensors) on the one GPU of the box, bf16 numerics and bf16 gradient images,
    engine-scheduled step, sharded and replicated optimizer: three steps land on the single-GPU result of the 8x batch within the
    bound stated in tests/dp_world8_onegpu_check.py, every rank holds bit-identical parameters, the engine's exchange
    statistics name the bytes (the metric's 8-GPU numerics, which the two-rank tests do not reach)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [os.sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "8", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "dp_world8_onegpu_check.py")]
    out = subprocess.run(cmd, capture_output=True, text=True, timeout=900)
    print(out.stdout[-1500:])
    assert out.returncode == 0 and "DP_WORLD8_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize("numerics", ["bf16", "fp32"])
def test_loss_mailbox_returns_this_steps_loss(numerics):
    """train_batch returns THIS step's loss (reference models.py:835 `return loss.item()`).  Default: the float comes from the
    engine's host mailbox (rtx_engine_wait_loss spins on the step count; the stream is not drained); `loss_mailbox = False`:
    `loss.item()`.  Same seeds -> the two models report bit-identical losses step by step and end with bit-identical parameters;
    the mailbox's value equals the device loss buffer's after a synchronise; waiting for a step that never ran times out."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd import _lib
    I, H, L, B = 3000, 600, 200, 128
    X = synth_interactions(6 * B, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=11)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 3)
    out = []
    for mailbox in (True, False):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics=numerics)
        net.to("cuda")
        model.loss_mailbox = mailbox
        torch.manual_seed(99)
        losses = []
        for rb in DataSampler(X, batch_size=B, shuffle=False).iter_rows():
            losses.append(model.train_batch(rb))
            assert isinstance(losses[-1], float) and np.isfinite(losses[-1])
            if mailbox:
                torch.cuda.synchronize()
                assert losses[-1] == float(model._rtx.loss_buf[0].item())
        eng = net._rtx_engines[numerics]
        assert bool(getattr(eng, "_mailbox", False)) == mailbox
        out.append((losses, [p.detach().cpu().numpy().copy() for p in net._param_list()]))
        if mailbox:
            # a step that never ran: the mailbox belongs to the LAST step enqueued (keyed by the engine's own ticket), said at once
            with pytest.raises(_lib.RtxError, match="holds the LAST step enqueued"):
                eng.wait_loss(10 ** 6, timeout_s=0.05)
            # a step count that RESTARTS on the same engine (a new trainer = a new optimizer around the same network, as after
            # reloading a checkpoint): step 1 of the new trainer must not be answered by the mailbox entry of the old trainer's step 1
            from rectorch_amd.models import MultiVAE
            model2 = MultiVAE(net, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics=numerics)
            model2.loss_mailbox = True
            assert net._rtx_engines[numerics] is eng
            for rb in list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())[:2]:
                got = model2.train_batch(rb)
                torch.cuda.synchronize()
                assert got == float(model2._rtx.loss_buf[0].item()) and got != losses[0]
    assert out[0][0] == out[1][0], (out[0][0], out[1][0])
    for a, b in zip(out[0][1], out[1][1]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))


def test_stress_folded_hops_under_foreign_load():
    """20 000 steps at the flagship shape with prefetch + deferred (folded) join + folded fork while a second
    process hammers the same GPU; an integer checksum of all parameters every 1 000 steps equals, bit for bit, the one of the same
    run on the plain event-only schedule (tests/stress_sync_check.py).  Reference semantics: models.py:409-419 (a step sees the
    previous step's completed update)."""
    out = subprocess.run([os.sys.executable, os.path.join(ROOT, "tests", "stress_sync_check.py")], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "STRESS OK" in out.stdout, out.stdout[-3000:] + out.stderr[-3000:]


def test_prefetched_batches_equal_self_gathered_batches():
    """rtx_engine_set_next_batch: the NEXT step's gather on the side stream under this step's last weight kernel, into a second
    batch image.  Same seeds -> the run that announces every next batch (train_epoch's default) ends with bit-identical
    parameters and loss sums to the run where every step gathers for itself; every step but the first starts from a prefetched
    image; a ragged last batch, an announced batch that is NOT the one that arrives, a prediction between two steps and a model
    that switches the hint off mid-run all land on the same bits."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 3000, 600, 200, 192
    X = synth_interactions(5 * B + 77, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=13)      # 5 full batches + a ragged one
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5)

    def fresh(prefetch):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics="bf16",
                              predict_numerics="bf16")      # (predictions run on the SAME engine, between two training steps)
        net.to("cuda")
        model.prefetch_batches = prefetch
        return net, model

    def params(net):
        return [p.detach().cpu().numpy().copy() for p in net._param_list()]

    # (1) two epochs through train_epoch (shuffled resident sampler, look-ahead inside) vs the same without the hint
    res = []
    for prefetch in (True, False):
        net, model = fresh(prefetch)
        np.random.seed(3)
        torch.manual_seed(17)
        smp = DataSampler(X, batch_size=B, shuffle=True)
        for ep in (1, 2):
            model.train_epoch(ep, smp, verbose=0)
        eng = net._rtx_engines["bf16"]
        hits, issued = eng.get_option("prefetch_hits"), eng.get_option("prefetch_issued")
        assert (hits, issued) == ((10, 10) if prefetch else (0, 0)), (prefetch, hits, issued)   # 2 x (6 batches - the first)
        res.append((params(net), model._read_loss_sum()))
    assert res[0][1] == res[1][1], (res[0][1], res[1][1])
    for a, b in zip(res[0][0], res[1][0]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))
    # (2) the announced batch is not the one that arrives; a prediction sits between two steps; the hint goes off mid-run
    batches = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
    order = [0, 1, 2, 5, 3, 4]
    res = []
    for variant in ("plain", "wrong-announcements"):
        net, model = fresh(variant != "plain")
        torch.manual_seed(23)
        for n, k in enumerate(order):
            nxt = None
            if variant != "plain":
                nxt = batches[(k + 1) % len(batches)]           # right for 0 -> 1 -> 2, wrong for 2 -> 5 -> 3, right for 3 -> 4
                if n == 4:
                    model.prefetch_batches = False
            model._fused_step(batches[k], None, want_loss=False, next_x=nxt)
            if n == 1:
                model.predict(batches[0].tr.gather_dense(batches[0].rows[:7]))
                net.train()
        res.append(params(net))
        if variant != "plain":
            eng = net._rtx_engines["bf16"]
            assert eng.get_option("prefetch_issued") == 4 and eng.get_option("prefetch_hits") == 2, \
                (eng.get_option("prefetch_issued"), eng.get_option("prefetch_hits"))
    for a, b in zip(res[0], res[1]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32))


def test_deferred_join_equals_joined_steps():
    """RTX_STEP_DEFER_JOIN (train_epoch's steps): no wait for the engine's side stream at the end of a step; the next step resolves
    the join inside its first-layer product (prefetched batch) or with a one-wave kernel (any other start), every other engine
    entry point and `_read_loss_sum` / `_join` resolve it first.  Same seeds: deferred and joined runs end in bit-identical
    parameters, losses and predictions -- with and without announced batches, with a prediction, a loss read-back and a
    checkpoint in the middle of the run."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 3000, 600, 200, 192
    X = synth_interactions(6 * B, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=21)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 9)
    batches = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
    outs = []
    for defer, announce in ((False, False), (True, False), (True, True), (False, True)):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics="bf16", predict_numerics="bf16")
        net.to("cuda")
        torch.manual_seed(31)
        res = []
        for t in range(12):
            nxt = batches[(t + 1) % 6] if announce else None
            model._fused_step(batches[t % 6], None, want_loss=False, next_x=nxt, defer_join=defer)
            if t == 3:      # an engine call of another kind right behind a deferred step
                res.append(model.predict(batches[0].tr.gather_dense(batches[0].rows[:9]))[0].cpu().numpy())
                net.train()
            if t == 6:      # the loss read-back joins first
                res.append(np.float64(model._read_loss_sum()))
            if t == 8:      # torch reads the decoder matrix (written on the side stream) after an explicit join
                model._join()
                res.append(net._param_list()[6].detach().cpu().numpy().copy())
        model._join()
        torch.cuda.synchronize()
        res += [p.detach().cpu().numpy().copy() for p in net._param_list()]
        outs.append(res)
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)


def test_dp_stream_ordered_ranks_on_one_gpu():
    """the data-parallel step at world 2 and 4 with collectives that are stream-ordered device work and nothing else (ranks =
    threads of one process, parallel.LocalGroup): the run WITHOUT any device drain equals the drained run bit for bit, replicas
    are identical, bucket A's collectives have a table of their own -- the ordering between the engine's kernels, the side
    stream's bucket and the caller's stream's bucket is what this test can see and the gloo tests cannot"""
    env = dict(os.environ, GPU_MAX_HW_QUEUES="8")
    out = subprocess.run([os.sys.executable, os.path.join(ROOT, "tests", "dp_local_threads_check.py")], capture_output=True, text=True,
                         timeout=1200, env=env)
    print(out.stdout[-1500:])
    assert out.returncode == 0 and "DP_LOCAL_THREADS_OK" in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


def test_bench_starts_its_own_ranks():
    """`python3 bench.py --gpus 2 --steps 3 --warmup 1` with NO launcher (the form the driver uses): bench.py starts the two ranks
    itself (gloo: they share the one GPU of the box), rank 0 prints the one JSON line, the exchange ran between 2 ranks and left
    bit-identical replicas"""
    import json
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["RTX_DIST_BACKEND"] = "gloo"
    out = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1"], capture_output=True,
                         text=True, timeout=900, env=env, cwd=ROOT)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-4000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["steps"] == 3 and d["warmup"] == 1
    assert d["replica_check"]["identical"] and d["replica_check"]["finite"] and d["replica_check"]["ranks"] == 2
    assert d["comm"] and d["comm"]["collectives_per_step"] > 0 and d["comm"]["bytes_per_step_per_rank"]["total"] > 0
    assert d["config"]["global_batch"] == 1000 and d["scaling"] == "weak" and d["value"] > 0
    # a rank that dies takes the job down with a non-zero exit and its own message (here: an unknown engine option)
    bad = subprocess.run(["python3", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--opt", "no_such_knob=1"],
                         capture_output=True, text=True, timeout=600, env=env, cwd=ROOT)
    assert bad.returncode != 0 and "no_such_knob" in bad.stderr and "rank" in bad.stderr, bad.stderr[-3000:]
    assert not [l for l in bad.stdout.strip().splitlines() if l.startswith("{")]


@pytest.mark.parametrize("seed", [1, 2, 3, 4, 5, 6])
def test_random_architectures_vs_oracle(seed):
    """randomly drawn networks (depths 1..3 per side, widths that are not multiples of anything, VAE / DAE / conditioned,
    ragged batches, dropout on or off, explicit targets or not): two training steps and a prediction against the C oracle"""
    from oracle import c_oracle
    from rectorch_amd.utils import hash_state_dict
    rng = np.random.RandomState(1000 + seed)
    I = int(rng.randint(70, 700))
    n_enc, n_dec = int(rng.randint(1, 4)), int(rng.randint(1, 4))
    L = int(rng.randint(3, 40))
    enc = [I] + [int(rng.randint(5, 150)) for _ in range(n_enc - 1)] + [L]
    dec = [L] + [int(rng.randint(5, 150)) for _ in range(n_dec - 1)] + [I]
    variant = "vae" if seed % 2 else "dae"
    cond = int(rng.randint(1, 9)) if (variant == "vae" and seed % 3 == 0) else 0
    p = float(rng.choice([0.0, 0.3, 0.5]))
    B = int(rng.randint(1, 140))
    use_gt = bool(cond) or bool(rng.randint(0, 2))
    enc_full = [I + cond] + enc[1:]
    sd = hash_state_dict(enc_full, dec, variant, 77 + seed, bias_std=0.2)
    params, keys = params_in_order(sd)
    if cond:
        net, model = make_cvae(cond, enc, dec, p, sd, beta=0.3, numerics="fp32")
        ref = c_oracle.OracleTrainer(enc, dec, params, "vae", p, 0.3, 0, lr=1e-3, cond_dim=cond)
    elif variant == "vae":
        net, model = make_vae(enc, dec, p, sd, beta=0.3, numerics="fp32")
        ref = c_oracle.OracleTrainer(enc, dec, params, "vae", p, 0.3, 0, lr=1e-3)
    else:
        net, model = make_dae(enc, dec, p, sd, lam=0.1, numerics="fp32")
        ref = c_oracle.OracleTrainer(enc, dec, params, "dae", p, lam=0.1, lr=1e-3)
    for t in range(2):
        x = (rng.rand(B, I) < 0.15).astype(np.float32) * rng.choice([1.0, 1.0, 2.0, 0.5], size=(B, I)).astype(np.float32)
        x[0, :] = 0.0                                                   # an empty row
        xin = x
        if cond:
            c = np.zeros((B, cond), dtype=np.float32)
            c[np.arange(B), rng.randint(0, cond, size=B)] = 1.0
            xin = np.concatenate([x, c], axis=1)
        gt = (x * (rng.rand(B, I) < 0.6)).astype(np.float32) if use_gt else None
        if variant == "dae":
            gt = None                                                   # MultiDAE.train_batch ignores te_batch
        mask = (rng.rand(B, I) >= p).astype(np.uint8)
        eps = rng.randn(B, L).astype(np.float32)
        model._rtx.inject = (dev(mask, torch.uint8), dev(eps) if variant == "vae" else None)
        loss = model.train_batch(torch.from_numpy(xin), None if gt is None else torch.from_numpy(gt))
        ref_loss = ref.train_batch(xin, gt, mask, eps if variant == "vae" else None)
        assert abs(loss - ref_loss) < 3e-5 * max(1.0, abs(ref_loss)), (seed, t, loss, ref_loss)
        for k, prm, gr in zip(keys, net._param_list(), ref.last["grads"]):
            if variant == "dae":
                break        # the engine folds the gradient of lam * sum ||W||_2 into its Adam kernel: p.grad holds the
                             # likelihood part only (the parameters below include it)
            scale = max(1e-6, float(np.max(np.abs(gr))))
            assert float(np.max(np.abs(prm.grad.cpu().numpy() - gr))) < 3e-4 * scale + 1e-7, (seed, t, k)
    for prm, r, k in zip(net._param_list(), ref.params, keys):
        d = np.abs(prm.detach().cpu().numpy() - r)
        assert float(d.max()) < 2.1e-3 and float(np.mean(d > 2e-5)) < 2e-3, (seed, k, float(d.max()), float(np.mean(d > 2e-5)))
    model._rtx.inject = None
    pred = model.predict(torch.from_numpy(xin), remove_train=True)[0].cpu().numpy()
    pref = ref.predict(xin, True)[0]
    assert np.array_equal(np.isneginf(pred), np.isneginf(pref))
    fin = np.isfinite(pred)
    # after two independent Adam trajectories the parameters differ by the few +-lr elements above: compare loosely
    assert rel(pred[fin], pref[fin]) < 5e-3


@pytest.mark.parametrize("unit,U,I", [(0.25, 500, 260), (0.125, 300, 140), (0.3, 400, 200)])
def test_ease_fractional_ratings(unit, U, I):
    """dyadic rating steps (quarter / eighth stars) are scaled to integers and take the MFMA Gram path; a step of 0.3 is not
    representable that way and takes the float64 path -- all equal to the numpy oracle"""
    from oracle.ease_oracle import ease_fit
    from rectorch_amd.engine import EaseSolver
    rng = np.random.RandomState(int(unit * 1000))
    X = ((rng.rand(U, I) < 0.12) * rng.randint(1, 9, size=(U, I))).astype(np.float32) * np.float32(unit)
    B = EaseSolver(csr_matrix(X.astype(np.float64)), 20.0).weights().cpu().numpy()
    Bo = ease_fit(X.astype(np.float64), 20.0)
    assert np.max(np.abs(B - Bo)) <= 1e-10 * max(1.0, np.max(np.abs(Bo)))


# ------------------------------------------------------------------------------------------------ round-2 parity holes
def test_g9_reference_checkpoint_predict_and_resume_on_device():
    """a checkpoint WRITTEN BY THE REFERENCE is loaded, predict() equals the reference's, and one more training step --
    which consumes the loaded Adam moments, step count and gradient_updates (annealed beta) -- lands on the reference's
    parameters and optimizer state (reference models.py:496-516, 905-908; golden G9 + g9_resume_step)"""
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.models import MultiVAE
    g = load_golden("g9_checkpoint_predict")
    r = load_golden("g9_resume_step")
    I, H, L = [int(v) for v in g["dims"]]
    net = MultiVAE_net([L, H, I], dropout=0.5)
    model = MultiVAE(net, beta=0.2, anneal_steps=5, numerics="fp32")
    ck = model.load_model(os.path.join(ROOT, "tests", "golden", "g9_reference_checkpoint.pth"))
    assert int(ck["epoch"]) == int(r["epoch"]) and model.gradient_updates == float(r["gradient_updates_after"]) - 1.0
    pred = model.predict(torch.from_numpy(g["x"]), remove_train=True)[0].cpu().numpy()
    assert np.array_equal(np.isneginf(pred), np.isneginf(g["pred"]))
    fin = np.isfinite(pred)
    assert rel(pred[fin], g["pred"][fin]) < 1e-5
    model._rtx.inject = (dev(r["mask"], torch.uint8), dev(r["eps"]))
    loss = model.train_batch(torch.from_numpy(r["x"]), None)
    assert abs(loss - float(r["loss"])) < 1e-5 * abs(float(r["loss"])), (loss, float(r["loss"]))
    assert model.gradient_updates == float(r["gradient_updates_after"])
    sd_t, keys = params_in_order(sd_from(r, "sd__"))
    for i, (k, prm, want) in enumerate(zip(keys, net._param_list(), sd_t)):
        assert float(np.max(np.abs(prm.detach().cpu().numpy() - want))) < 5e-6, k
        st = model.optimizer.state[prm]
        assert rel(st["exp_avg"].cpu(), r["exp_avg_%d" % i]) < 1e-4, k
        assert rel(st["exp_avg_sq"].cpu(), r["exp_avg_sq_%d" % i]) < 1e-4, k
    model._sync_optimizer_state()
    assert float(model.optimizer.state[net._param_list()[0]]["step"]) == float(r["step_after"])


def test_custom_op_train_step_dense_equals_train_batch():
    """torch.ops.rectorch_hip.train_step_dense (rectorch_amd/ops.py) drives the same C entry point as train_batch"""
    from rectorch_amd import ops  # noqa: F401
    from rectorch_amd.utils.hashinit import hash_state_dict
    rng = np.random.RandomState(2)
    I, H, L, B = 200, 32, 8, 24
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 7, 1.0)
    x = (rng.rand(B, I) < 0.1).astype(np.float32)
    x[:, 0] = 1
    out = []
    for use_op in (False, True):
        net, model = make_vae([I, H, L], [L, H, I], 0.0, sd, beta=0.3, numerics="fp32")
        xt = torch.from_numpy(x)
        if not use_op:
            torch.manual_seed(5)
            from rectorch_amd.nets import draw_seed
            seed = draw_seed()
            torch.manual_seed(5)
            loss = model.train_batch(xt)
        else:
            st, params, m, v = model._ensure_train_state()
            eng = net.rtx_engine("fp32", B, train_buffers=(st.grads, m, v))
            g = model.optimizer.param_groups[0]
            loss = float(torch.ops.rectorch_hip.train_step_dense(eng.op_handle, xt.cuda(), None, 0.3, 0.0, g["lr"], g["betas"][0], g["betas"][1],
                                                                 g["eps"], g["weight_decay"], 1, seed))
        out.append((loss, [p.detach().cpu().numpy().copy() for p in net.parameters()]))
    assert abs(out[0][0] - out[1][0]) < 1e-6 * abs(out[0][0])
    for a, b in zip(out[0][1], out[1][1]):
        np.testing.assert_allclose(a, b, rtol=0, atol=1e-7)
    assert len(ops._HANDLES) >= 0     # handles are weak: nothing to clean up


@pytest.mark.parametrize("numerics", ["fp32", "bf16"])
def test_config3_netflix_shape_two_steps_vs_oracle(numerics):
    """BASELINE.json configs[3] shape, one GPU's share (I = 17769, 512 users): two steps with injected masks / noise against the
    C oracle, parameters of all 8 tensors.  W1 is [600, 17769]: rows that are not a multiple of 4 floats -- the fused
    weight-gradient + Adam kernel takes them with dword-aligned 16-byte accesses and an element-wise last group (round 3;
    before: gradient store + flat multi-tensor Adam); W4 [17769, 600] the aligned form (bf16)."""
    from oracle import c_oracle
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 17769, 600, 200, 512
    X = synth_interactions(1024, I, mu=4.3, sigma=1.0, dmax=5000, seed=17)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 3, bias_std=0.05)
    params, keys = params_in_order(sd)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, numerics=numerics)
    model.keep_grads = True
    ref = c_oracle.OracleTrainer([I, H, L], [L, H, I], params, "vae", 0.5, beta=0.2, anneal_steps=0, lr=1e-3)
    gen = torch.Generator().manual_seed(5)
    for t, rb in enumerate(DataSampler(X, batch_size=B, shuffle=False).iter_rows()):
        mask = (torch.rand(B, I, generator=gen) >= 0.5).to(torch.uint8)
        eps = torch.randn(B, L, generator=gen)
        model._rtx.inject = (mask.cuda(), eps.cuda())
        loss = model._fused_step(rb, None, want_loss=True)
        dense = np.asarray(X[t * B:(t + 1) * B].toarray(), dtype=np.float32)
        ref_loss = ref.train_batch(dense, None, mask.numpy(), eps.numpy())
        worst = 0.0
        for k, prm, gr in zip(keys, net._param_list(), ref.last["grads"]):
            scale = max(1e-9, float(np.max(np.abs(gr))))
            err = float(np.max(np.abs(prm.grad.cpu().numpy() - gr))) / scale
            worst = max(worst, err)
            assert err < (1e-4 if numerics == "fp32" else 3e-2), (t, k, err)      # achieved: 2.3e-5 / 1.1e-2
        print("config3 %s step %d: loss rel %.2e, worst gradient rel %.2e" % (numerics, t, abs(loss - ref_loss) / abs(ref_loss), worst))
        assert abs(loss - ref_loss) < (1e-6 if numerics == "fp32" else 5e-6) * abs(ref_loss), (t, loss, ref_loss)   # achieved: 5e-8 / 2.5e-7
    for prm, r, k in zip(net._param_list(), ref.params, keys):
        d = np.abs(prm.detach().cpu().numpy() - r)
        print("config3 %s %-22s |dp| max %.2e mean %.2e frac>2e-5 %.2e frac>5e-4 %.2e" % (numerics, k, float(d.max()), float(d.mean()), float(np.mean(d > 2e-5)), float(np.mean(d > 5e-4))))
        if numerics == "fp32":
            # Adam's normalised step turns a gradient that is round-off noise around zero into a +-lr move (see config0)
            # achieved (round 3): max 1.4e-4, at most 1.7e-5 of a tensor's elements off by more than 2e-5
            assert float(d.max()) < 6e-4 and float(np.mean(d > 2e-5)) < 7e-5 and float(d.mean()) < 5e-9, (k, float(d.max()), float(np.mean(d > 2e-5)))
        else:
            # achieved (round 3): mean 0.4-5.0e-6, at most 0.13 % of a tensor's elements off by more than 5e-4 (round 2 accepted 3 %)
            assert float(d.max()) < 4.2e-3 and float(np.mean(d > 5e-4)) < 5e-3 and float(d.mean()) < 1.5e-5, (k, float(d.max()), float(np.mean(d > 5e-4)))


def test_config3_global_batch_4096_on_one_gpu_is_the_sum_of_its_shards():
    """configs[3] with the whole global batch on one GPU (B = 4096: eight 512-row tiles per GEMM): loss and gradients equal
    the sum over the eight 512-user shards (each checked against the oracle above) -- the data-parallel identity"""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.engine import RowBatch
    I, H, L, B = 17769, 600, 200, 4096
    X = synth_interactions(B, I, mu=4.3, sigma=1.0, dmax=5000, seed=18)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 3, bias_std=0.05)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, numerics="bf16")
    (rb,) = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
    st, params, m, v = model._ensure_train_state()
    eng = net.rtx_engine("bf16", B, train_buffers=(st.grads, m, v))
    gen = torch.Generator().manual_seed(6)
    mask = (torch.rand(B, I, generator=gen) >= 0.5).to(torch.uint8).cuda()
    eps = torch.randn(B, L, generator=gen).cuda()
    kw = dict(beta=0.1, lam=0.0, inv_batch=1.0 / B, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1)
    loss = torch.zeros(2, device="cuda")
    eng.loss_grads(rb, None, eng._step(mask=mask, noise=eps, **kw), loss[0:1])
    full, full_loss = st.flat_grads.clone(), loss[0].item()
    acc, part_loss = torch.zeros_like(full), 0.0
    for s in range(0, B, 512):
        sub = RowBatch(rb.tr, None, rb.rows[s:s + 512].contiguous())
        eng.loss_grads(sub, None, eng._step(mask=mask[s:s + 512].contiguous(), noise=eps[s:s + 512].contiguous(), **kw), loss[0:1])
        acc += st.flat_grads
        part_loss += loss[0].item()
    assert np.isfinite(full_loss) and abs(part_loss - full_loss) < 1e-4 * abs(full_loss)
    assert float((acc - full).abs().max() / full.abs().max()) < 4e-3
    # and the fused step at this batch trains
    torch.manual_seed(0)
    losses = [model._fused_step(rb, None, want_loss=True) for _ in range(4)]
    assert np.isfinite(losses).all() and losses[-1] < losses[0]


def test_ease_full_size_kkt_properties():
    """EASE at the ml-20m shape (136 677 x 20 108): the launches the smaller tests never reach (128x128-tile f64 GEMM of
    the two top recursion levels, the 79-tile-row Gram kernel).  Size-independent properties: diag(B) = 0 and
    (G + lam I)(I - B) is diagonal, checked on sampled columns with G applied through the sparse matrix in float64 on the host;
    sampled score rows against rows of X times the downloaded B."""
    from rectorch_amd.engine import CsrMatrix, EaseSolver
    from rectorch_amd.utils import synth_interactions
    U, I, lam = 136677, 20108, 500.0
    X = synth_interactions(U, I, seed=20)
    s = EaseSolver(CsrMatrix(X), lam)
    B = s.weights()
    assert float(torch.diagonal(B).abs().max()) == 0.0
    rng = np.random.RandomState(0)
    cols = rng.choice(I, size=24, replace=False)
    Bc = B[:, torch.as_tensor(cols, device="cuda")].cpu().numpy()
    worst = 0.0
    for q, j in enumerate(cols):
        w = -Bc[:, q]
        w[j] += 1.0                               # column j of (I - B)
        r = X.T @ (X @ w) + lam * w               # (G + lam I) w without forming G on the host
        d = r[j]
        r[j] = 0.0
        worst = max(worst, float(np.max(np.abs(r)) / abs(d)))
    assert worst < 1e-10, worst
    ids = rng.choice(U, size=64, replace=False)
    sc = s.scores(ids).cpu().numpy()
    ref = X[ids] @ B.cpu().numpy()
    assert float(np.max(np.abs(sc - ref))) < 1e-9 * max(1.0, float(np.max(np.abs(ref))))


def test_svae_vs_oracle_ml1m_widths():
    """SVAE at the benchmarked ml-1m widths (3 416 items, embedding 256, GRU 200 -> 600 of the 1 024 recurrence threads
    active, split-K path of the [T, 150] x [150, 3416]-sized products) against the numpy oracle"""
    from oracle.svae_oracle import SvaeOracle
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    torch.manual_seed(4)
    I, E, R, H, L, D = 3416, 256, 200, 150, 64, 150
    net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=0)
    orc = SvaeOracle(sd, n_enc=2, n_dec=2, beta=0.2)
    rng = np.random.RandomState(10)
    for T in (154, 40):
        items = rng.randint(0, I, size=T)
        y = np.zeros((T, I), dtype=np.float32)
        for t in range(T):
            y[t, rng.choice(I, size=4, replace=False)] = 1.0
        eps = rng.randn(T, L).astype(np.float32)
        model._rtx.inject = (None, dev(eps))
        loss = model.train_batch(torch.from_numpy(items[None, :]), torch.from_numpy(y[None]))
        lo = orc.train_batch(items, y.astype(np.float64), eps.astype(np.float64))
        assert abs(loss - lo) < 2e-5 * abs(lo), (T, loss, lo)
        for k, prm in zip(orc.keys, net._param_list()):
            assert rel(prm.grad.cpu(), orc.last_grads[k]) < 5e-4, (T, k)
            dlt = np.abs(prm.detach().cpu().numpy() - orc.p[k])
            assert float(dlt.max()) < 1e-3 and float(np.mean(dlt > 2e-5)) < 1e-4, (T, k, float(dlt.max()))


def test_svae_bf16_products_vs_oracle():
    """SVAE(numerics="bf16") -- BASELINE.json configs[4]'s dtype: every matrix product with bf16 operands and float32 accumulation
    (k_sv_gemm<true>), recurrences / loss / Adam in float32 -- against the float64 oracle at the benchmarked widths: loss and every
    gradient within bf16 rounding of the oracle's (the achieved figures are printed; the bounds are ~4x them), and not bit-equal
    to the float32 mode (the option really changes the arithmetic)."""
    from oracle.svae_oracle import SvaeOracle
    from rectorch_amd.nets import SVAE_net
    from rectorch_amd.models import SVAE
    torch.manual_seed(14)
    I, E, R, H, L, D = 800, 256, 200, 150, 64, 150
    ref = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
    sd = {k: v.detach().numpy().copy() for k, v in ref.state_dict().items()}
    rng = np.random.RandomState(21)
    T = 120
    items = rng.randint(0, I, size=T)
    y = np.zeros((T, I), dtype=np.float32)
    for t in range(T):
        y[t, rng.choice(I, size=4, replace=False)] = 1.0
    eps = rng.randn(T, L).astype(np.float32)
    orc = SvaeOracle(sd, n_enc=2, n_dec=2, beta=0.2)
    lo = orc.train_batch(items, y.astype(np.float64), eps.astype(np.float64))
    out = {}
    for mode in ("fp32", "bf16"):
        net = SVAE_net(n_items=I, embed_size=E, rnn_size=R, dec_dims=[L, D, I], enc_dims=[R, H, L])
        net.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})
        model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=0, numerics=mode)
        model._rtx.inject = (None, dev(eps))
        loss = model.train_batch(torch.from_numpy(items[None, :]), torch.from_numpy(y[None]))
        grads = {k: prm.grad.detach().cpu().numpy().copy() for k, prm in zip(orc.keys, net._param_list())}
        out[mode] = (loss, grads)
    l32, g32 = out["fp32"]
    l16, g16 = out["bf16"]
    e32 = max(rel(g32[k], orc.last_grads[k]) for k in orc.keys)
    e16 = {k: rel(g16[k], orc.last_grads[k]) for k in orc.keys}
    print("svae bf16 products: loss rel err %.2e (fp32 mode %.2e), worst gradient rel err %.2e at %s (fp32 mode %.2e)" % (
        abs(l16 - lo) / abs(lo), abs(l32 - lo) / abs(lo), max(e16.values()), max(e16, key=e16.get), e32))
    assert abs(l32 - lo) < 2e-5 * abs(lo) and e32 < 5e-4
    assert abs(l16 - lo) < 5e-5 * abs(lo), (l16, lo)          # achieved 8.7e-6
    assert max(e16.values()) < 3e-2, e16                      # achieved 7.9e-3 (the embedding's gradient)
    assert l16 != l32 and any(not np.array_equal(g16[k], g32[k]) for k in orc.keys)


def test_c_abi_rccl_hooks_one_rank():
    """rtx_comm_* (RCCL bound at run time by librectorch_hip): a one-rank communicator built from the C ABI alone; the in-place
    all-reduce / reduce-scatter / all-gather leave a one-rank buffer unchanged and the data-parallel step written with them
    (loss_grads -> rtx_comm_allreduce_many of the bound gradient buffers -> apply_adam) equals train_step"""
    import ctypes as C
    from rectorch_amd import _lib
    from rectorch_amd.utils.hashinit import hash_state_dict
    L_ = _lib.lib()
    ident = (C.c_uint8 * 128)()
    _lib.check(L_.rtx_comm_unique_id(ident))
    comm = C.c_void_p()
    _lib.check(L_.rtx_comm_init(ident, 0, 1, C.byref(comm)))
    r, w = C.c_int32(-1), C.c_int32(-1)
    _lib.check(L_.rtx_comm_rank(comm, C.byref(r), C.byref(w)))
    assert (r.value, w.value) == (0, 1)
    st = _lib.stream_ptr()
    x = torch.arange(4096, dtype=torch.float32, device="cuda") * 0.5
    ref = x.clone()
    _lib.check(L_.rtx_comm_allreduce(comm, C.c_void_p(x.data_ptr()), x.numel(), _lib.RTX_FP32, st))
    _lib.check(L_.rtx_comm_reduce_scatter(comm, C.c_void_p(x.data_ptr()), x.numel(), _lib.RTX_FP32, st))
    _lib.check(L_.rtx_comm_allgather(comm, C.c_void_p(x.data_ptr()), x.numel() * 4, st))
    xb = ref.to(torch.bfloat16)
    _lib.check(L_.rtx_comm_allreduce(comm, C.c_void_p(xb.data_ptr()), xb.numel(), _lib.RTX_BF16, st))
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and torch.equal(xb, ref.to(torch.bfloat16))
    # the data-parallel step through the C ABI only
    I, H, L, B = 300, 40, 12, 32
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 9, 1.0)
    rng = np.random.RandomState(1)
    xs = torch.from_numpy((rng.rand(B, I) < 0.1).astype(np.float32))
    outs = []
    for use_comm in (False, True):
        net, model = make_vae([I, H, L], [L, H, I], 0.0, sd, beta=0.3, numerics="fp32")
        stt, params, m, v = model._ensure_train_state()
        eng = net.rtx_engine("fp32", B, train_buffers=(stt.grads, m, v))
        g = model.optimizer.param_groups[0]
        step = eng._step(seed=77, beta=0.3, lam=0.0, inv_batch=1.0 / B, lr=g["lr"], beta1=g["betas"][0], beta2=g["betas"][1], eps=g["eps"],
                         weight_decay=g["weight_decay"], step=1)
        loss = torch.zeros(1, device="cuda")
        if not use_comm:
            eng.train_step(xs.cuda(), None, step, loss)
        else:
            eng.loss_grads(xs.cuda(), None, step, loss)
            n = len(stt.grads)
            bufs = (C.c_void_p * n)(*[t.data_ptr() for t in stt.grads])
            cnts = (C.c_int64 * n)(*[t.numel() for t in stt.grads])
            _lib.check(L_.rtx_comm_allreduce_many(comm, bufs, cnts, n, _lib.RTX_FP32, st))
            eng.apply_adam(step)
        torch.cuda.synchronize()
        outs.append((float(loss.item()), [p.detach().cpu().numpy().copy() for p in net.parameters()]))
    assert outs[0][0] == outs[1][0]
    for a, b in zip(outs[0][1], outs[1][1]):
        assert np.array_equal(a, b)
    _lib.check(L_.rtx_comm_destroy(comm))


@pytest.mark.parametrize("case", ["vae-mid", "dae-deep-odd", "vae-head-first"])
def test_bf16_fast_paths_match_the_generic_kernels(case):
    """bf16 numerics: the sparse first layer (spmm_in.hip) and the one-launch hidden layers of both passes (small_layers.hip)
    against the dense split-K product and the GEMM + post-kernel chain they replace -- same batches, same Philox draws, three
    training steps and a prediction.  Both sides multiply the same bf16 operands; only the order of the float32 additions
    differs, so the trajectories agree far inside the bf16 tolerance of the fp32 reference."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    enc, dec, variant, p = {"vae-mid": ([3000, 600, 200], [200, 600, 3000], "vae", 0.5),
                            "dae-deep-odd": ([777, 77, 33], [33, 50, 777], "dae", 0.3),
                            "vae-head-first": ([500, 64], [64, 120, 500], "vae", 0.5)}[case]
    I, B = enc[0], 130
    X = synth_interactions(3 * B, I, mu=3.2, sigma=0.9, dmax=I // 2, seed=11)
    sd = hash_state_dict(enc, dec, variant, 5, bias_std=0.1)

    def run(opts):
        if variant == "vae":
            net, model = make_vae(enc, dec, p, sd, beta=0.2, anneal_steps=10, numerics="bf16")
        else:
            net, model = make_dae(enc, dec, p, sd, lam=0.05, numerics="bf16")
        st, params, m, v = model._ensure_train_state()
        eng = net.rtx_engine("bf16", B, train_buffers=(st.grads, m, v))
        for k, val in opts.items():
            eng.set_option(k, val)
        smp = DataSampler(X, batch_size=B, shuffle=False)
        rbs = list(smp.iter_rows())
        torch.manual_seed(3)
        losses = [model._fused_step(rbs[i % len(rbs)], None, want_loss=True) for i in range(3)]
        pred = model.predict(smp._csr_tr.gather_dense(rbs[0].rows))[0].cpu().numpy()
        return losses, [q.detach().cpu().numpy().copy() for q in net._param_list()], pred

    fast = run({"sparse_in": 1})
    slow = run({"sparse_in": 0, "small_fwd": 0, "small_bwd": 0})
    for a, b in zip(fast[0], slow[0]):
        assert abs(a - b) < 2e-4 * abs(b), (case, fast[0], slow[0])
    for a, b in zip(fast[1], slow[1]):
        d = np.abs(a - b)
        # three Adam steps of lr = 1e-3: an element whose tiny gradient changes sign moves by up to 2 lr per step
        assert float(d.max()) <= 6.1e-3 and float(np.mean(d > 2e-5)) < 1e-2, (case, float(d.max()), float(np.mean(d > 2e-5)))
    assert np.array_equal(np.isneginf(fast[2]), np.isneginf(slow[2]))
    fin = np.isfinite(slow[2])
    assert rel(fast[2][fin], slow[2][fin]) < 5e-3
    # round 6: the launch shapes of the one-launch hidden layers -- 16-row waves per workgroup (1 is the default, 4 the form of rounds
    # 3-5) and the K range of a block shared by 2 or 4 waves (partial accumulators combined through LDS) -- against the same generic chain
    try:
        for knobs in ({"small_waves": 4}, {"small_kw": 2}, {"small_kw": 4, "small_waves": 2}):
            alt = run({"sparse_in": 0, **knobs})
            for a, b in zip(alt[0], slow[0]):
                assert abs(a - b) < 2e-4 * abs(b), (case, knobs, alt[0], slow[0])
            for a, b in zip(alt[1], slow[1]):
                d = np.abs(a - b)
                assert float(d.max()) <= 6.1e-3 and float(np.mean(d > 2e-5)) < 1e-2, (case, knobs, float(d.max()), float(np.mean(d > 2e-5)))
            assert rel(alt[2][fin], slow[2][fin]) < 5e-3, (case, knobs)
    finally:
        run({"small_waves": 1, "small_kw": 1})      # (process-wide knobs: back to the defaults for the tests that follow)


def test_batch_image_by_scatter_equals_the_full_rewrite():
    """round 4: with a resident matrix the dense batch image is kept all-zero between batches and only the stored entries are
    cleared / rewritten (k_gather_scatter).  A sequence of batches through ONE engine -- different users, a smaller batch (fewer
    padded rows), a DENSE tensor batch in between (full rewrite: the lists go stale and the next sparse batch resets the image),
    training steps with dropout in between -- must give bit-identical predictions and parameters with the scatter form on and off."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L = 1500, 96, 24
    X = synth_interactions(700, I, mu=3.0, sigma=1.0, dmax=I // 3, seed=3)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 9, bias_std=0.1)
    outs = []
    for scatter in (1, 0):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, numerics="bf16", predict_numerics="bf16")
        st, _, m, v = model._ensure_train_state()
        eng = net.rtx_engine("bf16", 130, train_buffers=(st.grads, m, v))
        eng.set_option("gather_scatter", scatter)
        smp = DataSampler(X, batch_size=130, shuffle=False)
        rbs = list(smp.iter_rows())            # 5 batches of 130 + one of 50
        res = []
        torch.manual_seed(5)
        res.append(model.predict(rbs[0])[0].cpu().numpy())
        res.append(model.predict(rbs[5])[0].cpu().numpy())                       # 50 users: fewer padded rows
        model._fused_step(rbs[1], None, want_loss=True)                          # training batches (dropout) in between
        res.append(model.predict(smp._csr_tr.gather_dense(rbs[2].rows))[0].cpu().numpy())   # dense tensor: the full-rewrite kernel
        model._fused_step(rbs[3], None, want_loss=True)
        res.append(model.predict(rbs[4])[0].cpu().numpy())
        res.append(model.predict(rbs[0])[0].cpu().numpy())
        res += [q.detach().cpu().numpy().copy() for q in net._param_list()]
        outs.append(res)
    for a, b in zip(*outs):
        assert np.array_equal(a, b)


def test_stream_value_sequence_restart():
    """round 4: the step's two cross-stream dependencies are hipStreamWriteValue32 / hipStreamWaitValue32 pairs with a growing
    sequence number; long before 2^31 both streams drain and the words restart from zero.  With the restart forced every second
    step ("hop_wrap" = 3), with the plain value form, with events instead ("hop_values" = 0) and with round 5's third form -- the
    dependency as two one-wave kernels (k_hop_set / k_hop_wait, "hop_kernels" = 1; measured, not faster, not the default) and the
    default: the fork folded into the data-gradient product ("hop_fold" = 1: that kernel's first instruction stores the number the
    side stream's k_hop_wait spins on; the caller's stream gets no packet of its own) -- eight training steps end in bit-identical
    parameters and losses.  The network is big enough (2200 x 512 > 2^20 elements) for the decoder matrix's kernel to run on the
    side stream, so both hops of the step carry a real dependency."""
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L = 2200, 512, 24
    X = synth_interactions(400, I, mu=3.0, sigma=1.0, dmax=I // 3, seed=4)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 11, bias_std=0.1)
    outs = []
    for opts in ({"hop_wrap": 3, "hop_fold": 0}, {"hop_fold": 0}, {"hop_values": 0, "hop_fold": 0}, {"hop_kernels": 1, "hop_fold": 0}, {}):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, numerics="bf16", predict_numerics="bf16")
        st, _, m, v = model._ensure_train_state()
        eng = net.rtx_engine("bf16", 100, train_buffers=(st.grads, m, v))
        for k, val in opts.items():
            eng.set_option(k, val)
        assert eng.get_option("hop_kernels") == opts.get("hop_kernels", 0)
        rbs = list(DataSampler(X, batch_size=100, shuffle=False).iter_rows())
        torch.manual_seed(6)
        losses = [float(model._fused_step(rbs[i % 4], None, want_loss=True)) for i in range(8)]
        torch.cuda.synchronize()
        outs.append(losses + [q.detach().cpu().numpy().copy() for q in net._param_list()])
    for other in outs[1:]:
        for a, b in zip(outs[0], other):
            assert np.array_equal(a, b)


# ---------------------------------------------------------------------------------------------- round 3: the benchmarked shape
@pytest.mark.parametrize("numerics", ["fp32", "bf16", "bf16-sparse-in"])
def test_ml20m_shape_b500_two_steps_vs_oracle(numerics):
    """BASELINE.json configs[1] exactly as bench.py times it -- I = 20108, B = 500, the DEFAULT step configuration (bf16: dense MFMA
    first layer, half-precision logits, one-launch hidden layers, the 1580 + 60-tile grouped weight-gradient + Adam launch with the
    encoder matrix and the transposed hidden copies, the decoder matrix's kernel on the side stream; gradients never stored) -- two
    steps with injected dropout masks / noise against oracle/mvae_oracle.c: loss, all eight parameter tensors, exp_avg and
    exp_avg_sq (reference models.py:817-835).  "bf16-sparse-in": the same with the first layer as the sparse VALU product
    (bench.py's `first_layer_sparse_valu` line).  Prints the achieved errors."""
    sparse_in = numerics.endswith("sparse-in")
    numerics = numerics.split("-")[0]
    from oracle import c_oracle
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 20108, 600, 200, 500
    X = synth_interactions(2 * B, I, seed=23)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 41, bias_std=0.05)
    params, keys = params_in_order(sd)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=0, numerics=numerics)
    assert model.keep_grads is False
    if sparse_in:
        st_, _, m_, v_ = model._ensure_train_state()
        net.rtx_engine(numerics, B, train_buffers=(st_.grads, m_, v_)).set_option("sparse_in", 1)
    ref = c_oracle.OracleTrainer([I, H, L], [L, H, I], params, "vae", 0.5, beta=0.2, anneal_steps=0, lr=1e-3)
    gen = torch.Generator().manual_seed(11)
    fp32 = numerics == "fp32"
    for t, rb in enumerate(DataSampler(X, batch_size=B, shuffle=False).iter_rows()):
        mask = (torch.rand(B, I, generator=gen) >= 0.5).to(torch.uint8)
        eps = torch.randn(B, L, generator=gen)
        model._rtx.inject = (mask.cuda(), eps.cuda())
        loss = model._fused_step(rb, None, want_loss=True)
        dense = np.asarray(X[t * B:(t + 1) * B].toarray(), dtype=np.float32)
        ref_loss = ref.train_batch(dense, None, mask.numpy(), eps.numpy())
        print("%s step %d: loss hip %.6f oracle %.6f (rel %.2e)" % (numerics, t, loss, ref_loss, abs(loss - ref_loss) / abs(ref_loss)))
        assert abs(loss - ref_loss) < (2e-5 if fp32 else 2e-3) * abs(ref_loss), (t, loss, ref_loss)
    torch.cuda.synchronize()
    lr = 1e-3
    for prm, r, rm, rv, k in zip(net._param_list(), ref.params, ref.m, ref.v, keys):
        state = model.optimizer.state[prm]
        d = np.abs(prm.detach().cpu().numpy() - r)
        m_hip, v_hip = state['exp_avg'].cpu().numpy(), state['exp_avg_sq'].cpu().numpy()
        em = float(np.max(np.abs(m_hip - rm))) / max(1e-30, float(np.max(np.abs(rm))))
        ev = float(np.max(np.abs(v_hip - rv))) / max(1e-30, float(np.max(np.abs(rv))))
        frac = float(np.mean(d > (2e-5 if fp32 else 5e-4)))
        print("%s %-22s |dp| max %.2e mean %.2e frac>%s %.2e | exp_avg rel %.2e | exp_avg_sq rel %.2e"
              % (numerics, k, float(d.max()), float(d.mean()), "2e-5" if fp32 else "5e-4", frac, em, ev))
        # Adam's normalised step turns a gradient that is round-off noise around zero into a +-lr move: two trajectories can part
        # by 2 lr per step on such a parameter; the statistics say how rare that is.  Bounds = about 4x the errors achieved on
        # MI355X (round 3: fp32 |dp| max 4.5e-5, moments 1e-5; bf16 |dp| mean 2.6e-6, 0.1 % of the parameters off by > 5e-4,
        # moments 5e-3)
        assert float(d.max()) <= 2 * 2 * lr * 1.05, (k, float(d.max()))
        if fp32:
            assert float(d.max()) < 2e-4 and frac < 1e-5 and float(d.mean()) < 5e-9, (k, float(d.max()), frac, float(d.mean()))
            assert em < 5e-5 and ev < 5e-5, (k, em, ev)
        else:
            assert frac < 5e-3 and float(d.mean()) < 1.2e-5, (k, frac, float(d.mean()))
            # (round 6: tightened from 2.5e-2 towards what is achieved; the approximate sqrt / rcp of the fused epilogue has no part in it --
            #  tests/native/test_gemm.cpp run_adam_approx_isolation holds that to 5e-7 of the parameters' motion -- this is operand rounding)
            # achieved on MI355X: exp_avg <= 1.3e-2 (the 200 -> 600 decoder matrix; 5e-3 elsewhere), exp_avg_sq <= 5.7e-3
            assert em < 2e-2 and ev < 1e-2, (k, em, ev)
    assert model._rtx.adam_step == 2
    if not fp32:
        assert bool(net._rtx_engines[numerics].get_option("last_sparse_in")) == sparse_in


def _grouped_interactions(n_users, n_items, n_groups, seed):
    """ml-20m-shaped synthetic rows with structure a recommender can learn: user u belongs to group u % n_groups and every group
    ranks the items by its own permutation of the Zipf popularity law (rectorch_amd.utils.synth_interactions draws the rows)"""
    from rectorch_amd.utils import synth_interactions
    X = synth_interactions(n_users, n_items, seed=seed).tocsr()
    rng = np.random.default_rng(seed + 1)
    perms = np.stack([rng.permutation(n_items) for _ in range(n_groups)])
    rows = np.repeat(np.arange(n_users), np.diff(X.indptr))
    cols = perms[rows % n_groups, X.indices]
    Y = csr_matrix((np.ones(cols.size), (rows, cols)), shape=X.shape)
    Y.sum_duplicates()
    Y.sort_indices()
    return Y


def test_trained_model_ndcg_recall_parity_bf16_vs_cpu_port():
    """The metric's second half ("nDCG@100 parity") for the HEADLINE mode: MultiVAE [20108, 600, 200], B = 500, bf16 training on the
    HIP path vs the same K = 120 steps of the reference's op sequence on the CPU (oracle/rectorch_cpu.py, float32), same hash init,
    same dropout masks and noise (the CPU port draws them from torch's generator; they are replayed from the seed and injected),
    then evaluate() on 1000 held-out users (80/20 item split): mean nDCG@100 and Recall@50 within 1e-2 relative, the smoothed loss
    curves within 1 %.  Reference: models.py:817-835 -> evaluation.py:100-106 -> metrics.py:136-147, 187-196."""
    import torch.nn.functional as F
    from oracle.rectorch_cpu import CpuNet, CpuTrainer
    from rectorch_amd.utils import hash_state_dict
    from rectorch_amd.utils.synth import split_heldout
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.evaluation import evaluate
    from rectorch_amd.metrics import Metrics
    I, H, L, B, K = 20108, 600, 200, 500, 120
    n_train, n_val = 8000, 1000
    X = _grouped_interactions(n_train + n_val, I, 24, seed=77)
    train, held = X[:n_train], X[n_train:]
    val_tr, val_te = split_heldout(held, 0.2, seed=5)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 2024, bias_std=0.05)
    params, _ = params_in_order(sd)
    net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, anneal_steps=1000, learning_rate=1e-3, numerics="bf16")
    cnet = CpuNet([I, H, L], [L, H, I], "vae", 0.5)
    cnet.load_numpy(params)
    ctr = CpuTrainer(cnet, beta=0.2, anneal_steps=1000, lr=1e-3)
    torch.set_num_threads(max(1, min(64, (os.cpu_count() or 8))))
    np.random.seed(4242)
    smp = DataSampler(train, batch_size=B, shuffle=True)
    hip_losses, cpu_losses = [], []
    t = 0
    while t < K:
        for rb in smp.iter_rows():
            if t >= K or len(rb) < B:
                break
            seed = 5000 + t
            rows = rb.rows.cpu().numpy()
            x = torch.from_numpy(np.asarray(train[rows].toarray(), dtype=np.float32))
            torch.manual_seed(seed)
            cpu_losses.append(ctr.train_batch(x))
            torch.manual_seed(seed)                      # replay: the draws the CPU step just consumed (dropout first, then eps)
            mask = (F.dropout(torch.ones(B, I), 0.5, True) != 0).to(torch.uint8)
            eps = torch.randn(B, L)
            model._rtx.inject = (mask.cuda(), eps.cuda())
            hip_losses.append(model._fused_step(rb, None, want_loss=True))
            t += 1
    model._rtx.inject = None
    hip_losses, cpu_losses = np.array(hip_losses), np.array(cpu_losses)
    w = 10
    sm = lambda a: np.convolve(a, np.ones(w) / w, mode="valid")
    curve = float(np.max(np.abs(sm(hip_losses) - sm(cpu_losses)) / np.abs(sm(cpu_losses))))
    res = evaluate(model, DataSampler(val_tr, val_te, batch_size=500, shuffle=False), ["ndcg@100", "recall@50"])
    xo = torch.from_numpy(np.asarray(val_tr.toarray(), dtype=np.float32))
    lo = ctr.predict(xo, True)[0].numpy()
    ro = Metrics.compute(lo, np.asarray(val_te.toarray(), dtype=np.float32), ["ndcg@100", "recall@50"])
    nd_h, nd_c = float(np.mean(res["ndcg@100"])), float(np.mean(ro["ndcg@100"]))
    rc_h, rc_c = float(np.mean(res["recall@50"])), float(np.mean(ro["recall@50"]))
    print("trained %d steps: loss first %.4f/%.4f last-20 mean hip %.4f cpu %.4f | smoothed curve max rel diff %.2e"
          % (K, hip_losses[0], cpu_losses[0], hip_losses[-20:].mean(), cpu_losses[-20:].mean(), curve))
    print("nDCG@100 hip(bf16) %.5f cpu(f32) %.5f rel %.2e | Recall@50 hip %.5f cpu %.5f rel %.2e | per-user nDCG max |d| %.3f"
          % (nd_h, nd_c, abs(nd_h - nd_c) / nd_c, rc_h, rc_c, abs(rc_h - rc_c) / rc_c,
             float(np.max(np.abs(res["ndcg@100"] - ro["ndcg@100"])))))
    assert nd_c > 0.15, "the CPU port must have learned the groups' rankings in %d steps (untrained: 0.05)" % K
    assert curve < 1e-3, curve                       # (achieved on MI355X: 8e-6; nDCG 2e-5, Recall 3e-3 relative)
    assert abs(nd_h - nd_c) < 1e-2 * nd_c and abs(rc_h - rc_c) < 1e-2 * rc_c


def test_dp_native_plan_issues_the_documented_collectives():
    """The engine-scheduled data-parallel step (rtx_engine_train_step_dp) as rank 3 of 8 with RECORDING collectives (caller-supplied
    rtx_dp_ops that move nothing): which collectives it issues, on which buffers, of which sizes, in which order -- DESIGN 6.1's
    plan -- and that the sharded optimizer touches exactly this rank's rows of the two big matrices (everything else of the
    network is replicated and updated in full)."""
    import ctypes as C
    from rectorch_amd import _lib
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B, G, RANK = 20108, 600, 200, 64, 8, 3
    X = synth_interactions(B, I, seed=3)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5, bias_std=0.05)
    calls = []

    def rec(name):
        def f(_ctx, buf, n, *rest):
            calls.append((name, int(buf), int(n), int(rest[-1] or 0)))       # (collective, buffer, count, stream)
            return 0
        return f

    def grp(name):
        def f(_ctx):
            calls.append((name, 0, 0, 0))
            return 0
        return f

    for sharded in (True, False):
        net, model = make_vae([I, H, L], [L, H, I], 0.5, sd, beta=0.2, numerics="bf16")
        st, params, m, v = model._ensure_train_state()
        eng = net.rtx_engine("bf16", B, train_buffers=(st.grads, m, v))
        ops = _lib.DpOps()
        keep = (_lib.DP_REDUCE_FN(rec("all_reduce")), _lib.DP_REDUCE_FN(rec("reduce_scatter")), _lib.DP_GATHER_FN(rec("all_gather")),
                _lib.DP_GROUP_FN(grp("group_start")), _lib.DP_GROUP_FN(grp("group_end")))
        ops.all_reduce, ops.reduce_scatter, ops.all_gather, ops.group_start, ops.group_end = keep
        cfg = _lib.DpCfg()
        cfg.rank, cfg.world, cfg.sharded, cfg.comm_dtype, cfg.emulate, cfg.comm, cfg.ops = RANK, G, int(sharded), _lib.RTX_BF16, 0, None, C.pointer(ops)
        _lib.check(_lib.lib().rtx_engine_dp_attach(eng.handle, C.byref(cfg)))
        (rb,) = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
        before = [p.detach().clone() for p in params]
        step = eng._step(seed=1, beta=0.1, lam=0.0, inv_batch=1.0 / (B * G), lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.0, step=1)
        del calls[:]
        eng.train_step_dp(rb, None, step, st.loss_buf[0:1], st.loss_buf[1:2])
        torch.cuda.synchronize()
        names = [c[0] for c in calls]
        prow_out, prow_in = (I + 1 + 127) // 128 * 128, (H + 1 + 127) // 128 * 128          # 20224, 640
        if sharded:
            # bucket A (decoder matrix, side stream): RS of its padded region + AR of its bias in ONE group, then the all-gather of
            # its compute copy; bucket B (caller's stream): AR of the small layers + b0 as one range, RS of the encoder matrix, AG
            assert names == ["group_start", "reduce_scatter", "all_reduce", "group_end", "all_gather",
                             "group_start", "all_reduce", "reduce_scatter", "group_end", "all_gather"], names
            rsA, arA, agA, arB, rsB, agB = calls[1], calls[2], calls[4], calls[6], calls[7], calls[9]
            assert rsA[2] == prow_out * H and rsA[2] % G == 0 and arA[2] == I
            assert agA[2] == prow_out * 640 * 2                                              # bf16 compute copy [P(I)][P(600)]
            assert rsB[2] == prow_in * I and rsB[2] % G == 0
            assert arB[2] >= 400 * 600 + 400 + 600 * 200 + 600 + 600 and arB[2] < 400 * 600 + 600 * 200 + 4 * 1024   # W2,b2,W3,b3(dec0),b0 + alignment
            assert agB[2] == prow_in * ((I + 1 + 127) // 128 * 128) * 2
            assert rsA[3] == arA[3] == agA[3] != rsB[3] == arB[3] == agB[3]                   # bucket A on the side stream, B on the caller's
            assert arA[1] == rsA[1] + 2 * (prow_out * H + (-(prow_out * H)) % 64)             # the bias follows the padded region (64-element alignment)
        else:
            assert names == ["group_start", "all_reduce", "group_end", "group_start", "all_reduce", "group_end"], names
            assert calls[1][2] >= I * H + I and calls[4][2] >= H * I + 400 * 600 + 600 * 200
            assert calls[1][3] != calls[4][3]
        lo0, hi0, sh0 = eng.dp_owned_rows(0)
        lo3, hi3, sh3 = eng.dp_owned_rows(3)
        if sharded:
            assert (lo3, hi3, sh3) == (RANK * prow_out // G, (RANK + 1) * prow_out // G, True)
            assert (lo0, hi0, sh0) == (RANK * prow_in // G, (RANK + 1) * prow_in // G, True)
            assert eng.dp_owned_rows(1) == (0, 400, False)
        else:
            assert (lo3, hi3, sh3) == (0, I, False) and (lo0, hi0, sh0) == (0, H, False)
        for t, (p, b) in enumerate(zip(params, before)):
            changed = (p.detach() != b)
            if sharded and t in (0, 6):
                lo, hi = (lo0, hi0) if t == 0 else (lo3, hi3)
                assert bool(changed[lo:hi].any()) and not bool(changed[:lo].any()) and not bool(changed[hi:].any()), t
            else:
                assert bool(changed.any()), t
        _lib.check(_lib.lib().rtx_engine_dp_attach(eng.handle, None))
        del keep
