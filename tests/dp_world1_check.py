"""Run by test_gpu_parity.py::test_dp_path_world1_rccl in its own process: the data-parallel step with a world of ONE RCCL
rank must reproduce the single-GPU step: exactly the golden G2 trajectory with the float32 exchange, and within bf16 rounding of
the gradients with the bf16 exchange.  Both schedulers: "native" (the engine issues RCCL itself through its own communicator:
rtx_engine_train_step_dp, one C call per step) and "python" (round 2's host-driven reducer: per-layer callback -> collective on
a side stream -> per-bucket Adam on a third stream); in bf16 numerics the two must agree bit for bit."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import load_golden, sd_from, params_in_order          # noqa: E402
from rectorch_amd import parallel                                   # noqa: E402
from rectorch_amd.models import MultiVAE                            # noqa: E402
from rectorch_amd.nets import MultiVAE_net                          # noqa: E402


def dev(a, dtype=torch.float32):
    return torch.as_tensor(np.ascontiguousarray(a)).to("cuda", dtype)


def run(g, comm_dtype, bucket_adam, min_bucket, sharded=False, numerics="fp32", direct16=True, engine="python"):
    enc, dec = [int(v) for v in g["enc_dims"]], [int(v) for v in g["dec_dims"]]
    beta, anneal, p, lr = [float(v) for v in g["meta"]]
    net = MultiVAE_net(dec, enc, dropout=p)
    net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd_from(g, "sd0__").items()})
    net.to("cuda")
    model = MultiVAE(net, beta=beta, anneal_steps=int(anneal), learning_rate=lr, numerics=numerics)
    if comm_dtype is not None:
        red = parallel.attach(model, min_bucket_bytes=min_bucket, comm_dtype=comm_dtype, bucket_adam=bucket_adam, sharded=sharded,
                              engine=engine)
        if engine == "python":
            red.allow_direct16 = direct16
            assert len(red.buckets()) >= 2 and bool(red.shard_layers) == sharded
        else:
            assert red.native and red.transport == "rccl" and red.sharded == sharded
            # the start-up self-check of the communicators ran (one all-reduce + reduce-scatter + all-gather of the real bucket sizes on
            # each communicator, both busy at once, against host-computed sums) -- the only part of a first multi-rank run that a
            # one-GPU box can exercise: same calls, same streams, world 1
            rep = red.self_check_report
            assert rep["checked"] and rep["world"] == 1 and len(rep["comms"]) == (2 if red.comm_side is not None else 1), rep
            assert all(c["elements"] >= 64 for c in rep["comms"]), rep
    losses = []
    for t in range(g["xs"].shape[0]):
        model._rtx.inject = (dev(g["mask_%d" % t], torch.uint8), dev(g["eps_%d" % t]))
        gt = torch.from_numpy(g["gts"][t]) if "gts" in g else None
        losses.append(model.train_batch(torch.from_numpy(g["xs"][t]), gt))
    model._gather_sharded_state()
    torch.cuda.synchronize()
    return losses, [p_.detach().cpu().numpy().copy() for p_ in net._param_list()]


def main():
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[1], rank=0, world_size=1)
    g = load_golden("g2c_mvae_train_step_deep")
    _, keys = params_in_order(sd_from(g, "sd0__"))
    ref_losses, ref_params = run(g, None, False, 0)
    for bucket_adam in (False, True):
        losses, params = run(g, torch.float32, bucket_adam, 256)
        assert losses == ref_losses, (bucket_adam, losses, ref_losses)
        for k, a, b in zip(keys, params, ref_params):
            assert np.array_equal(a, b), ("fp32 exchange must be bit-identical to the single-GPU step", bucket_adam, k)
        sd_t, _ = params_in_order(sd_from(g, "sd_%d__" % (g["xs"].shape[0] - 1)))
        for k, a, b in zip(keys, params, sd_t):
            assert float(np.max(np.abs(a - b))) < 5e-6, (bucket_adam, k)
    # sharded optimizer with one rank: in-place RCCL reduce-scatter / all-gather over the whole matrices, same arithmetic
    losses, params = run(g, torch.float32, True, 256, sharded=True)
    assert losses == ref_losses, (losses, ref_losses)
    for k, a, b in zip(keys, params, ref_params):
        assert np.array_equal(a, b), ("sharded fp32 exchange must be bit-identical to the single-GPU step", k)
    for bucket_adam in (False, True):
        losses, params = run(g, torch.bfloat16, bucket_adam, 256)
        for k, a, b in zip(keys, params, ref_params):
            # Adam normalises the step: a gradient rounded to 8 bits moves a parameter by at most ~lr per step
            assert float(np.max(np.abs(a - b))) < 3 * 1e-3 * 0.02 + 1e-6, (bucket_adam, k, float(np.max(np.abs(a - b))))
    # bf16 numerics (what bench.py --gpus N runs): the weight-gradient kernels write the bf16 images of the exchange directly
    # (RTX_STEP_GRADS_BF16) -- bit-identical to storing float32 gradients and casting them, replicated and sharded optimizer;
    # and close to the single-GPU bf16 step (whose Adam sees the unrounded gradient)
    one_losses, one_params = run(g, None, False, 0, numerics="bf16")
    for sharded in (False, True):
        l_cast, p_cast = run(g, torch.bfloat16, True, 256, sharded=sharded, numerics="bf16", direct16=False)
        l_dir, p_dir = run(g, torch.bfloat16, True, 256, sharded=sharded, numerics="bf16", direct16=True)
        assert l_cast == l_dir, (sharded, l_cast, l_dir)
        for k, a, b in zip(keys, p_cast, p_dir):
            assert np.array_equal(a, b), ("direct bf16 gradients must equal cast float32 gradients", sharded, k)
        for k, a, b in zip(keys, p_dir, one_params):
            assert float(np.max(np.abs(a - b))) < 3 * 1e-3 * 0.02 + 1e-6, (sharded, k, float(np.max(np.abs(a - b))))
    # ---- the engine-scheduled step (rtx_engine_train_step_dp over the engine's own RCCL communicator) ------------------------
    for sharded in (False, True):
        losses, params = run(g, torch.float32, True, 256, sharded=sharded, engine="native")
        assert losses == ref_losses, ("native", sharded, losses, ref_losses)
        for k, a, b in zip(keys, params, ref_params):
            assert np.array_equal(a, b), ("native fp32 exchange must be bit-identical to the single-GPU step", sharded, k)
        losses, params = run(g, torch.bfloat16, True, 256, sharded=sharded, engine="native")
        for k, a, b in zip(keys, params, ref_params):
            assert float(np.max(np.abs(a - b))) < 3 * 1e-3 * 0.02 + 1e-6, ("native", sharded, k, float(np.max(np.abs(a - b))))
        # bf16 numerics: the same bf16 gradient images, the same Adam kernel -> the same bits as the host-driven scheduler
        l_nat, p_nat = run(g, torch.bfloat16, True, 256, sharded=sharded, numerics="bf16", engine="native")
        l_py, p_py = run(g, torch.bfloat16, True, 256, sharded=sharded, numerics="bf16", direct16=True)
        assert l_nat == l_py, (sharded, l_nat, l_py)
        for k, a, b in zip(keys, p_nat, p_py):
            assert np.array_equal(a, b), ("native and python schedulers must agree bit for bit", sharded, k)
        l32, p32 = run(g, torch.float32, True, 256, sharded=sharded, numerics="bf16", engine="native")   # float32 exchange of bf16 numerics
        for k, a, b in zip(keys, p32, one_params):
            assert float(np.max(np.abs(a - b))) < 1e-6, ("bf16 numerics, float32 exchange = the fused single-GPU step", sharded, k, float(np.max(np.abs(a - b))))
    # ---- round 6: the data-parallel step with the NEXT batch announced (gathered behind bucket A on the engine's side stream) ends
    #      with the same bits as the step that gathers for itself, and really starts from the prefetched image
    from rectorch_amd.utils import synth_interactions, hash_state_dict
    from rectorch_amd.samplers import DataSampler
    I, H, L, B = 3000, 600, 200, 128
    X = synth_interactions(5 * B, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=21)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 4)
    outs = []
    for announce in (False, True):
        for sharded in (False, True):
            net = MultiVAE_net([L, H, I], dropout=0.5)
            net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
            net.to("cuda")
            model = MultiVAE(net, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics="bf16")
            plan = parallel.attach(model, comm_dtype=torch.bfloat16, sharded=sharded, engine="native", fixed_global_batch=B)
            torch.manual_seed(77)
            rbs = [parallel.shard_batch(rb, 0, 1) for rb in DataSampler(X, batch_size=B, shuffle=False).iter_rows()]
            for t in range(8):
                model._fused_step(rbs[t % len(rbs)], None, want_loss=False, next_x=rbs[(t + 1) % len(rbs)] if announce else None)
            model._gather_sharded_state()
            torch.cuda.synchronize()
            eng = net._rtx_engines["bf16"]
            hits = eng.get_option("prefetch_hits")
            assert (hits >= 7) if announce else (hits == 0), (announce, sharded, hits)
            outs.append([p.detach().cpu().numpy().copy() for p in net._param_list()])
            plan.close()
    for a, b in zip(outs[0] + outs[1], outs[2] + outs[3]):
        assert np.array_equal(a.view(np.int32), b.view(np.int32)), "data-parallel steps from prefetched batches must equal self-gathered ones bit for bit"
    dist.destroy_process_group()
    print("DP_WORLD1_OK")


if __name__ == "__main__":
    main()
