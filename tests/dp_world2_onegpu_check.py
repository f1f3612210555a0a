"""Run by test_gpu_parity.py::test_dp_world2_on_one_gpu under torch.distributed.run with TWO ranks sharing the one MI355X
of the test box (gloo carries the device tensors; RCCL refuses two ranks on one device).  Each rank feeds ITS rows of the
golden G2 batches (with its rows of the reference's dropout masks and noise); after three data-parallel steps every
rank must hold the reference's parameters -- the sum of the two partial gradients equals the full-batch gradient."""
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


def main():
    rank, world, _ = parallel.init_from_env(backend="gloo")
    assert world == 2
    cases = (("g2b_mvae_train_step_te", torch.float32, True, 5e-6, False),
                                                  ("g2c_mvae_train_step_deep", torch.float32, False, 5e-6, False),
                                                  ("g2c_mvae_train_step_deep", torch.bfloat16, True, 7e-5, False),
                                                  # sharded optimizer: reduce-scatter, Adam on the local rows, all-gather of
                                                  # the compute copies; the masters are gathered before they are compared
                                                  ("g2b_mvae_train_step_te", torch.float32, True, 5e-6, True),
                                                  ("g2c_mvae_train_step_deep", torch.float32, True, 5e-6, True),
                                                  ("g2c_mvae_train_step_deep", torch.bfloat16, True, 7e-5, True))
    # both schedulers: "native" = the engine runs the step and calls back for the collectives (torch.distributed on its streams),
    # "python" = round 2's host-driven reducer
    for engine, (name, comm, bucket_adam, tol, sharded) in [(e, c) for e in ("native", "python") for c in cases]:
        g = load_golden(name)
        enc, dec = [int(v) for v in g["enc_dims"]], [int(v) for v in g["dec_dims"]]
        beta, anneal, p, lr = [float(v) for v in g["meta"]]
        net = MultiVAE_net(dec, enc, dropout=p)
        sd = sd_from(g, "sd0__")
        if rank == 1:                                   # attach() must overwrite rank 1's parameters with rank 0's
            sd = {k: v + 1.0 for k, v in sd.items()}
        net.load_state_dict({k: torch.from_numpy(np.array(v)) for k, v in sd.items()})
        net.to("cuda")
        model = MultiVAE(net, beta=beta, anneal_steps=int(anneal), learning_rate=lr, numerics="fp32")
        # shard_min_elems = 1: the golden networks are tiny (64 x 16, 77 x 21 ...); with the engine's default threshold (2^20
        # elements) the "sharded" plan would shard NOTHING and reduce-scatter / row-range Adam / all-gather would never see data
        red = parallel.attach(model, min_bucket_bytes=256, comm_dtype=comm, bucket_adam=bucket_adam, sharded=sharded, engine=engine,
                              shard_min_elems=1)
        if engine == "native":
            assert red.native and red.transport == "torch" and red.sharded == sharded
        else:
            assert bool(red.shard_layers) == sharded
        _, keys = params_in_order(sd_from(g, "sd0__"))
        for t in range(g["xs"].shape[0]):
            B = g["xs"][t].shape[0]
            s, e = parallel.shard_rows(B, rank, world)
            model._rtx.inject = (dev(g["mask_%d" % t][s:e], torch.uint8), dev(g["eps_%d" % t][s:e]))
            gt = torch.from_numpy(g["gts"][t][s:e]) if "gts" in g else None
            loss = model.train_batch(torch.from_numpy(g["xs"][t][s:e]), gt)
            ref = float(g["loss_%d" % t])
            assert abs(loss - ref) < (1e-5 if comm == torch.float32 else 1e-5) * abs(ref), (engine, name, t, loss, ref)
            sd_t, _ = params_in_order(sd_from(g, "sd_%d__" % t))
            if sharded:
                assert model._rtx.masters_sharded
                if engine == "native":              # the ENGINE really shards: first and last layer, this rank's half of the rows
                    eng = net._rtx_engines["fp32"]
                    n_l = eng.n_tensors // 2
                    owned = [eng.dp_owned_rows(l) for l in range(n_l)]
                    assert owned[0][2] and owned[n_l - 1][2], owned
                    for l in (0, n_l - 1):          # this rank's block of the rows PADDED to a multiple of 128 (the golden matrices have
                        rows = net._param_list()[2 * l].shape[0]       # 16 .. 77 rows: rank 1's block can be all padding)
                        per = (rows + 1 + 127) // 128 * 128 // world
                        assert owned[l][:2] == (min(rank * per, rows), min((rank + 1) * per, rows)), (rank, l, owned)
                    assert eng.get_option("dp_bytes_reduce_scatter") > 0 and eng.get_option("dp_bytes_all_gather") > 0
                model._gather_sharded_state()       # collective: every rank completes its float32 rows
            for k, prm, want in zip(keys, net._param_list(), sd_t):
                dl = np.abs(prm.detach().cpu().numpy() - want)
                if comm == torch.float32:
                    assert float(dl.max()) < tol, (engine, name, str(comm), "sharded" if sharded else "replicated", t, k, float(dl.max()))
                else:
                    if os.environ.get("DP_DEBUG") and float(np.mean(dl > tol)) >= 0.03 and rank == 0:
                        p0 = np.array(sd_from(g, "sd0__" if t == 0 else "sd_%d__" % (t - 1))[k])
                        got = prm.detach().cpu().numpy()
                        bad = dl > tol
                        rows = got.shape[0] if got.ndim == 2 else 1
                        print("DEBUG", engine, name, "sharded" if sharded else "replicated", t, k, got.shape,
                              "bad per row:", bad.reshape(rows, -1).mean(1).round(2).tolist()[:80],
                              "| bad==unchanged:", float(np.mean(np.abs(got - p0)[bad] < 1e-7)),
                              "| bad moved 2x:", float(np.mean(np.abs(np.abs(got - p0)[bad] - 2e-3) < 1e-4)), flush=True)
                    # the two ranks' partial gradients are rounded to bf16 before they are summed: where they nearly
                    # cancel, the sign of the sum -- and with it Adam's +-lr move -- can flip on a few elements
                    assert float(dl.max()) <= (t + 1) * 2.1e-3 and float(np.mean(dl > tol)) < 0.03, \
                        (engine, name, str(comm), "sharded" if sharded else "replicated", t, k, float(dl.max()), float(np.mean(dl > tol)))
    dist.barrier()
    if rank == 0:
        print("DP_WORLD2_OK")
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
