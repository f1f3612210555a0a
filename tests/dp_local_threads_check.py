"""Run by test_gpu_parity.py::test_dp_stream_ordered_ranks_on_one_gpu: the data-parallel step at world 2 and 4 with collectives that
are STREAM-ORDERED device work and nothing else -- what RCCL kernels are -- on the one MI355X of the test box.

Why: every other multi-rank test moves the gradients through gloo, which needs the DEVICE drained before and after each
collective (rectorch_amd/parallel.py, NativePlan._torch_ops); a drained device cannot show an ordering bug between the engine's
kernels and a collective, or between bucket A (the decoder matrix: exchange + Adam + all-gather on the engine's SIDE stream
beside the data-gradient chain) and bucket B (everything else, on the caller's stream).  Here the ranks are threads of this
process, each with an engine, a caller's stream and a side stream of its own (parallel.LocalGroup): a collective copies the
rank's buffer to a staging slot on the stream the engine names, the threads meet on a HOST barrier (the device keeps running),
and the stream waits for the peers' "staged" events and combines the slots.  No hipDeviceSynchronize, no stream synchronise
inside the step.

Checked, for {sharded, replicated} x {world 2, world 4}, bf16 numerics (the two-stream schedule), MultiVAE [3000, 600, 200]:
  * the run WITHOUT any drain equals, bit for bit, the same run with the device drained around every collective
    (LocalGroup(drain=True), the gloo discipline) -- parameters of every rank after 4 steps, and the last loss;
  * every rank holds the same parameters (bit for bit);
  * bucket A's collectives went through a table of their own (engine option dp_two_comms), and the single-table schedule
    (two_comms=False) gives the same bits;
  * the result is within the stated bf16-exchange bound of what ONE GPU computes on the whole batch.
"""
import os
import sys
import threading

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from rectorch_amd import parallel                                   # noqa: E402
from rectorch_amd.models import MultiVAE                            # noqa: E402
from rectorch_amd.nets import MultiVAE_net                          # noqa: E402
from rectorch_amd.samplers import DataSampler                       # noqa: E402
from rectorch_amd.utils import hash_state_dict, synth_interactions  # noqa: E402

I, H, L, B_GLOBAL, STEPS, LR = 3000, 600, 200, 256, 4, 1e-3


def build(sd):
    net = MultiVAE_net([L, H, I], dropout=0.5)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to("cuda")
    return net, MultiVAE(net, beta=0.2, anneal_steps=0, learning_rate=LR, numerics="bf16")


def run_world(world, sharded, drain, two_comms, sd, batches, masks, noise):
    grp = parallel.LocalGroup(world, drain=drain)
    out, errs = [None] * world, [None] * world

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):            # every rank: a caller's stream of its own
                net, model = build(sd)
                plan = parallel.attach(model, group=grp, transport="local", local_rank=r, sharded=sharded, two_comms=two_comms)
                assert plan.native and plan.transport == "local"
                loss = None
                for t in range(STEPS):
                    rb = parallel.shard_batch(batches[t], r, world)
                    s, e = parallel.shard_rows(B_GLOBAL, r, world)
                    model._rtx.inject = (masks[t][s:e].cuda(), noise[t][s:e].cuda())
                    loss = model._fused_step(rb, None, want_loss=(t == STEPS - 1))
                eng = net._rtx_engines["bf16"]
                info = {"two_comms": eng.get_option("dp_two_comms"), "two_stream": eng.get_option("side_concurrent"),
                        "owned": [eng.dp_owned_rows(l)[2] for l in range(4)]}
                if sharded:
                    model.consolidate()
                torch.cuda.synchronize()
                out[r] = ([p.detach().cpu().numpy().copy() for p in net._param_list()], loss, info)
                plan.close()
        except BaseException as ex:                                  # noqa: BLE001 (reported by the main thread)
            errs[r] = ex
            try:
                grp.barrier.abort()
            except Exception:
                pass

    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    for t in th:
        t.start()
    for t in th:
        t.join(timeout=240)
    assert not any(t.is_alive() for t in th), "a rank thread hangs (world %d, sharded %s, drain %s)" % (world, sharded, drain)
    first = next((e for e in errs if e is not None and not isinstance(e, threading.BrokenBarrierError)), None) or next((e for e in errs if e), None)
    if first is not None:
        raise first
    return out, grp.calls


def main():
    assert torch.cuda.is_available()
    X = synth_interactions(STEPS * B_GLOBAL, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=5)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 31, bias_std=0.05)
    gen = torch.Generator().manual_seed(77)
    masks = [(torch.rand(B_GLOBAL, I, generator=gen) >= 0.5).to(torch.uint8) for _ in range(STEPS)]
    noise = [torch.randn(B_GLOBAL, L, generator=gen) for _ in range(STEPS)]
    batches = list(DataSampler(X, batch_size=B_GLOBAL, shuffle=False).iter_rows())
    # what one GPU computes on the whole batch
    net1, one = build(sd)
    for t in range(STEPS):
        one._rtx.inject = (masks[t].cuda(), noise[t].cuda())
        loss1 = one._fused_step(batches[t], None, want_loss=True)
    want = [p.detach().cpu().numpy().copy() for p in net1._param_list()]
    init = [p for p in (sd[k] for k in ("enc_layers.0.weight", "enc_layers.0.bias", "enc_layers.1.weight", "enc_layers.1.bias",
                                        "dec_layers.0.weight", "dec_layers.0.bias", "dec_layers.1.weight", "dec_layers.1.bias"))]
    for world in (2, 4):
        for sharded in (True, False):
            ref, _ = run_world(world, sharded, True, True, sd, batches, masks, noise)        # drained around every collective
            for rep in range(3):                                                              # a race does not show on every run
                got, calls = run_world(world, sharded, False, True, sd, batches, masks, noise)
                assert calls > 0
                for r in range(world):
                    info = got[r][2]
                    assert info["two_stream"] == 1, "the step ran on ONE stream: nothing to order (%r)" % (info,)
                    assert info["two_comms"] == 1, info
                    assert info["owned"] == ([True, False, False, True] if sharded else [False] * 4), info
                    assert got[r][1] == ref[r][1], ("loss", world, sharded, rep, r, got[r][1], ref[r][1])
                    for k, (a, b, c) in enumerate(zip(got[r][0], ref[r][0], got[0][0])):
                        assert np.array_equal(a.view(np.int32), b.view(np.int32)), \
                            ("stream-ordered run differs from the drained run", world, sharded, rep, r, k, float(np.abs(a - b).max()))
                        assert np.array_equal(a.view(np.int32), c.view(np.int32)), ("replicas differ", world, sharded, rep, r, k)
            one_tab, _ = run_world(world, sharded, False, False, sd, batches, masks, noise)  # bucket A shares bucket B's table
            assert one_tab[0][2]["two_comms"] == 0
            for k, (a, b) in enumerate(zip(one_tab[0][0], ref[0][0])):
                assert np.array_equal(a.view(np.int32), b.view(np.int32)), ("single-table schedule differs", world, sharded, k)
            assert abs(ref[0][1] - loss1) < 2e-3 * abs(loss1), (world, sharded, ref[0][1], loss1)
            for k, (g, w, p0) in enumerate(zip(ref[0][0], want, init)):
                d = np.abs(g - w)
                moved = float(np.mean(np.abs(w - p0)))
                assert float(d.max()) <= STEPS * 2.1 * LR, (world, sharded, k, float(d.max()))
                assert float(d.mean()) < 0.02 * max(moved, 1e-6), (world, sharded, k, float(d.mean()), moved)
            print("world %d %s: 3 stream-ordered runs == drained run (bits), replicas identical, single-table schedule identical"
                  % (world, "sharded   " if sharded else "replicated"), flush=True)
    print("DP_LOCAL_THREADS_OK")


if __name__ == "__main__":
    main()
