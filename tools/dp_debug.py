"""diagnostic for tests/dp_local_threads_check.py: per-step, per-tensor differences between stream-ordered and drained runs"""
import os
import sys
import threading
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import dp_local_threads_check as C      # noqa: E402
from rectorch_amd import parallel       # noqa: E402
from rectorch_amd.samplers import DataSampler   # noqa: E402
from rectorch_amd.utils import hash_state_dict, synth_interactions  # noqa: E402


def run(world, sharded, drain, two_comms, sd, batches, masks, noise, opts=(), sync_each=False):
    grp = parallel.LocalGroup(world, drain=drain)
    out, errs = [None] * world, [None] * world

    def rank_main(r):
        try:
            torch.cuda.set_device(0)
            with torch.cuda.stream(torch.cuda.Stream()):
                net, model = C.build(sd)
                plan = parallel.attach(model, group=grp, transport="local", local_rank=r, sharded=sharded, two_comms=two_comms)
                st_, _, m_, v_ = model._ensure_train_state()
                eng0 = net.rtx_engine("bf16", C.B_GLOBAL // world, train_buffers=(st_.grads, m_, v_))
                for kv in opts:
                    k, v = kv.split("=")
                    eng0.set_option(k, int(v))
                snaps = []
                for t in range(C.STEPS):
                    rb = parallel.shard_batch(batches[t], r, world)
                    s, e = parallel.shard_rows(C.B_GLOBAL, r, world)
                    model._rtx.inject = (masks[t][s:e].cuda(), noise[t][s:e].cuda())
                    loss = model._fused_step(rb, None, want_loss=(sync_each or t == C.STEPS - 1))
                    if sync_each or t == C.STEPS - 1:
                        torch.cuda.synchronize()
                        if sharded:
                            model.consolidate()
                        snaps.append(([p.detach().cpu().numpy().copy() for p in net._param_list()], loss))
                out[r] = snaps
                plan.close()
        except BaseException as ex:
            errs[r] = ex
            try:
                grp.barrier.abort()
            except Exception:
                pass
    th = [threading.Thread(target=rank_main, args=(r,), daemon=True) for r in range(world)]
    [t.start() for t in th]
    [t.join(timeout=240) for t in th]
    for e in errs:
        if e is not None and not isinstance(e, threading.BrokenBarrierError):
            raise e
    return out


def diff(a, b, tag):
    for r in range(len(a)):
        for t in (-1,):
            d = [float(np.abs(x - y).max()) for x, y in zip(a[r][t][0], b[r][t][0])]
            nd = [int((x.view(np.int32) != y.view(np.int32)).sum()) for x, y in zip(a[r][t][0], b[r][t][0])]
            print("%s rank %d step %d loss %.6f vs %.6f  max|d| %s  n_diff %s" % (tag, r, t, a[r][t][1], b[r][t][1], ["%.1e" % x for x in d], nd), flush=True)


def main():
    X = synth_interactions(C.STEPS * C.B_GLOBAL, C.I, mu=3.5, sigma=0.9, dmax=C.I // 2, seed=5)
    sd = hash_state_dict([C.I, C.H, C.L], [C.L, C.H, C.I], "vae", 31, bias_std=0.05)
    gen = torch.Generator().manual_seed(77)
    masks = [(torch.rand(C.B_GLOBAL, C.I, generator=gen) >= 0.5).to(torch.uint8) for _ in range(C.STEPS)]
    noise = [torch.randn(C.B_GLOBAL, C.L, generator=gen) for _ in range(C.STEPS)]
    batches = list(DataSampler(X, batch_size=C.B_GLOBAL, shuffle=False).iter_rows())
    for sharded in (False, True):
        ref = run(2, sharded, True, True, sd, batches, masks, noise)
        for rep in range(5):
            diff(run(2, sharded, False, True, sd, batches, masks, noise), ref, "sharded=%s ordered rep %d" % (sharded, rep))
        for rep in range(3):
            diff(run(2, sharded, False, False, sd, batches, masks, noise), ref, "sharded=%s one-table rep %d" % (sharded, rep))
        for rep in range(3):
            diff(run(2, sharded, False, True, sd, batches, masks, noise, opts=("two_stream=0",)), ref, "sharded=%s two_stream=0 rep %d" % (sharded, rep))
        for rep in range(3):
            diff(run(2, sharded, False, True, sd, batches, masks, noise, opts=("hop_values=0",)), ref, "sharded=%s hop_values=0 rep %d" % (sharded, rep))
        for rep in range(2):
            diff(run(2, sharded, False, True, sd, batches, masks, noise, sync_each=True), ref, "sharded=%s sync-each rep %d" % (sharded, rep))
        for rep in range(2):
            diff(run(2, sharded, True, True, sd, batches, masks, noise), ref, "sharded=%s drained rep %d" % (sharded, rep))


if __name__ == "__main__":
    main()
