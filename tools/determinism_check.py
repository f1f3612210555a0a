"""Repeat the same seeded training run R times under several engine option sets and count the distinct final parameter checksums
(a race shows as more than one).  usage: determinism_check.py [I H L B steps reps]"""
import os
import sys
import hashlib
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from rectorch_amd.models import MultiVAE            # noqa: E402
from rectorch_amd.nets import MultiVAE_net          # noqa: E402
from rectorch_amd.samplers import DataSampler       # noqa: E402
from rectorch_amd.utils import hash_state_dict, synth_interactions  # noqa: E402


def one(I, H, L, B, steps, opts, announce, X, sd):
    net = MultiVAE_net([L, H, I], dropout=0.5)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to("cuda")
    model = MultiVAE(net, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics="bf16")
    st_, _, m_, v_ = model._ensure_train_state()
    eng = net.rtx_engine("bf16", B, train_buffers=(st_.grads, m_, v_))
    for kv in opts:
        k, v = kv.split("=")
        eng.set_option(k, int(v))
    torch.manual_seed(5)
    batches = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
    for t in range(steps):
        nxt = batches[(t + 1) % len(batches)] if announce else None
        model._fused_step(batches[t % len(batches)], None, want_loss=False, next_x=nxt)
    torch.cuda.synchronize()
    h = hashlib.sha256()
    for p in net._param_list():
        h.update(p.detach().cpu().numpy().tobytes())
    return h.hexdigest()[:12], eng.get_option("prefetch_hits")


def main():
    a = [int(x) for x in sys.argv[1:]]
    I, H, L, B, steps, reps = (a + [3000, 600, 200, 192, 12, 8][len(a):])[:6]
    X = synth_interactions(6 * B + 77, I, mu=3.5, sigma=0.9, dmax=I // 2, seed=13)
    sd = hash_state_dict([I, H, L], [L, H, I], "vae", 5)
    for name, opts, announce in (("default, no announcements", (), False), ("default, announcements", (), True),
                                 ("two_stream=0", ("two_stream=0",), False), ("hop_values=0 no announcements", ("hop_values=0",), False),
                                 ("hop_values=0 announcements", ("hop_values=0",), True),
                                 ("fuse_adam=0", ("fuse_adam=0",), False), ("gather_scatter=0", ("gather_scatter=0",), False),
                                 ("small_fwd=0 small_bwd=0", ("small_fwd=0", "small_bwd=0"), False)):
        got = [one(I, H, L, B, steps, opts, announce, X, sd) for _ in range(reps)]
        kinds = {}
        for g, _ in got:
            kinds[g] = kinds.get(g, 0) + 1
        print("%-34s I=%d B=%d steps=%d: %d distinct of %d runs %s hits %s" % (name, I, B, steps, len(kinds), reps, kinds, got[0][1]), flush=True)


if __name__ == "__main__":
    main()
