"""Stress test of the step's in-kernel synchronisation: many training steps at the flagship shape with EVERY
scheduling shortcut on -- next batch prefetched on the side stream, join deferred and folded into the first-layer product, fork
folded into the data-gradient product, second stream -- while a SECOND PROCESS keeps the same GPU busy with foreign kernels and
copies (uneven load: the timing the folded hops were tuned under does not hold).  Every `every` steps an integer checksum of all
parameters is taken on the device; the sequence must equal, bit for bit, that of the same run on the plain schedule (events, no
prefetch, no deferred join, no folds).  A lost acquire / release, a join dropped or taken too early or a batch image read before
its gather finished changes the bits of some later checksum.

    python tests/stress_sync_check.py [steps every I B]        (the pytest item runs 20000 / 1000 at I = 20108, B = 500)
"""
import os
import subprocess
import sys
import tempfile
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

HAMMER = r"""
import os, sys, time, torch
flag = sys.argv[1]
torch.cuda.set_device(0)
a = torch.randn(3072, 3072, device="cuda", dtype=torch.bfloat16)
big = torch.empty(192 << 20, device="cuda", dtype=torch.uint8)
dst = torch.empty_like(big)
side = torch.cuda.Stream()
n = 0
t0 = time.time()
while not os.path.exists(flag) and time.time() - t0 < 600:
    # uneven on purpose: bursts of matrix products, a long copy on another stream, short idle gaps
    for _ in range(1 + n % 7):
        a = (a @ a).clamp_(-1, 1)
    with torch.cuda.stream(side):
        dst.copy_(big)
    if n % 5 == 0:
        torch.cuda.synchronize()
        time.sleep(0.0005 * (n % 3))
    n += 1
torch.cuda.synchronize()
print("hammer: %d rounds" % n)
"""


def run(steps, every, I, B, fancy, X, sd):
    from rectorch_amd.models import MultiVAE
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.samplers import DataSampler
    net = MultiVAE_net([200, 600, I], dropout=0.5)
    net.load_state_dict({k: torch.from_numpy(v.copy()) for k, v in sd.items()})
    net.to("cuda")
    model = MultiVAE(net, beta=0.2, anneal_steps=0, learning_rate=1e-3, numerics="bf16")
    st_, _, m_, v_ = model._ensure_train_state()
    eng = net.rtx_engine("bf16", B, train_buffers=(st_.grads, m_, v_))
    if not fancy:
        for k in ("hop_fold", "hop_values", "hop_kernels"):
            eng.set_option(k, 0)          # cross-stream dependencies as plain events
        model.prefetch_batches = False
    torch.manual_seed(5)
    batches = list(DataSampler(X, batch_size=B, shuffle=False).iter_rows())
    batches = [b for b in batches if len(b) == B]
    sums = []
    t0 = time.time()
    for t in range(steps):
        nxt = batches[(t + 1) % len(batches)] if fancy else None
        model._fused_step(batches[t % len(batches)], None, want_loss=False, next_x=nxt, defer_join=fancy)
        if (t + 1) % every == 0:
            model._join()
            acc = torch.zeros((), dtype=torch.int64, device="cuda")
            for p in net._param_list():
                acc += p.detach().view(torch.int32).sum(dtype=torch.int64)
            sums.append(int(acc.item()))
    model._join()
    torch.cuda.synchronize()
    info = {"us_per_step": (time.time() - t0) / steps * 1e6, "prefetch_hits": eng.get_option("prefetch_hits"),
            "join_folds": eng.get_option("join_folds") if fancy else 0}
    return sums, info


def main():
    a = [int(x) for x in sys.argv[1:]]
    steps, every, I, B = (a + [20000, 1000, 20108, 500][len(a):])[:4]
    from rectorch_amd.utils import hash_state_dict, synth_interactions
    X = synth_interactions(8 * B, I, mu=3.9, sigma=0.9, dmax=I // 4, seed=17)
    sd = hash_state_dict([I, 600, 200], [200, 600, I], "vae", 9)
    flag = os.path.join(tempfile.mkdtemp(), "stop")
    hammer = subprocess.Popen([sys.executable, "-c", HAMMER, flag], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    try:
        time.sleep(3.0)                                   # the foreign process is up and loading the device
        assert hammer.poll() is None, "the foreign-load process died: " + (hammer.stdout.read() or "")
        fancy, fi = run(steps, every, I, B, True, X, sd)
    finally:
        open(flag, "w").close()
        out = hammer.communicate(timeout=120)[0]
    plain, pi = run(steps, every, I, B, False, X, sd)     # the reference schedule, device to itself
    print("foreign load:", out.strip().splitlines()[-1] if out.strip() else "(no output)")
    print("fancy schedule under foreign load: %.1f us/step, %s" % (fi["us_per_step"], fi))
    print("plain schedule: %.1f us/step" % pi["us_per_step"])
    bad = [i for i, (x, y) in enumerate(zip(fancy, plain)) if x != y]
    print("checksums: %d, first %s, last %s" % (len(fancy), fancy[:1], fancy[-1:]))
    assert len(fancy) == len(plain) == steps // every
    assert not bad, "checksum %d (step %d) differs: %d vs %d" % (bad[0], (bad[0] + 1) * every, fancy[bad[0]], plain[bad[0]])
    assert fi["prefetch_hits"] >= steps - steps // every - 2, fi      # (a step right after a checksum gathers for itself: no announcement was skipped)
    print("STRESS OK: %d steps, %d checksums identical" % (steps, len(fancy)))


if __name__ == "__main__":
    main()
