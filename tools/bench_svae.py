#!/usr/bin/env python
"""SVAE training throughput at an ml-1m-like shape (BASELINE.json configs[4]; reference rectorch/models.py:1609-1635):
one user sequence per Adam step, embedding 256 -> GRU 200 -> [150] -> latent 64 -> [150] -> n_items, 'next_k' targets.

    python tools/bench_svae.py [--users 6040] [--items 3416] [--mean-len 165] [--steps 2000] [--cpu-seconds 15]

Prints one JSON line: sequences/s and time steps/s on the MI355X (inputs resident: the sampler's compact targets are
uploaded before the timed region), the per-kernel HIP time if rocprofv3 wraps the run, and the numpy-oracle baseline on
a bounded sample of the same users."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from rectorch_amd.models import SVAE                       # noqa: E402
from rectorch_amd.nets import SVAE_net                     # noqa: E402
from rectorch_amd.samplers import SVAE_Sampler             # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--users", type=int, default=6040)
    ap.add_argument("--items", type=int, default=3416)
    ap.add_argument("--mean-len", type=float, default=165.0)
    ap.add_argument("--max-len", type=int, default=1000)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--warmup", type=int, default=100)
    ap.add_argument("--k", type=int, default=4)
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--pack", type=int, default=1, help="users per optimizer step (SVAE_Sampler(pack=N); 1 = the reference's per-user step)")
    ap.add_argument("--numerics", default="fp32", choices=["fp32", "bf16"], help="bf16: bf16 operands / f32 accumulate in the matrix products")
    a = ap.parse_args()
    rng = np.random.RandomState(1)
    lens = np.clip(rng.lognormal(np.log(a.mean_len) - 0.5, 1.0, size=a.users).astype(int), 5, a.max_len)
    pop = rng.zipf(1.3, size=int(lens.sum()) * 2)
    pop = pop[pop <= a.items][: int(lens.sum())] - 1
    seqs, o = {}, 0
    for u in range(a.users):
        seqs[u] = pop[o:o + lens[u]].tolist()
        o += lens[u]
    torch.manual_seed(0)
    net = SVAE_net(n_items=a.items, embed_size=256, rnn_size=200, dec_dims=[64, 150, a.items], enc_dims=[200, 150, 64])
    sd = {k: v.detach().numpy().copy() for k, v in net.state_dict().items()}
    model = SVAE(net.to("cuda"), beta=0.2, anneal_steps=20000, numerics=a.numerics)
    np.random.seed(0)
    smp = SVAE_Sampler(a.items, seqs, None, pred_type="next_k", k=a.k, shuffle=True, sparse=True, pack=a.pack)
    batches = []
    for i, (x, y) in enumerate(smp):
        batches.append((x if a.pack > 1 else x.to("cuda"), y))
        if i + 1 >= a.steps + a.warmup:
            break
    n = len(batches)
    w = min(a.warmup, n // 2)
    n_users = lambda b: len(b[0]) if a.pack > 1 else 1                       # noqa: E731
    n_steps = lambda b: b[0].n_steps if a.pack > 1 else b[0].numel()         # noqa: E731
    for x, y in batches[:w]:
        model.train_batch(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    steps_t = users_t = longest = 0
    for b in batches[w:]:
        model.train_batch(*b)            # returns loss.item(): one host sync per optimizer step, as in the reference
        steps_t += n_steps(b)
        users_t += n_users(b)
        longest += max(b[0].lens) if a.pack > 1 else n_steps(b)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"workload": "SVAE train, synthetic ml-1m shape: %d items, embed 256, GRU 200, enc [200,150,64], dec [64,150,I], "
                       "next_k k=%d, %s" % (a.items, a.k, "one user per Adam step (the reference's semantics)" if a.pack == 1 else
                                            "packs of up to %d users per Adam step (mean of the users' losses; not in the reference)" % a.pack),
           "pack": a.pack, "users_per_s": users_t / dt, "timesteps_per_s": steps_t / dt, "ms_per_user": dt / users_t * 1e3,
           "ms_per_optimizer_step": dt / (n - w) * 1e3, "mean_len": steps_t / users_t, "users_timed": users_t,
           "optimizer_steps_timed": n - w,
           "dtype": "f32" if a.numerics == "fp32" else "bf16 operands / f32 accumulate in the matrix products, f32 recurrences + Adam"}
    # the recurrences (k_sv_gru_fwd_ks / k_sv_gru_bwd_ks: one persistent workgroup per sequence) keep W_hh resident -- 180 of 250
    # weights per thread in registers, the rest in LDS -- so what a time step moves is the LDS-resident part of the weights
    # (18 x 8 KB forward + 18 x 8 KB backward at R = 200) through one CU's LDS port (128 B/clk at 2.4 GHz = 307 GB/s per sequence in
    # flight).  `achieved` prices those bytes for the LONGEST sequence of every step (the recurrences of a pack run side by side)
    # against the WHOLE step time -- a lower bound for the recurrence kernels themselves, since the step also holds the GEMMs and
    # Adam; the forward kernel's own cycles per time step: tools/svae_stamps.py (profiles/r3_svae_gru_step_cycles.txt).
    lds_bytes = (144 + 144) * 1024
    ach = lds_bytes * longest / dt / 1e9
    out["roofline"] = {"kernel": "k_sv_gru_fwd_ks + k_sv_gru_bwd_ks (weights resident in registers + LDS, one workgroup per sequence)",
                       "bound": "lds_port_of_one_cu", "achieved": ach, "peak": 307.2, "unit": "GB/s per sequence in flight", "frac": ach / 307.2,
                       "traffic": None, "algorithmic_bytes_per_time_step": lds_bytes,
                       "note": "lower bound (whole-step time); the recurrences are latency-bound: two / three barriers and an LDS round trip per step"}
    if a.cpu_seconds > 0:
        from oracle.svae_oracle import SvaeOracle
        orc = SvaeOracle(sd, n_enc=2, n_dec=2, beta=0.2, anneal_steps=20000)
        t0 = time.perf_counter()
        done = ts = 0
        cpu_batches = batches[w:]
        if a.pack > 1:       # the baseline is the reference's algorithm: one user per step
            np.random.seed(0)
            cpu_batches = [(x.to("cuda"), y) for x, y in SVAE_Sampler(a.items, seqs, None, pred_type="next_k", k=a.k, shuffle=True, sparse=True)]
        for x, y in cpu_batches:
            items = x.cpu().numpy().reshape(-1)
            T = len(items)
            yd = np.zeros((T, a.items))
            ptr, idx = y.indptr.cpu().numpy(), y.indices.cpu().numpy()
            for t in range(T):
                yd[t, idx[ptr[t]:ptr[t + 1]]] = 1.0
            orc.train_batch(items, yd, rng.randn(T, 64))
            done += 1
            ts += T
            if time.perf_counter() - t0 > a.cpu_seconds:
                break
        cdt = time.perf_counter() - t0
        out["cpu_baseline"] = {"users_per_s": done / cdt, "timesteps_per_s": ts / cdt, "kind": "port", "cores": os.cpu_count(),
                               "sample": "%d users (numpy float64 restatement, BLAS threads as configured)" % done}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
