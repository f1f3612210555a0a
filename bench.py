#!/usr/bin/env python
"""Headline benchmark: MultiVAE training users/sec on ml-20m-shaped synthetic data (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W
    N > 1, either way:  python bench.py --gpus N ...   -- bench.py starts its N ranks itself (one process per visible GPU, rendezvous
                        on 127.0.0.1 at a free port; rank 0 prints the line; any rank's failure is a non-zero exit with its stderr)
                        python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

A "step" is one MultiVAE.train_batch on one batch of users per GPU: sparse-row gather -> forward -> multinomial + beta-KL
loss -> backward -> (RCCL exchange of the gradients when N > 1) -> Adam, through the same ``_fused_step`` that
``MultiVAE.train_epoch`` drives with a device-resident ``DataSampler``.

Workloads (``--workload``):
  ml20m   (default) BASELINE.json configs[1]: MultiVAE [20108, 600, 200], synthetic CSR 116 677 x 20 108,
          B = 500 users per GPU (weak scaling: global batch 500 * N); ``--scaling strong`` keeps the GLOBAL batch at 500.
  netflix BASELINE.json configs[3]: MultiVAE [17769, 600, 200], synthetic CSR 480 000 x 17 769 (reference shape source
          config/config_data_netflix.json), GLOBAL batch 4096 (strong scaling: 4096 / N users per GPU).
bf16 MFMA operands with f32 accumulation / f32 master weights + Adam, dropout 0.5, beta 0.2 annealed over 100 000 steps,
lr 1e-3, inputs resident in HBM before the timed region.

Timing: an untimed PRE-HEAT (``--preheat-seconds``, default 0.4 s of steps: clocks, caches and the side stream settle -- the
first 18 ms after start are 3-4 % slower than steady state, which is what short windows would otherwise measure), W warm-up
steps, then ``--windows`` (default 3) windows of EXACTLY K steps, each bracketed by a barrier +
``torch.cuda.synchronize()`` on both sides and reduced with MAX over ranks; the line reports the MEDIAN window
(median of >= 3 windows), all windows are listed.

Rank 0 prints ONE JSON line with `roofline` (the dominant kernel -- single GPU: the fused weight-gradient + Adam kernel,
HBM-bound, timed live with HIP events on the stream it runs on), `step_roofline` (the whole step against its
algorithmic bytes), `fp32_parity` (the float32 parity mode's throughput, N = 1) and `cpu_baseline` (the oracle's
torch-CPU restatement of the reference trainer on this box's host cores, bounded sample; N = 1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_TBS = 8.0          # MI355X HBM3E spec (MI355X_MICROARCH.md); ~6.3 TB/s is the measured copy ceiling
MFMA_BF16_PEAK_TF = 2500.0  # dense bf16 MFMA peak


F32_MFMA_PEAK_TF = 157.3    # v_mfma_f32_32x32x2_f32: the f32 vector rate (MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200, help="steps per timed window")
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--windows", type=int, default=3, help="timed windows of --steps steps; the median is reported")
    ap.add_argument("--preheat-seconds", type=float, default=0.4,
                    help="untimed steps for about this long BEFORE the declared warm-up (0 = none): steady-state clocks for short windows")
    ap.add_argument("--workload", default="ml20m", choices=["ml20m", "netflix"])
    ap.add_argument("--scaling", default=None, choices=["weak", "strong"],
                    help="weak: --batch users per GPU; strong: --batch users in the GLOBAL batch (default: weak for ml20m, strong for netflix)")
    ap.add_argument("--batch", type=int, default=None, help="users per GPU (weak) or per global batch (strong); default 500 / 4096")
    ap.add_argument("--numerics", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--users", type=int, default=None)
    ap.add_argument("--items", type=int, default=None)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-fp32-parity", action="store_true")
    ap.add_argument("--kernel-timing-every", type=int, default=16, help="bracket every N-th launch of the dominant kernel with HIP events")
    ap.add_argument("--no-kernel-timing", action="store_true",
                    help="no HIP events around the dominant kernel (roofline.achieved becomes null): measures what the events cost")
    ap.add_argument("--opt", action="append", default=[], metavar="KEY=VALUE",
                    help="engine measurement knob for an A/B run (include/rectorch_hip.h, rtx_engine_set_option)")
    ap.add_argument("--cpu-seconds", type=float, default=15.0)
    ap.add_argument("--cond-dim", type=int, default=0,
                    help="measure the conditioned variant (CMultiVAE, SURVEY 8f-4): this many condition columns are "
                         "appended to every input row, the target stays the item row; not the headline workload")
    ap.add_argument("--force-dp", action="store_true",
                    help="exercise the data-parallel code path (RCCL exchange + split step) even with one rank")
    ap.add_argument("--sharded", action="store_true",
                    help="data parallel: reduce-scatter -> Adam on the local 1/N shard -> all-gather of the weights "
                         "(the default with more than one rank, real or emulated)")
    ap.add_argument("--replicated", action="store_true", help="data parallel: all-reduce + the whole Adam update on every rank")
    ap.add_argument("--dp-engine", default=None, choices=["native", "python"],
                    help="who schedules the data-parallel step: the engine (one C call per step, default) or round 2's Python reducer")
    ap.add_argument("--dp-transport", default=None, choices=["rccl", "torch"],
                    help="engine-scheduled step: RCCL through the engine's own communicator (default) or torch.distributed callbacks")
    ap.add_argument("--emulate-world", type=int, default=0, metavar="G",
                    help="one GPU plays rank 0 of a G-rank job: the engine's data-parallel schedule with every collective replaced by "
                         "device copies of the bytes a rank moves and Adam on 1/G of the rows (HBM cost of the per-GPU step; timing only)")
    ap.add_argument("--pmc-json", default=None,
                    help="JSON written by tools/pmc_bench.sh for THIS build (field hbm_bytes_per_launch): reported as roofline.traffic "
                         "with its file name; without it the newest committed profiles/r*_pmc_summary.json counts while the dominant "
                         "kernel's source is byte-for-byte the profiled one (sha256), else traffic is null")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the side measurements (train_batch API, sparse first layer; N > 1: the replicated all-reduce A/B)")
    ap.add_argument("--replicated-ab", action="store_true",
                    help="N > 1: after the timed region also time the replicated all-reduce schedule (a second set of communicators)")
    ap.add_argument("--no-defer-join", action="store_true",
                    help="every step ends with the caller's stream waiting for the engine's side stream (rounds 1-4)")
    ap.add_argument("--no-prefetch", action="store_true",
                    help="every step gathers its own batch at its head (rounds 1-4) instead of announcing the next one to the engine")
    ap.add_argument("--first-layer", default="dense", choices=["dense", "sparse"],
                    help="dense (default, BASELINE.json's north star): [batch, n_items] x [n_items, hidden] on MFMA (k_gather -> split-K "
                         "rtx_gemm_nt -> k_post); sparse: the VALU product over the stored entries (k_in_chunks -> k_spmm_in)")
    a = ap.parse_args()
    if a.workload == "netflix":
        a.users = a.users or 480000
        a.items = a.items or 17769
        a.scaling = a.scaling or "strong"
        a.batch = a.batch or 4096
    else:
        a.users = a.users or 116677
        a.items = a.items or 20108
        a.scaling = a.scaling or "weak"
        a.batch = a.batch or 500
    if a.replicated:
        a.sharded = False
    elif a.gpus > 1 or a.emulate_world > 1:
        a.sharded = True
    return a


def _cpu_run(X, dims, batch, seconds, threads):
    from oracle.rectorch_cpu import CpuNet, CpuTrainer, densify_batch
    torch.set_num_threads(threads)
    I, H, L = dims
    net = CpuNet([I, H, L], [L, H, I], "vae", 0.5)
    tr = CpuTrainer(net, beta=0.2, anneal_steps=100000, lr=1e-3)
    rng = np.random.default_rng(0)
    perm = rng.permutation(X.shape[0])
    t_all, t_smp, t_start, i = [], [], time.time(), 0
    while True:
        idx = perm[(i * batch) % (len(perm) - batch):][:batch]
        t0 = time.time()
        x = densify_batch(X, list(idx))           # DataSampler.__iter__ work (samplers.py:99-100)
        t1 = time.time()
        tr.train_batch(x)                         # MultiVAE.train_batch (models.py:817-835)
        t2 = time.time()
        if i >= 2:                                # 2 warm-up steps
            t_all.append(t2 - t0)
            t_smp.append(t1 - t0)
        i += 1
        if (time.time() - t_start > seconds and len(t_all) >= 3) or len(t_all) >= 40:
            break
    return float(np.median(t_all)), float(np.median(t_smp)), len(t_all)


def cpu_baseline(X, dims, batch, seconds):
    """the reference trainer's op sequence on this box's host cores (oracle/rectorch_cpu.py), bounded sample:
    all physical cores (the reported baseline) and 8 threads (comparable with the survey container's probe)"""
    import psutil
    cores = psutil.cpu_count(logical=False) or os.cpu_count()
    per, smp, n = _cpu_run(X, dims, batch, seconds, cores)
    per8, smp8, n8 = _cpu_run(X, dims, batch, seconds * 0.6, 8)
    best = min(per, per8)
    return {"value": batch / best, "unit": "users/s", "cores": int(cores if per <= per8 else 8), "kind": "port",
            # `cores` = the threads of the run whose rate is `value` (the faster of the two); the box itself has:
            "cores_physical": int(cores), "cores_logical": int(os.cpu_count() or cores), "threads_tried": [int(cores), 8],
            "users_per_s_all_physical_cores": batch / per, "users_per_s_8_threads": batch / per8,
            "sample": "%d steps (all %d physical cores) + %d steps (8 threads) of B=%d, sampler densify + train_batch, "
                      "2 warm-up, median; torch %s CPU; faster of the two reported" % (n, cores, n8, batch, torch.__version__),
            "all_cores": {"threads": int(cores), "users_per_s": batch / per, "ms_sampler_plus_step": per * 1e3,
                          "ms_sampler": smp * 1e3, "ms_step_only": (per - smp) * 1e3},
            "threads_8": {"threads": 8, "users_per_s": batch / per8, "ms_sampler_plus_step": per8 * 1e3,
                          "ms_sampler": smp8 * 1e3, "ms_step_only": (per8 - smp8) * 1e3}}


def _flush_c_stdio():
    """RCCL writes an informational line through C stdio, which (not a tty) sits in libc's buffer until exit and would land
    AFTER the result line: flush libc early (every rank, once the communicator exists) and again before the JSON."""
    try:
        import ctypes
        ctypes.CDLL(None).fflush(None)
    except Exception:
        pass


def committed_traffic(root, sha=None):
    """(hbm_bytes_per_launch, source) from the newest ``profiles/r*_pmc_summary.json`` whose ``launch_sources_sha256`` equals the
    sha256 over EVERY source that shapes the dominant launch as they are NOW (tools/launch_hash.py: the kernel, the engine that
    picks its tile / grouping / streams, their headers; ``sha`` overrides it: tests), else (None, None).  The caller uses it for
    the default configuration only (no --opt knob, default first layer)."""
    import glob
    from tools.launch_hash import launch_sources_sha, LAUNCH_SOURCES
    if sha is None:
        sha = launch_sources_sha(root)
    for f in sorted(glob.glob(os.path.join(root, "profiles", "r*_pmc_summary.json")), reverse=True):
        try:
            pj = json.load(open(f))
        except (OSError, ValueError):
            continue
        if sha and pj.get("launch_sources_sha256") == sha and pj.get("hbm_bytes_per_launch") and not pj.get("bench_opts"):
            return pj["hbm_bytes_per_launch"], {"file": os.path.relpath(f, root), "git": pj.get("git"),
                                                "kind": "committed counters of a companion run (tools/pmc_bench.sh) of the default "
                                                        "configuration; " + ", ".join(os.path.basename(x) for x in LAUNCH_SOURCES)
                                                        + " unchanged since (sha256 match)"}
    return None, None


def build_model(args, I, H, L, numerics):
    from rectorch_amd.utils import hash_state_dict
    from rectorch_amd.nets import MultiVAE_net
    from rectorch_amd.models import MultiVAE
    Cd = args.cond_dim
    if Cd:
        from rectorch_amd.nets import CMultiVAE_net
        from rectorch_amd.models import CMultiVAE
        net = CMultiVAE_net(Cd, [L, H, I], dropout=0.5)
        model = CMultiVAE(net, beta=0.2, anneal_steps=100000, learning_rate=1e-3, numerics=numerics)
    else:
        net = MultiVAE_net([L, H, I], dropout=0.5)
        model = MultiVAE(net, beta=0.2, anneal_steps=100000, learning_rate=1e-3, numerics=numerics)
    net.load_state_dict({k: torch.from_numpy(v) for k, v in hash_state_dict([I + Cd, H, L], [L, H, I], "vae", 1234).items()})
    net.train()
    return net, model


def timed_windows(run, steps, windows, world, start):
    """`windows` windows of exactly `steps` steps, each bracketed by barrier + synchronize; per window MAX over ranks"""
    import torch.distributed as dist
    out = []
    for w in range(windows):
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        t0 = time.perf_counter()
        run(steps, start + w * steps)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        el = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([el], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item())
        out.append(el)
    return out


def self_launch(n):
    """``python bench.py --gpus N`` without a launcher: start the N ranks (this script again, one process per visible GPU, the
    environment ``python -m torch.distributed.run`` would export, rendezvous on 127.0.0.1 at a free port), pass rank 0's stdout
    through (the ONE JSON line), keep every rank's stderr, and exit non-zero with the failing rank's stderr as soon as one fails."""
    import socket
    import subprocess
    import tempfile
    gloo = os.environ.get("RTX_DIST_BACKEND") == "gloo"      # tests: the ranks share one GPU
    n_dev = torch.cuda.device_count()
    if n_dev < 1 or (n_dev < n and not gloo):
        sys.stderr.write("bench.py --gpus %d: %d HIP device(s) visible; one MI355X per rank is needed "
                         "(RTX_DIST_BACKEND=gloo lets the ranks share a device: functional tests only)\n" % (n, n_dev))
        return 2
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs, logs = [], []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RTX_BENCH_SELF_LAUNCHED="1")
        env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or n) // n)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        log = tempfile.TemporaryFile(mode="w+")
        logs.append(log)
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else log, stderr=log))
    failed = None
    try:
        while failed is None and any(p.poll() is None for p in procs):
            for r, p in enumerate(procs):
                if p.poll() is not None and p.returncode != 0:
                    failed = r
                    break
            time.sleep(0.05)
        if failed is None:
            failed = next((r for r, p in enumerate(procs) if p.returncode != 0), None)
        if failed is not None:
            t_end = time.time() + 5.0           # the peers of a failed rank block in a collective: give them a moment, then stop them
            while time.time() < t_end and any(p.poll() is None for p in procs):
                time.sleep(0.05)
    finally:
        for p in procs:                         # (exactly the processes started above)
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                p.kill()
    def tail(r, limit=6000):
        logs[r].seek(0)
        return logs[r].read()[-limit:]
    if failed is not None:
        sys.stderr.write("bench.py --gpus %d: rank %d exited with code %s; its output:\n%s\n" % (n, failed, procs[failed].returncode, tail(failed)))
        return procs[failed].returncode if procs[failed].returncode and procs[failed].returncode > 0 else 1
    sys.stderr.write(tail(0))                   # rank 0's warnings (RCCL's informational lines) stay visible, after the result
    return 0


def main():
    args = parse()
    if args.gpus > 1 and os.environ.get("WORLD_SIZE", "1") in ("", "1") and "RTX_BENCH_SELF_LAUNCHED" not in os.environ:
        # no launcher around this process (or one that says "one rank" while N were asked for): start the N ranks here
        sys.exit(self_launch(args.gpus))
    from rectorch_amd import parallel
    rank, world, local = parallel.init_from_env()
    assert world == args.gpus, "WORLD_SIZE=%d but --gpus %d: under a launcher, start --nproc-per-node %d" % (world, args.gpus, args.gpus)
    assert torch.cuda.is_available(), "bench.py measures the HIP path: an MI355X is required"
    import torch.distributed as dist
    from rectorch_amd.utils import synth_interactions
    from rectorch_amd.samplers import DataSampler
    from rectorch_amd.engine import RowBatch

    I, H, L = args.items, 600, 200
    emu = args.emulate_world
    assert not (emu and world > 1), "--emulate-world runs on ONE GPU"
    global_batch = args.batch * world if args.scaling == "weak" else args.batch
    if args.workload == "netflix":
        X = synth_interactions(args.users, I, mu=4.3, sigma=1.0, dmax=5000, seed=20240927)
    else:
        X = synth_interactions(args.users, I, seed=20240927)          # same matrix on every rank
    Cd = args.cond_dim
    if Cd:
        from scipy.sparse import csr_matrix, hstack
        rs = np.random.RandomState(7)
        cu = rs.randint(-1, Cd, size=args.users)              # -1: unconditioned example
        has = cu >= 0
        onehot = csr_matrix((np.ones(int(has.sum()), dtype=X.dtype), (np.nonzero(has)[0], cu[has])), shape=(args.users, Cd))
        Xin, Xtg = hstack([X, onehot], format="csr"), X
    else:
        Xin, Xtg = X, None
    net, model = build_model(args, I, H, L, args.numerics)
    if world == 1 and args.force_dp:
        # a file store: under torch.distributed.run a tcp:// init would wait for the elastic agent's store
        # (TORCHELASTIC_USE_AGENT_STORE makes every rank a client) on a port nobody serves
        import tempfile
        store = os.path.join(tempfile.mkdtemp(prefix="rtx_dp1_"), "store")
        dist.init_process_group("nccl", init_method="file://" + store, rank=0, world_size=1)
    rccl_ranks = None
    dp = world > 1 or args.force_dp or emu > 0
    plan = None
    if emu:
        # rank 0 of an emu-rank job on this one GPU: weak scaling keeps --batch users here, strong scaling --batch / emu
        if args.scaling == "strong":
            global_batch, args.batch = args.batch, max(1, args.batch // emu)
        else:
            global_batch = args.batch * emu
        plan = parallel.attach(model, fixed_global_batch=global_batch, sharded=args.sharded, emulate_world=emu)
    elif dp:
        probe = torch.ones(1, device="cuda")
        dist.all_reduce(probe)                                 # an actual RCCL collective: the line is self-checking
        rccl_ranks = int(round(float(probe.item())))
        plan = parallel.attach(model, fixed_global_batch=global_batch, sharded=args.sharded, engine=args.dp_engine, transport=args.dp_transport)
    # resident sampler over the global batch; each rank takes its slice of every global batch
    np.random.seed(20240927)
    smp = DataSampler(Xin, Xtg, batch_size=global_batch, shuffle=True)
    batches = []
    for rb in smp.iter_rows():
        if len(rb) < global_batch:
            break
        sb = parallel.shard_batch(rb, 0 if emu else rank, emu if emu else world)    # (knows the global batch: no per-step collective)
        batches.append(RowBatch(sb.tr, sb.te if Cd else None, sb.rows, global_len=sb.global_len if dp else None))
    B = len(batches[0])                                        # users per GPU per step (this rank)
    torch.manual_seed(1000 + rank)

    def run(n, start):
        # as MultiVAE.train_epoch does with the resident sampler: every step announces the batch after it, which the engine gathers
        # on its side stream under the step's last weight kernel (single GPU; rtx_engine_set_next_batch).  Every step still pays
        # for one gather: step k's launch carries batch k + 1's.  Like train_epoch, the steps leave the join with the engine's side
        # stream to the next step (RTX_STEP_DEFER_JOIN); the window's closing torch.cuda.synchronize() drains every stream.
        for i in range(n):
            k = start + i
            model._fused_step(batches[k % len(batches)], None, want_loss=False,
                              next_x=None if args.no_prefetch else batches[(k + 1) % len(batches)], defer_join=not args.no_defer_join)

    st_, _, m_, v_ = model._ensure_train_state()        # the engine exists before its first step: some knobs must be set by then
    eng0 = net.rtx_engine(args.numerics, B, train_buffers=(st_.grads, m_, v_))
    eng0.set_option("sparse_in", int(args.first_layer == "sparse"))
    for kv in args.opt:                                 # measurement knobs (rtx_engine_set_option), e.g. --opt two_stream=0
        k, v = kv.split("=")
        eng0.set_option(k, int(v))
    sites = ("adam",) if (dp or args.numerics != "bf16") else ("dW_adam_out", "dW_adam_in")
    if dp:   # the exchange of each bucket, on the stream it runs on (caller's stream = the end of the step's critical path)
        sites += ("dp_exchange_main", "dp_exchange_side", "dp_allgather_main", "dp_allgather_side")
    if not args.no_kernel_timing:
        # every timed bracket is followed by an EMPTY one on the same stream (site "<name>#empty"): what two event records cost with
        # nothing between them.  A bracket holds its kernel PLUS that; the reported duration is bracket - empty bracket, both listed.
        eng0.set_option("timing_calibrate", 1)
    for sname in (() if args.no_kernel_timing else sites):
        # HIP events around the dominant kernel, on the stream it runs on, inside the timed region; every 8th launch is
        # bracketed (the two records of a timed launch cost the step ~5 us each on its critical stream: 322 vs 313 us measured).
        # Switched on BEFORE the pre-heat: the event pool is built there, not in the first timed window.
        eng0.set_timing(sname, args.kernel_timing_every)
    # pre-heat: untimed steps for ~preheat_seconds.  Every rank runs the SAME number of steps (a data-parallel step is a
    # collective): the count comes from a short probe whose duration is max-reduced over the ranks.
    preheat_steps = 0
    if args.preheat_seconds > 0:
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run(4, 0)
        torch.cuda.synchronize()
        per = (time.perf_counter() - t0) / 4
        if world > 1:
            t = torch.tensor([per], device="cuda", dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            per = float(t.item())
        # (the probe's 4 steps include one-off costs -- the first step builds the side stream -- so the count is a lower bound)
        preheat_steps = int(min(4000, max(0, args.preheat_seconds / max(per, 1e-6))))
        run(preheat_steps, 4)
        preheat_steps += 4
    run(args.warmup, 0)
    torch.cuda.synchronize()
    _flush_c_stdio()
    eng = net._rtx_engines[args.numerics]
    eng.get_timings()                                   # (drop what the pre-heat and the warm-up recorded: the timed windows only)
    wins = timed_windows(run, args.steps, args.windows, world, args.warmup)
    timings = eng.get_timings()
    eng.set_timing(None, False)
    n_steps_total = preheat_steps + args.warmup + args.steps * args.windows
    loss_mean = model._read_loss_sum() / n_steps_total

    comm = None
    replica_check = None
    if dp:
        G = emu or world
        ar, rs_, ag = (eng.get_option("dp_bytes_" + k) for k in ("all_reduce", "reduce_scatter", "all_gather"))
        every = 1 if args.no_kernel_timing else args.kernel_timing_every

        def site_us(name):       # mean duration of one timed bracket of this site
            ms_, n_ = timings.get(name, (0.0, 0))
            return (ms_ * 1e3 / n_) if n_ else None
        comm = {"collectives_per_step": eng.get_option("dp_collectives"),
                "bytes_per_step_per_rank": {"all_reduce": ar, "reduce_scatter": rs_, "all_gather": ag, "total": ar + rs_ + ag,
                                            "what": "buffer bytes handed to the collectives (gradient images in comm dtype, compute copies)"},
                "ring_wire_bytes_per_step_per_rank": (2.0 * ar + rs_ + ag) * (G - 1) / G,
                "exchange_us": {"main_stream_reduce": site_us("dp_exchange_main"), "main_stream_all_gather": site_us("dp_allgather_main"),
                                "side_stream_reduce": site_us("dp_exchange_side"), "side_stream_all_gather": site_us("dp_allgather_side"),
                                "timed_every": every},
                # what sits on the step's critical path: the caller's stream runs bucket B's collectives behind the last
                # weight-gradient launch, nothing can hide them; bucket A's (side stream) run beside the data-gradient chain
                "exposed_us": sum(x or 0.0 for x in (site_us("dp_exchange_main"), site_us("dp_allgather_main"))),
                "emulated": bool(emu)}
    if dp and not emu:
        # every rank checksums its parameters after the timed region: replicated -> bit-identical; sharded -> consolidate()
        # (collective) first, then bit-identical.  MIN == MAX over the ranks of a per-tensor bit-pattern checksum.
        if args.sharded:
            model.consolidate()
        chk = torch.stack([(p.detach().view(torch.int32).to(torch.int64) % 1000003).sum() for p in net.parameters()]).to(torch.float64)
        lo_, hi_ = chk.clone(), chk.clone()
        if dist.is_initialized() and dist.get_world_size() > 1:
            dist.all_reduce(lo_, op=dist.ReduceOp.MIN)
            dist.all_reduce(hi_, op=dist.ReduceOp.MAX)
        replica_check = {"mode": "sharded optimizer, consolidate() then compare" if args.sharded else "replicated optimizer",
                         "tensors": int(chk.numel()), "ranks": world, "identical": bool(torch.equal(lo_, hi_)),
                         "finite": bool(all(torch.isfinite(p).all().item() for p in net.parameters()))}
    # N > 1: the north star names the ALL-REDUCE schedule; the default is the sharded optimizer -- time the other one beside it
    # With more than one REAL rank this side measurement is opt-in (--replicated-ab): it tears the first plan's communicators down and
    # brings up two more, which has never run on a multi-GPU node -- nothing after the headline's timed region may be able to take
    # the line with it.  With one rank (--force-dp, what the one-GPU boxes can run) it stays on.
    replicated_ab = None
    if ((world > 1 and args.replicated_ab) or (world == 1 and args.force_dp)) and not emu and args.sharded and not args.no_extras:
        # (the first plan's communicator goes before the second comes up: never two live RCCL communicators in the process;
        #  `--force-dp --sharded` runs this block with ONE real RCCL rank, which is how it is exercised on the one-GPU boxes)
        if world > 1:
            dist.barrier()
        if hasattr(plan, "close"):
            plan.close()
        try:
            net_r, model_r = build_model(args, I, H, L, args.numerics)
            plan_r = parallel.attach(model_r, fixed_global_batch=global_batch, sharded=False, engine=args.dp_engine, transport=args.dp_transport)
            st_r, _, m_r, v_r = model_r._ensure_train_state()
            net_r.rtx_engine(args.numerics, B, train_buffers=(st_r.grads, m_r, v_r)).set_option("sparse_in", int(args.first_layer == "sparse"))

            def run_r(n, start):
                for i in range(n):
                    model_r._fused_step(batches[(start + i) % len(batches)], None, want_loss=False)
            k_r = max(10, min(args.steps, 50))
            run_r(5, 0)
            w_r = timed_windows(run_r, k_r, 2, world, 5)
            replicated_ab = {"ms_per_step": float(np.median(w_r)) / k_r * 1e3, "value": global_batch * k_r / float(np.median(w_r)), "unit": "users/s",
                             "steps": k_r, "what": "the same job with all-reduce + the whole Adam update on every rank (--replicated)"}
            if hasattr(plan_r, "close"):
                plan_r.close()
            del net_r, model_r
        except Exception as ex:                      # the headline line must survive a failure of the side measurement
            replicated_ab = {"error": repr(ex)[:300]}
    if world > 1:
        dist.barrier()
    if plan is not None and hasattr(plan, "close"):
        plan.close()                 # detaches the engine; the engine's own RCCL communicator goes while every rank is still here
    if rank != 0:
        dist.destroy_process_group()
        return
    elapsed = float(np.median(wins))
    ms_step = elapsed / args.steps * 1e3
    # (emulation: ONE GPU's share of the job is measured; the job-level figure would be users_here * G / time only if the
    #  exchange itself were free -- it is reported as what this GPU processed)
    value = (B if emu else global_batch) * args.steps / elapsed
    sparse_first = bool(eng.get_option("last_sparse_in"))
    step_bytes, step_flops = eng.step_cost(B)
    P = sum(p.numel() for p in net.parameters())
    traffic = None
    if dp or args.numerics != "bf16":
        # one multi-tensor Adam launch per step (single GPU, float32) or one per gradient bucket (data parallel)
        adam_ms, adam_n = timings.get("adam", (0.0, 0))
        launches_per_step = max(1, round(adam_n * (1 if args.no_kernel_timing else args.kernel_timing_every) / max(1, args.steps * args.windows)))
        adam_us = adam_ms * 1e3 / max(adam_n, 1) * launches_per_step
        # Adam reads p,g,m,v (16 B/param) and writes p,m,v (12 B/param); with the bf16 gradient exchange of the
        # data-parallel bf16 mode the reduced gradient is read as bf16 (2 B/param less); a sharded optimizer touches P / N
        per_param = 26.0 if dp and args.numerics == "bf16" else 28.0
        shard = (emu or world) if (dp and args.sharded) else 1
        kbytes = per_param * P / shard
        kname = "k_adam (multi-tensor Adam + compute-copy refresh)"
        kus, kn = adam_us, adam_n
    else:
        # the fused weight-gradient + Adam kernel, one launch per weight matrix; the two n_items x 600 matrices are 98 % of
        # the parameters.  Algorithmic bytes per launch: p, exp_avg, exp_avg_sq read and written = 24 B per parameter
        # (the gradient's 4 + 4 B of SURVEY 8d's 32 B/param no longer exist; the 2-B compute copy and the operand reads
        # are not counted)
        (t_out, n_out), (t_in, n_in) = timings.get("dW_adam_out", (0.0, 0)), timings.get("dW_adam_in", (0.0, 0))
        (c_out, m_out), (c_in, m_in) = timings.get("dW_adam_out#empty", (0.0, 0)), timings.get("dW_adam_in#empty", (0.0, 0))
        kn = n_out + n_in
        kus_bracket = (t_out + t_in) * 1e3 / max(kn, 1)
        bracket_overhead_us = (c_out + c_in) * 1e3 / max(m_out + m_in, 1)
        kus = max(kus_bracket - bracket_overhead_us, 0.0)
        kbytes = 24.0 * I * H
        kname = ("rtx_dw_tn / rtx_dw_tn_group <64x128, RTX_DW_ADAM> (weight gradient fused with Adam: the decoder n_items x 600 matrix on "
                 "the side stream, the encoder matrix + the hidden layers' in one launch on the caller's; 2 launches/step)")
    traffic_src = None
    if args.pmc_json:
        # HBM bytes come from PMC counters, which need rocprofv3 --pmc passes of their own (tools/pmc_bench.sh): a companion
        # run of the same build, named here -- never a constant
        pj = json.load(open(args.pmc_json))
        traffic, traffic_src = pj.get("hbm_bytes_per_launch"), {"file": os.path.relpath(args.pmc_json, ROOT), "git": pj.get("git")}
    elif (not dp and args.numerics == "bf16" and args.workload == "ml20m" and B == 500 and I == 20108 and not args.opt
          and args.first_layer == "dense" and not Cd):
        # no counter file given (the driver's run): the newest committed counter summary of THIS shape counts only while the
        # dominant kernel's source is byte-for-byte what was profiled (its sha256 travels in the summary); otherwise null
        traffic, traffic_src = committed_traffic(ROOT)
    achieved = kbytes / (kus * 1e-6) / 1e9 if kus else None
    # flops actually executed: the sparse first layer replaces the dense forward product 2 * B * I_in * H by ~2 * nnz * H
    exec_flops = step_flops
    if sparse_first:
        nnz_b = float(X.nnz) / X.shape[0] * B          # stored entries of a batch (dropped ones are still walked, as zeros)
        exec_flops = step_flops - 2.0 * B * (I + Cd) * H + 2.0 * nnz_b * H
    mode = ("emulated rank 0 of %d" % emu) if emu else ("dp%d" % world)
    out = {
        "metric": ("CMultiVAE (cond_dim=%d) " % Cd if Cd else "MultiVAE ") + "train users/sec on %s (synthetic, %s-shaped)"
                  % (("ml-20m", "ml-20m") if args.workload == "ml20m" else ("netflix", "Netflix-prize")),
        "value": value, "unit": "users/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_step, "higher_is_better": True, "scaling": args.scaling, "vs_baseline": None,
        "dtype": "bf16" if args.numerics == "bf16" else "f32", "data": "synthetic",
        "config": {"workload": "MultiVAE [%d,600,200] on %s-shaped synthetic CSR %dx%d, B=%d per GPU, dropout 0.5, "
                               "beta 0.2 anneal 100000, Adam lr 1e-3 (BASELINE.json configs[%d])"
                               % (I, "ml-20m" if args.workload == "ml20m" else "Netflix", args.users, I, B,
                                  1 if args.workload == "ml20m" else 3),
                   "global_batch": global_batch,
                   "parallelism": mode + (("-sharded-adam" if args.sharded else "-replicated-adam") if dp else ""),
                   "dp_scheduler": (None if not dp else ("engine (%s)" % plan.transport if getattr(plan, "native", False) else "python reducer")),
                   "first_layer": "sparse (k_spmm_in, VALU)" if sparse_first else "dense (MFMA split-K GEMM)",
                   "second_stream_concurrent": bool(eng.get_option("side_concurrent")),
                   "batch_prefetch": {"steps_started_from_a_prefetched_image": eng.get_option("prefetch_hits"),
                                      "gathers_issued_on_the_side_stream": eng.get_option("prefetch_issued")},
                   "numerics": "bf16 MFMA operands, f32 accumulate, f32 master weights + Adam" if args.numerics == "bf16"
                               else "f32 MFMA (parity mode)"},
        "windows": {"n": args.windows, "steps_each": args.steps, "seconds": wins, "reported": "median"},
        "preheat": {"steps": preheat_steps, "seconds_asked": args.preheat_seconds,
                    "what": "untimed steps before the declared warm-up (steady-state clocks); not part of any timed window"},
        "rccl_ranks": rccl_ranks,
        "comm": comm,
        "replica_check": replica_check,
        "replicated_allreduce": replicated_ab,
        "roofline": {"kernel": kname, "bound": "hbm",
                     "achieved": achieved, "peak": HBM_PEAK_TBS * 1000.0, "unit": "GB/s",
                     "frac": (achieved / (HBM_PEAK_TBS * 1000.0)) if achieved else None,
                     "traffic": traffic, "traffic_source": traffic_src, "algorithmic_bytes_per_launch": kbytes, "avg_us": kus, "launches_timed": kn,
                     "avg_us_event_bracket": locals().get("kus_bracket"), "event_bracket_overhead_us": locals().get("bracket_overhead_us"),
                     # the same fraction from the RAW event bracket (nothing subtracted): the two are printed side by side so that they
                     # cannot drift apart unnoticed; the rocprofv3 --stats average of the same command (profiles/) is the third witness
                     "frac_raw_bracket": ((kbytes / (locals().get("kus_bracket") * 1e-6) / 1e9) / (HBM_PEAK_TBS * 1000.0))
                                         if (kbytes and locals().get("kus_bracket")) else None,
                     "timing": "HIP events on the stream each launch runs on, inside the timed region; avg_us = mean bracket minus the mean EMPTY "
                               "bracket recorded right behind each timed one (two event records with nothing between them); "
                               "profiles/r6_bench_kernel_stats.txt is the rocprofv3 summary of the same command",
                     "timed_every": args.kernel_timing_every},
        "step_roofline": {"algorithmic_bytes_per_step": step_bytes, "achieved_GBps": step_bytes / (ms_step * 1e-3) / 1e9,
                          "frac_of_hbm_peak": step_bytes / (ms_step * 1e-3) / 1e9 / (HBM_PEAK_TBS * 1000.0),
                          "algorithmic_flops_per_step": step_flops, "executed_flops_per_step": exec_flops,
                          "achieved_TFLOPs": exec_flops / (ms_step * 1e-3) / 1e12,
                          "frac_of_bf16_mfma_peak": exec_flops / (ms_step * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF},
        "mean_loss": loss_mean,
    }
    if world == 1 and not dp and args.numerics == "bf16" and not args.no_extras and not Cd:
        # (a) the same steps through the PUBLIC train_batch (reference models.py:835: `return loss.item()` -- one host sync per step)
        k_api = max(10, min(args.steps, 100))
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(k_api):
            model.train_batch(batches[i % len(batches)])
        torch.cuda.synchronize()
        e_api = time.perf_counter() - t0
        out["train_batch_api"] = {"value": global_batch * k_api / e_api, "unit": "users/s", "ms_per_step": e_api / k_api * 1e3, "steps": k_api,
                                  "what": "MultiVAE.train_batch(rows) per step, returning loss.item() like the reference (models.py:835)"}
        # (b) A/B of the first layer: the other product (headline = dense MFMA contraction; alternative = sparse VALU product)
        eng.set_option("sparse_in", int(not sparse_first))
        run(5, 0)
        wd = timed_windows(run, k_api, 2, 1, 5)
        eng.set_option("sparse_in", int(sparse_first))
        ms_other = float(np.median(wd)) / k_api * 1e3
        if sparse_first:
            out["first_layer_dense_mfma"] = {"ms_per_step": ms_other, "steps": k_api,
                                             "frac_of_bf16_mfma_peak": step_flops / (ms_other * 1e-3) / 1e12 / MFMA_BF16_PEAK_TF}
        else:
            nnz_b = float(X.nnz) / X.shape[0] * B
            out["first_layer_sparse_valu"] = {"ms_per_step": ms_other, "steps": k_api, "value": global_batch / (ms_other * 1e-3), "unit": "users/s",
                                              "executed_flops_per_step": step_flops - 2.0 * B * (I + Cd) * H + 2.0 * nnz_b * H,
                                              "what": "the same step with the first layer as a sparse VALU product over the stored entries "
                                                      "(k_in_chunks -> k_spmm_in): 12 GFLOP of dense product replaced by 2 nnz H"}
    if world == 1 and not dp and args.numerics == "bf16" and not args.no_fp32_parity and not Cd:
        # the float32 parity mode (exact-f32 MFMA, the arithmetic the 1e-5 logits criterion is met in): same workload, a
        # shorter sample; bound by the f32 MFMA rate ("two numerics modes ... report both")
        net32, model32 = build_model(args, I, H, L, "fp32")
        torch.manual_seed(1000)

        def run32(n, start):
            for i in range(n):
                model32._fused_step(batches[(start + i) % len(batches)], None, want_loss=False)
        k32 = max(10, min(args.steps, 50))
        run32(5, 0)
        w32 = timed_windows(run32, k32, 2, 1, 5)
        e32 = float(np.median(w32))
        out["fp32_parity"] = {"value": global_batch * k32 / e32, "unit": "users/s", "ms_per_step": e32 / k32 * 1e3,
                              "steps": k32, "windows": w32, "achieved_TFLOPs": step_flops / (e32 / k32) / 1e12,
                              "frac_of_f32_mfma_peak": step_flops / (e32 / k32) / 1e12 / F32_MFMA_PEAK_TF,
                              "peak_TFLOPs": F32_MFMA_PEAK_TF}
        del net32, model32
    if world == 1 and not args.no_cpu_baseline and not Cd:
        out["cpu_baseline"] = cpu_baseline(X, (I, H, L), min(B, 500), args.cpu_seconds)
    if dist.is_initialized():
        dist.destroy_process_group()      # RCCL may print while it shuts down: keep the JSON line the last one
    _flush_c_stdio()
    sys.stdout.flush()
    print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
