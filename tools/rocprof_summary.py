#!/usr/bin/env python
"""Turn rocprofv3's rocpd sqlite output (<name>_results.db) into the text summaries kept under profiles/.

    python tools/rocprof_summary.py stats  gpurun_out/prof/stats/bench_results.db   > profiles/rNN_kernel_stats.txt
    python tools/rocprof_summary.py pmc    gpurun_out/prof/pmc_fetch/eng_results.db > profiles/rNN_pmc_fetch.txt
"""
import sqlite3
import sys


def stats(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats : per-kernel summary (durations in us)")
    print("%-64s %7s %12s %10s %10s %10s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us", "max_us", "pct"))
    for n, c, s, a, mn, mx in rows:
        print("%-64s %7d %12.1f %10.2f %10.2f %10.2f %6.2f" % (n[:64], c, s / 1e3, a / 1e3, mn / 1e3, mx / 1e3, 100.0 * s / tot))


def pmc(path):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select kernel_name, counter_name, grid_size, count(*), avg(value), avg(duration) from counters_collection "
                       "group by kernel_name, counter_name, grid_size order by avg(value)*count(*) desc").fetchall()
    print("# rocprofv3 --pmc : per (kernel, grid) average counter value per launch; duration in us (profiled run)")
    print("%-56s %-12s %10s %6s %16s %10s" % ("kernel", "counter", "grid", "calls", "avg_value", "avg_us"))
    for k, c, g, n, v, d in rows:
        print("%-56s %-12s %10d %6d %16.1f %10.2f" % (k[:56], c, g, n, v, d / 1e3))


def bygrid(path):
    """per (kernel, grid size) totals: separates the big and the small nodes of a recursion that reuses one kernel"""
    cur = sqlite3.connect(path).cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)").fetchall()]
    gx = "grid_x" if "grid_x" in cols else "grid_size_x" if "grid_size_x" in cols else "grid_size"
    gy = "grid_y" if "grid_y" in cols else "grid_size_y" if "grid_size_y" in cols else "0"
    rows = cur.execute("select name, %s, %s, count(*), sum(duration), avg(duration) from kernels group by name, %s, %s "
                       "order by sum(duration) desc" % (gx, gy, gx, gy)).fetchall()
    print("# per (kernel, grid) totals (us); columns of the kernels view: %s" % ",".join(cols))
    for n, x, y, c, sm, a in rows[:60]:
        print("%-40s grid=(%s,%s) calls=%d total=%.1f avg=%.2f" % (n[:40], x, y, c, sm / 1e3, a / 1e3))


def timeline(path, anchor="k_gather", nth=40, count=2):
    """start / end of every kernel of `count` consecutive steps (a step begins at the `nth` launch of the `anchor` kernel), in
    us from the step's first kernel, with the queue it ran on: shows what overlaps what and the gaps between dependent kernels"""
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute("select name, start, end, queue_id, stream_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if anchor in r[0]]
    if not idx:   # the sparse first layer: the step opens with k_in_chunks instead of k_gather
        idx = [i for i, r in enumerate(rows) if "k_in_chunks" in r[0]]
    if len(idx) < nth + count + 1:
        nth = max(0, len(idx) - count - 1)
    lo, hi = idx[nth], idx[nth + count]
    t0 = rows[lo][1]
    print("# kernel timeline of %d step(s): start_us end_us dur_us gap_to_prev_end_on_same_queue queue stream kernel" % count)
    last_end = {}
    for n, st, en, q, sid in rows[lo:hi]:
        gap = (st - last_end[q]) / 1e3 if q in last_end else 0.0
        last_end[q] = en
        print("%9.2f %9.2f %8.2f %8.2f  q%-3s s%-3s %s" % ((st - t0) / 1e3, (en - t0) / 1e3, (en - st) / 1e3, gap, q, sid, n[:70]))


def _read_pmc_table(path):
    """rows of a `pmc` summary written above: (kernel, counter, grid, calls, avg_value, avg_us)"""
    out = []
    for ln in open(path):
        if ln.startswith("#") or ln.startswith("kernel "):
            continue
        # the kernel name is a fixed-width field of 56 characters and may contain spaces
        name, rest = ln[:56].strip(), ln[56:].split()
        if len(rest) == 5:
            out.append((name, rest[0], int(rest[1]), int(rest[2]), float(rest[3]), float(rest[4])))
    return out


N_XCD, SIMD_PER_XCD, CLOCK_GHZ = 8, 32 * 4, 2.4     # MI355X: 8 XCDs x 32 CUs x 4 SIMDs, 2.4 GHz (MI355X_MICROARCH.md)


def mfma(path):
    """MFMA utilisation per kernel from the SQ pass.  SQ_VALU_MFMA_BUSY_CYCLES counts busy cycles summed over all SIMDs (32 per
    v_mfma_f32_32x32x16_bf16, MI355X_MICROARCH.md); GRBM_GUI_ACTIVE arrives summed over the 8 XCDs (a 37-us kernel reports 0.8 M
    cycles = 8 x 37 us x 2.7 GHz incl. the counter start/stop), so
        util_active = MFMA_BUSY / (GUI_ACTIVE x 128 SIMDs per XCD)            (the counters' own time base)
        util_wall   = MFMA_BUSY / (duration x 2.4 GHz x 1024 SIMDs)           (the traced duration)
    1.0 = every SIMD's matrix pipe busy all the time = the dense peak of the instruction mix (2.5 PFLOP/s for bf16)."""
    rows = _read_pmc_table(path)
    by = {}
    for name, ctr, grid, calls, val, us in rows:
        by.setdefault((name, grid), {})[ctr] = (val, us, calls)
    print("# MFMA utilisation per kernel (rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE; counters serialise")
    print("# the streams: every kernel is ALONE on the device here, and the step runs its one-stream schedule).")
    print("# util_active = MFMA_BUSY / (GRBM_GUI_ACTIVE [summed over %d XCDs] x %d SIMDs per XCD); util_wall = MFMA_BUSY / (avg_us x %.1f GHz x %d SIMDs)"
          % (N_XCD, SIMD_PER_XCD, CLOCK_GHZ, N_XCD * SIMD_PER_XCD))
    print("%-56s %10s %6s %10s %14s %14s %12s %10s" % ("kernel", "grid", "calls", "avg_us", "mfma_busy_cyc", "gui_active_cyc", "util_active", "util_wall"))
    for (name, grid), c in sorted(by.items(), key=lambda kv: -kv[1].get("SQ_VALU_MFMA_BUSY_CYCLES", (0, 0, 0))[0]):
        busy = c.get("SQ_VALU_MFMA_BUSY_CYCLES", (0.0, 0.0, 0))
        act = c.get("GRBM_GUI_ACTIVE", (0.0, 0.0, 0))
        ua = busy[0] / (act[0] * SIMD_PER_XCD) if act[0] else float("nan")
        uw = busy[0] / (busy[1] * 1e3 * CLOCK_GHZ * N_XCD * SIMD_PER_XCD) if busy[1] else float("nan")
        print("%-56s %10d %6d %10.2f %14.0f %14.0f %12.4f %10.4f" % (name, grid, busy[2], busy[1], busy[0], act[0], ua, uw))


def pmcjson(directory):
    """FETCH_SIZE / WRITE_SIZE summaries of one tools/pmc_bench.sh run -> JSON: HBM bytes per launch of every kernel and of the
    dominant pair (the two weight-gradient + Adam launches).  gfx950 correction: FETCH_SIZE x 2 (MI355X_MICROARCH.md, HBM)."""
    import json
    import os
    import subprocess
    f = {(r[0], r[2]): r for r in _read_pmc_table(os.path.join(directory, "bench_FETCH_SIZE.txt"))}
    w = {(r[0], r[2]): r for r in _read_pmc_table(os.path.join(directory, "bench_WRITE_SIZE.txt"))}
    kernels = {}
    for key, fr in f.items():
        wr = w.get(key)
        if wr is None:
            continue
        rd, wrb = fr[4] * 1024.0 * 2.0, wr[4] * 1024.0          # the counters report KB
        kernels["%s grid=%d" % key] = {"calls": fr[3], "FETCH_SIZE_KB_reported": fr[4], "WRITE_SIZE_KB": wr[4], "read_bytes_per_launch": rd,
                                       "write_bytes_per_launch": wrb, "hbm_bytes_per_launch": rd + wrb, "avg_us_serialised": fr[5],
                                       "hbm_TBps": (rd + wrb) / (fr[5] * 1e-6) / 1e12 if fr[5] else None}
    dw = [v for k, v in kernels.items() if k.startswith("void rtx_dw_tn") and v["calls"] >= 10]
    top = max((v["hbm_bytes_per_launch"] for v in dw), default=0.0)
    dom = [v for v in dw if v["hbm_bytes_per_launch"] >= 0.5 * top]      # the n_items x 600 matrices, not the hidden layers' launches
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    git = None
    try:
        git = subprocess.check_output(["git", "-C", root, "rev-parse", "--short", "HEAD"], text=True, stderr=subprocess.DEVNULL).strip()
    except Exception:
        if os.path.exists(os.path.join(root, ".git_rev")):                # the GPU box has no .git: tools/gpu.sh leaves the hash here
            git = open(os.path.join(root, ".git_rev")).read().strip()
    import hashlib
    sys.path.insert(0, root)
    from tools.launch_hash import launch_sources_sha, LAUNCH_SOURCES
    src = os.path.join(root, "rectorch_amd", "csrc", "dw_adam.hip")     # bench.py reports these counters only for the SAME kernel source
    print(json.dumps({"correction": "gfx950 rocprofv3: FETCH_SIZE doubled (reports 1/2 of a wide coalesced streaming read), WRITE_SIZE as reported; KB -> bytes",
                      "source": "tools/pmc_bench.sh: bench.py under rocprofv3, separate --pmc passes (kernels serialised: each is alone on the device)",
                      "git": git, "kernel_source": "rectorch_amd/csrc/dw_adam.hip",
                      "kernel_source_sha256": hashlib.sha256(open(src, "rb").read()).hexdigest() if os.path.exists(src) else None,
                      # every source that shapes the dominant launch (tools/launch_hash.py): what bench.py compares before citing this file
                      "launch_sources_sha256": launch_sources_sha(root), "launch_sources": list(LAUNCH_SOURCES),
                      "bench_opts": [a for a in os.environ.get("RTX_PMC_BENCH_ARGS", "").split() if a],
                      "kernels": kernels,
                      "hbm_bytes_per_launch": (sum(v["hbm_bytes_per_launch"] for v in dom) / len(dom)) if dom else None,
                      "dominant": "mean over the weight-gradient + Adam launches (rtx_dw_tn / rtx_dw_tn_group)"}, indent=1))


def diag(directory):
    """tools/pmc_diag.sh: every counter of every pass, pivoted per (kernel, grid): one line per kernel, one column per counter,
    plus the ratios that answer "what bounds it": L2 hit rate, bytes requested by the L1s (TCP_TCC_*_REQ x 64 B / 128 B as noted),
    share of wave-cycles parked / issue-stalled / issuing."""
    import glob
    import os
    by, order = {}, []
    for f in sorted(glob.glob(os.path.join(directory, "diag_g*.txt"))):
        for name, ctr, grid, calls, val, us in _read_pmc_table(f):
            d = by.setdefault((name, grid), {})
            d[ctr] = val
            d.setdefault("_us", []).append(us)
            if ctr not in order:
                order.append(ctr)
    keys = sorted(by, key=lambda k: -sum(by[k]["_us"]) / len(by[k]["_us"]))
    print("# per-kernel counters (avg per launch; kernels serialised under --pmc).  us = mean traced duration over the passes")
    for k in keys:
        d = by[k]
        us = sum(d["_us"]) / len(d["_us"])
        print("%s grid=%d  avg_us=%.2f" % (k[0][:70], k[1], us))
        for c in order:
            if c in d:
                print("    %-36s %18.1f" % (c, d[c]))
        g = d.get
        if g("TCC_HIT_sum") is not None and g("TCC_MISS_sum") is not None and g("TCC_HIT_sum") + g("TCC_MISS_sum") > 0:
            print("    %-36s %18.4f" % ("L2_hit_rate", g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum"))))
        if g("TCP_TCC_READ_REQ_sum") is not None:
            rd = g("TCP_TCC_READ_REQ_sum")
            print("    %-36s %18.1f  (x64 B = %.1f MB, x128 B = %.1f MB; %.2f / %.2f TB/s)" % ("L1->L2 read requests", rd, rd * 64 / 1e6, rd * 128 / 1e6,
                                                                                         rd * 64 / us / 1e6, rd * 128 / us / 1e6))
        if g("SQ_WAVE_CYCLES"):
            wc = g("SQ_WAVE_CYCLES")
            print("    %-36s parked %.3f  issue-stalled %.3f  issuing %.3f" % ("share of wave-cycles", (g("SQ_WAIT_ANY") or 0) / wc,
                                                                                 (g("SQ_WAIT_INST_ANY") or 0) / wc, (g("SQ_ACTIVE_INST_ANY") or 0) / wc))


if __name__ == "__main__":
    if sys.argv[1] == "timeline" and len(sys.argv) > 3:     # timeline DB ANCHOR [NTH] [COUNT]
        timeline(sys.argv[2], sys.argv[3], int(sys.argv[4]) if len(sys.argv) > 4 else 40, int(sys.argv[5]) if len(sys.argv) > 5 else 2)
    else:
        {"stats": stats, "pmc": pmc, "bygrid": bygrid, "timeline": timeline, "mfma": mfma, "pmcjson": pmcjson, "diag": diag}[sys.argv[1]](sys.argv[2])
