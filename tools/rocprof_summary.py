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


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc, "bygrid": bygrid, "timeline": timeline}[sys.argv[1]](sys.argv[2])
