"""tools/rocprof_summary.py on a hand-made rocpd-shaped sqlite file: the summaries under profiles/ are produced by it on the
GPU box, where a mistake costs a profiling run (CPU test, no GPU, no rocprofv3)."""
import io
import os
import sqlite3
import sys
from contextlib import redirect_stdout

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def _db(tmp_path, first_kernel):
    path = str(tmp_path / "p_results.db")
    con = sqlite3.connect(path)
    con.execute("create table kernels (name text, start integer, end integer, duration integer, queue_id integer, stream_id integer)")
    t = 0
    for step in range(6):
        for name, dur, q in ((first_kernel, 12000, 1), ("k_spmm_in(RtxSpmmInArgs)", 20000, 1), ("void rtx_dw_tn<2, 4, 3, 1>(RtxDw)", 90000, 2),
                             ("void rtx_dw_tn_group<2, 4, 3, 1>(RtxDwGroup)", 80000, 1)):
            con.execute("insert into kernels values (?,?,?,?,?,?)", (name, t, t + dur, dur, q, q - 1))
            t += dur + 1000
    con.commit()
    con.close()
    return path


def _run(fn, *args):
    buf = io.StringIO()
    with redirect_stdout(buf):
        fn(*args)
    return buf.getvalue()


def test_stats_orders_by_total_time_and_reports_microseconds(tmp_path):
    import rocprof_summary as rs
    out = _run(rs.stats, _db(tmp_path, "k_in_chunks(RtxInChunksArgs)")).splitlines()
    assert out[2].startswith("void rtx_dw_tn<2, 4, 3, 1>(RtxDw)")          # 6 x 90 us is the largest total
    cols = out[2].split()
    assert abs(float(cols[-4]) - 90.0) < 1e-6 and int(cols[-6]) == 6       # avg_us, calls
    assert abs(sum(float(l.split()[-1]) for l in out[2:]) - 100.0) < 0.1   # the pct column adds up


def test_timeline_anchors_on_the_first_kernel_of_a_step(tmp_path):
    import rocprof_summary as rs
    for first in ("void k_gather<unsigned short>(RtxGatherArgs)", "k_in_chunks(RtxInChunksArgs)"):   # dense / sparse first layer
        sub = tmp_path / first[:8].strip().replace("<", "_").replace(" ", "_")
        sub.mkdir()
        out = _run(rs.timeline, _db(sub, first)).splitlines()
        assert out[0].startswith("# kernel timeline of 2 step(s)")
        body = out[1:]
        assert len(body) == 8 and body[0].split()[0] == "0.00" and first[:20] in body[0]
        assert first[:20] in body[4]                                         # the second step starts with it again
        # the gap column is measured on the kernel's own queue: 1 us between consecutive launches of queue 1
        assert abs(float(body[1].split()[3]) - 1.0) < 1e-6
