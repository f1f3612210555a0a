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


def _pmc_db(tmp_path, counters):
    """a rocpd-shaped counters_collection view: per (kernel, counter) `calls` rows of `value`"""
    path = str(tmp_path / "pmc_results.db")
    con = sqlite3.connect(path)
    con.execute("create table counters_collection (kernel_name text, counter_name text, grid_size integer, value real, duration integer)")
    for kernel, grid, dur_us, vals in counters:
        for ctr, v in vals.items():
            for _ in range(12):
                con.execute("insert into counters_collection values (?,?,?,?,?)", (kernel, ctr, grid, v, int(dur_us * 1000)))
    con.commit()
    con.close()
    return path


def test_mfma_utilisation_and_hbm_json_from_the_pmc_passes(tmp_path):
    """tools/pmc_bench.sh's post-processing: MFMA busy cycles / (active cycles x 1024 SIMDs), and HBM bytes per launch with the
    gfx950 FETCH_SIZE x 2 correction"""
    import json
    import rocprof_summary as rs
    dw, gemm = "void rtx_dw_tn<2, 4, 3, 1>(RtxDw)", "void rtx_gemm_nt<unsigned short, 1, 2, 2, 2>(RtxGemm)"
    d = tmp_path / "pmc"
    d.mkdir()
    for tag, key in (("FETCH_SIZE", "FETCH_SIZE"), ("WRITE_SIZE", "WRITE_SIZE")):
        sub = tmp_path / tag
        sub.mkdir()
        db = _pmc_db(sub, [(dw, 404480, 72.0, {key: 90000.0 if tag == "FETCH_SIZE" else 170000.0}),
                           (dw, 20480, 14.0, {key: 900.0 if tag == "FETCH_SIZE" else 1700.0}),     # a hidden layer's launch: not "dominant"
                           (gemm, 161792, 36.0, {key: 12000.0 if tag == "FETCH_SIZE" else 41000.0})])
        (d / ("bench_%s.txt" % tag)).write_text(_run(rs.pmc, db))
    j = json.loads(_run(rs.pmcjson, str(d)))
    k = j["kernels"]["%s grid=404480" % dw]
    assert abs(k["read_bytes_per_launch"] - 2 * 90000.0 * 1024) < 1 and abs(k["write_bytes_per_launch"] - 170000.0 * 1024) < 1
    assert abs(j["hbm_bytes_per_launch"] - (2 * 90000.0 + 170000.0) * 1024) < 1          # the dominant pair = the rtx_dw_tn launches
    assert abs(k["hbm_TBps"] - (2 * 90000.0 + 170000.0) * 1024 / 72e-6 / 1e12) < 1e-9
    sq = tmp_path / "SQ"
    sq.mkdir()
    # 36 us at 2.4 GHz = 86 400 cycles per XCD (GRBM_GUI_ACTIVE arrives summed over the 8 XCDs); 1024 SIMDs busy a quarter of the time
    db = _pmc_db(sq, [(gemm, 161792, 36.0, {"SQ_VALU_MFMA_BUSY_CYCLES": 86400.0 * 1024 * 0.25, "GRBM_GUI_ACTIVE": 8 * 86400.0, "SQ_BUSY_CU_CYCLES": 1.0})])
    table = tmp_path / "bench_SQ.txt"
    table.write_text(_run(rs.pmc, db))
    out = [ln for ln in _run(rs.mfma, str(table)).splitlines() if ln.startswith("void rtx_gemm_nt")]
    assert len(out) == 1 and abs(float(out[0].split()[-1]) - 0.25) < 1e-4 and abs(float(out[0].split()[-2]) - 0.25) < 1e-4


def test_dw_stamps_groups_tiles_by_hardware_slot(tmp_path):
    """tools/dw_stamps.py on a hand-made stamp file ([workgroup][8] u64, 100-MHz ticks): two hardware slots, three tiles each,
    2-us gaps -- the report must find the slots, the gaps and the phase means (round 4's probe of the weight-gradient kernel)"""
    import subprocess
    import numpy as np
    rows = []
    for slot in range(2):
        t = 1000 + slot * 37
        for tile in range(3):
            st = t
            marks = [st, st + 400, st + 700, st + 1200, st + 1300, st + 1800, st + 1900]    # issue 4, first 3, walk 5, park 1, adam 5, ack 1 us
            rows.append(marks + [(slot << 32) | (0x100 * slot + 3)])                       # XCC_ID << 32 | HW_ID
            t = marks[-1] + 200                                                              # 2 us until the slot's next tile
    rows.append([0] * 8)                                                                     # a padding workgroup: no stamps
    path = str(tmp_path / "stamps_0.bin")
    np.array(rows, dtype=np.uint64).tofile(path)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dw_stamps.py"), path], capture_output=True, text=True, check=True).stdout
    assert "6 tiles" in out and "mean life 19.00 us" in out
    assert "2 distinct slots; tiles per slot: min 3 max 3" in out
    assert "n 4 mean 2.00" in out                                    # four gaps of 2 us
    assert "issue 4.00 | first slice 3.00 | K walk 5.00 | park 1.00 | adam+stores 5.00 | store ack 1.00" in out


def test_bench_reports_committed_counters_only_for_the_profiled_kernel_source():
    """bench.py's default run (no --pmc-json) takes roofline.traffic from the newest committed counter summary -- only while the
    sha256 of the dominant kernel's source equals the one recorded with the counters; any other source: null."""
    import hashlib
    import json
    import bench
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    from tools.launch_hash import launch_sources_sha, LAUNCH_SOURCES
    sha = launch_sources_sha(root)
    assert sha and len(sha) == 64 and "rectorch_amd/csrc/engine.hip" in LAUNCH_SOURCES and "rectorch_amd/csrc/dw_adam.hip" in LAUNCH_SOURCES
    traffic, source = bench.committed_traffic(root)
    if traffic is not None:            # the committed summary matches EVERY launch-shaping source: it must be the file it names
        pj = json.load(open(os.path.join(root, source["file"])))
        assert pj["launch_sources_sha256"] == sha and pj["hbm_bytes_per_launch"] == traffic and not pj.get("bench_opts")
        # the ml-20m decoder / encoder launches: between the algorithmic 24 B/param and twice that
        assert 24.0 * 20108 * 600 <= traffic <= 48.0 * 20108 * 600
    assert bench.committed_traffic(root, sha="0" * 64) == (None, None)


def test_bench_gpus_n_without_a_launcher_does_not_assert():
    """`python bench.py --gpus 2` with no WORLD_SIZE: bench.py is its own launcher (round 4 asserted "launch with --nproc-per-node").
    Without a HIP device it refuses with exit code 2 and says what it needs -- no traceback, no JSON line."""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        import pytest
        pytest.skip("the refusal path needs a box without a HIP device (the launch itself is tested under -m gpu)")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 2, (out.returncode, out.stderr[-1500:])
    assert "0 HIP device(s) visible" in out.stderr and "Traceback" not in out.stderr and "AssertionError" not in out.stderr
    assert not out.stdout.strip()
