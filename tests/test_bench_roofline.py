"""CPU tests of the measurement plumbing: the counter parser of tools/pmc_live.py on synthetic rocprofv3 CSVs and bench.py's roofline block
(no GPU, no profiler)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _write_pass(d, counters, kernel="void k_query_kh<1>(BftImage, unsigned char const*)", n_dispatch=5, dur_ns=2_700_000):
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "x_counter_collection.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Dispatch_Id", "Kernel_Name", "Counter_Name", "Counter_Value"])
        for disp in range(1, n_dispatch + 1):
            for name, total in counters.items():
                for part in range(8):  # one row per XCD, as rocprofv3 writes them: the parser sums them per dispatch
                    w.writerow([disp, kernel if disp > 2 else "void k_kh_assemble<1>(...)", name, total / 8 * (1.0 if disp > 2 else 0.01)])
    with open(os.path.join(d, "x_kernel_trace.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Kernel_Name", "Start_Timestamp", "End_Timestamp"])
        for disp in range(1, n_dispatch + 1):
            w.writerow([kernel if disp > 2 else "void k_kh_assemble<1>(...)", disp * 10_000_000, disp * 10_000_000 + dur_ns])


def test_pmc_parser_and_roofline_block(tmp_path):
    from tools import pmc_live
    import bench
    nq = 125_000_000
    _write_pass(str(tmp_path / "p0" / "a"), {"FETCH_SIZE": 8_661_539.0})
    _write_pass(str(tmp_path / "p1" / "a"), {"WRITE_SIZE": 53_167.0})
    _write_pass(str(tmp_path / "p2" / "a"), {"TCC_MISS_sum": 140e6, "TCC_REQ_sum": 155e6, "TCC_EA0_RDREQ_sum": 138e6})
    vals, durs = pmc_live._parse(str(tmp_path / "p0"), "k_query", 3)
    assert abs(vals["FETCH_SIZE"] - 8_661_539.0) < 1 and len(durs) == 3 and abs(durs[0] - 2700.0) < 1e-6
    vals2, _ = pmc_live._parse(str(tmp_path / "p2"), "k_query", 3)
    assert abs(vals2["TCC_MISS_sum"] - 140e6) < 1
    # the derived figures, as collect() computes them
    fetch, write = vals["FETCH_SIZE"] * 1024, 53_167.0 * 1024
    per_q = (fetch + write + nq * 7 / 2.0) / nq
    pmc = {"hbm_bytes_per_query": round(per_q, 3), "l2_misses_per_query": 1.12, "l2_requests_per_query": 1.24, "kernel_us_under_pmc_mean": 2700.0, "lib_source_hash": "x"}
    blk = bench.roofline_block(pmc, nq, 2.74, 20, "k_query_kh", 7, True)
    assert 0 < blk["frac"] <= 1 and abs(blk["achieved"] - per_q * nq / 2.74e-3 / 1e9) < 1 and blk["peak"] == 8000.0
    assert abs(blk["wasted"] - per_q / (7 + 0.125 + 64)) < 1e-3 and blk["traffic"] == round(pmc["hbm_bytes_per_query"] * nq)
    assert 0 < blk["gather"]["frac"] <= 1 and blk["gather"]["ceiling_G_per_s"] == bench.GATHER_CEILING_G
    stale = bench.roofline_block({"error": "no profiler"}, nq, 2.74, 20, "k_query_kh", 7, False)
    assert stale["frac"] is None and stale["pmc_stale"] is True and stale["traffic"] is None
    assert len(pmc_live.source_hash()) == 16


def test_committed_counter_file_names_its_sources():
    """profiles/r03/pmc_query.json (what bench.py falls back on without a profiler) carries the hash of the kernel sources it was
    collected on; bench.py uses it only when that hash is the current one."""
    import json
    p = os.path.join(ROOT, "profiles", "r03", "pmc_query.json")
    d = json.load(open(p))
    for wl in ("cfg4", "cfg4k31", "cfg2"):
        assert "lib_source_hash" in d[wl] and d[wl]["queries_per_launch"] > 0 and "hbm_bytes_per_query" in d[wl]
        assert 60 < d[wl]["hbm_bytes_per_query"] < 100 and 1.0 < d[wl]["l2_misses_per_query"] < 1.3
