#!/usr/bin/env python3
"""Turn a rocprofv3 (ROCm 7.x, rocpd sqlite) --kernel-trace --stats result into a short text summary.

usage: summarize_rocpd.py results.db [--pmc]  > profiles/<round>_<what>.txt
"""
import sqlite3
import sys


def short(name, n=110):
    name = name.replace("void ", "")
    return name if len(name) <= n else name[: n - 3] + "..."


def main():
    db = sqlite3.connect(sys.argv[1])
    cur = db.cursor()
    print(f"# source: {sys.argv[1]} (rocprofv3 --kernel-trace --stats)")
    print(f"{'kernel':112s} {'calls':>6s} {'total_us':>12s} {'avg_us':>12s} {'pct':>7s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{short(name):112s} {calls:6d} {tot:12.1f} {avg:12.2f} {pct:7.2f}")
    print("\n# product kernels: launch geometry and registers (first dispatch of each)")
    seen = set()
    q = ("select name,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count,min(duration),max(duration),"
         "avg(duration),count(*) from kernels where name like '%k_%' group by name")
    for row in cur.execute(q):
        name = row[0]
        if "at::native" in name or "rocprim" in name or name in seen:
            continue
        seen.add(name)
        print(f"{short(name, 90):92s} grid={row[1]} wg={row[2]} lds={row[3]} scratch={row[4]} vgpr={row[5]} agpr={row[6]} sgpr={row[7]} "
              f"min/avg/max_us={row[8] / 1e3:.1f}/{row[10] / 1e3:.1f}/{row[9] / 1e3:.1f} n={row[11]}")
    # the timed launches of the bench are the long ones: the warm-up index and the residency tuning of bft_gpu_build launch
    # the same kernels on small batches, so list the query kernels again by duration class
    print("\n# query kernels by duration class (bench.py times the launches of the full batch)")
    q = ("select name, case when duration >= 1000000 then '>=1ms' else '<1ms' end as cls, count(*), avg(duration), min(duration), max(duration) "
         "from kernels where name like '%k_query%' or name like '%k_branching%' group by name, cls order by name, cls")
    for name, cls, n, avg, mn, mx in cur.execute(q):
        print(f"{short(name, 90):92s} {cls:6s} n={n:4d} avg_us={avg / 1e3:10.2f} min/max_us={mn / 1e3:.1f}/{mx / 1e3:.1f}")
    # one line per (kernel, grid size): the full-batch launches of a persistent kernel have the full grid, tuning / warm-up
    # launches a smaller one or a much shorter duration -- the per-launch figure bench.py reports can be read off here
    print("\n# product kernels per (kernel, grid, workgroup): n, avg / min / max us")
    q = ("select name, grid_x, workgroup_x, count(*), avg(duration), min(duration), max(duration), sum(duration) from kernels "
         "where name like '%k_%' and name not like '%at::native%' and name not like '%rocprim%' group by name, grid_x, workgroup_x order by sum(duration) desc")
    for name, gx, wx, n, avg, mn, mx, tot in cur.execute(q):
        print(f"{short(name, 90):92s} grid={gx:<9d} wg={wx:<5d} n={n:4d} avg_us={avg / 1e3:10.2f} min/max_us={mn / 1e3:.1f}/{mx / 1e3:.1f} total_us={tot / 1e3:.1f}")
    # the same persistent grid serves several batches in one run (bench.py: the config-2 batch, the config-4 shares, the k=31
    # index, tuning batches): launches of one (kernel, grid) in dispatch order, consecutive launches whose durations stay within
    # 12 % of the run's first one form a run -- the K timed steps of the bench are one such run of >= K equal launches
    print("\n# query kernels: runs of consecutive launches of similar duration (kernel, grid): n, avg / min / max us")
    q = ("select name, grid_x, duration from kernels where (name like '%k_query%' or name like '%k_branching%' or name like '%k_seq_walk%') "
         "order by name, grid_x, start")
    runs, cur_key, cur_run = [], None, []
    def flush():
        if len(cur_run) >= 3:
            runs.append((cur_key, list(cur_run)))
    for name, gx, dur in cur.execute(q):
        key = (name, gx)
        if key != cur_key or not cur_run or abs(dur - cur_run[0]) > 0.12 * cur_run[0]:
            flush()
            cur_key, cur_run = key, []
        cur_run.append(dur)
    flush()
    for (name, gx), r in runs:
        if sum(r) / len(r) < 200000:  # (runs of sub-0.2 ms launches: warm-up and tuning)
            continue
        print(f"{short(name, 90):92s} grid={gx:<9d} n={len(r):4d} avg_us={sum(r) / len(r) / 1e3:10.2f} min/max_us={min(r) / 1e3:.1f}/{max(r) / 1e3:.1f}")
    if "--pmc" in sys.argv:
        tabs = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")]
        print("\n# tables:", [t for t in tabs if "pmc" in t.lower() or "counter" in t.lower()])


if __name__ == "__main__":
    main()
