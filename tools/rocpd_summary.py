#!/usr/bin/env python
"""Summarise a rocprofv3 rocpd database (<name>_results.db): per-kernel call count / total / average duration, and, if
the run collected PMC counters, the per-kernel average of each counter.  Writes CSV to stdout.

    python tools/rocpd_summary.py gpurun_out/prof1/r1_results.db > profiles/r01_bench_kernel_stats.csv
"""
import sqlite3
import sys


def short(name, n=140):
    """library kernels (rocPRIM, ATen) carry kilobyte-long template names"""
    return name if len(name) <= n else name[:n] + "..."


def main(path):
    c = sqlite3.connect(path)
    print("# kernel stats (durations in us)")
    print("name,calls,total_us,avg_us,pct")
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"\"{short(name)}\",{calls},{total:.3f},{avg:.3f},{pct:.3f}")
    try:
        rows = list(c.execute(
            "select k.name, p.counter_name, count(*), avg(p.value), sum(p.value) from counters_collection p "
            "join kernels k on k.dispatch_id = p.dispatch_id group by k.name, p.counter_name"))
    except sqlite3.Error:
        try:
            cur = c.execute("select * from counters_collection limit 1")
            cols = [d[0] for d in cur.description]
            print("# counters_collection columns:", cols)
            namecol = "kernel_name" if "kernel_name" in cols else ("name" if "name" in cols else None)
            rows = list(c.execute(f"select {namecol}, counter_name, count(*), avg(value), sum(value) from "
                                  f"counters_collection group by {namecol}, counter_name")) if namecol else []
        except sqlite3.Error as e:
            print("# no counters:", e)
            rows = []
    if rows:
        print("# PMC counters (per-dispatch average)")
        print("name,counter,dispatches,avg_value,sum_value")
        for name, cn, n, avg, tot in rows:
            print(f"\"{short(name)}\",{cn},{n},{avg:.3f},{tot:.3f}")


if __name__ == "__main__":
    main(sys.argv[1])
