#!/usr/bin/env python
"""Per-dispatch kernel durations of a rocprofv3 rocpd database, in launch order (name filter optional):
    python tools/rocpd_dispatches.py <results.db> [substring] [max_rows]"""
import sqlite3
import sys


def main(path, sub="", limit=200):
    c = sqlite3.connect(path)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    start = "start" if "start" in cols else "start_timestamp"
    end = "end" if "end" in cols else "end_timestamp"
    q = f"select name, {start}, {end} from kernels order by {start}"
    n = 0
    for name, s, e in c.execute(q):
        if sub and sub not in name:
            continue
        print(f"{(e - s) / 1e3:10.1f} us  {name[:110]}")
        n += 1
        if n >= limit:
            break


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "", int(sys.argv[3]) if len(sys.argv) > 3 else 200)
