#!/usr/bin/env python3
"""Summarise rocprofv3 .db outputs (ROCm 7.2 writes sqlite by default) into small text files for profiles/.

    python tools/rocprof_summary.py gpurun_out/prof_r1/trace_results.db            # kernel-trace --stats summary
    python tools/rocprof_summary.py --pmc gpurun_out/pmc_r1_sq/pmc_results.db      # counters per kernel
"""
import sqlite3
import sys
from collections import defaultdict


def short(name):
    name = name.replace("void ", "")
    return name.split("(")[0]


def stats(db):
    """rocprofv3 --kernel-trace --stats: calls, total, average and share as rocprofv3 reports them (top_kernels view), plus
    the median and minimum duration per kernel from the trace itself (the first search launch of a run is un-seeded and
    slower, so the median is the figure to hold against bench.py's hipEvent average over the timed, seeded launches)."""
    c = sqlite3.connect(db)
    durs = defaultdict(list)
    for name, s, e in c.execute("select name,start,end from kernels"):
        durs[short(name)[:44]].append(e - s)
    print("%-44s %6s %14s %14s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "median_us", "min_us", "pct"))
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        k = short(name)[:44]
        d = sorted(durs.get(k, [0]))
        # top_kernels reports microseconds, the trace nanoseconds
        print("%-44s %6d %14.3f %14.3f %14.3f %14.3f %8.3f" % (k, calls, tot, avg, d[len(d) // 2] / 1e3, d[0] / 1e3, pct))


def pmc(db):
    c = sqlite3.connect(db)
    acc = defaultdict(lambda: defaultdict(float))
    disp = defaultdict(set)
    dur = defaultdict(dict)
    for kn, did, cn, val, d in c.execute("select kernel_name,dispatch_id,counter_name,value,duration from counters_collection"):
        k = short(kn)
        acc[k][cn] += val
        disp[k].add(did)
        dur[k][did] = d
    for k in acc:
        n = len(disp[k])
        print("%s  dispatches=%d  avg_duration_us=%.3f" % (k, n, sum(dur[k].values()) / n / 1e3))
        for cn in sorted(acc[k]):
            print("    %-28s %22.1f  (per dispatch %.1f)" % (cn, acc[k][cn], acc[k][cn] / n))


def timeline(db, skip=0.5):
    """Average kernel duration and the gap before each kernel, over the steady-state part of the trace."""
    c = sqlite3.connect(db)
    cols = [r[1] for r in c.execute("pragma table_info(kernels)")]
    if not cols:
        print("views:", [r[0] for r in c.execute("select name from sqlite_master")])
        return
    rows = list(c.execute("select name,start,end from kernels order by start"))
    rows = rows[int(len(rows) * skip):]
    dur = defaultdict(list)
    gap = defaultdict(list)
    for i, (name, s, e) in enumerate(rows):
        k = short(name)[:44]
        dur[k].append(e - s)
        if i:
            gap[k].append(s - rows[i - 1][2])
    print("%-44s %6s %10s %14s" % ("kernel", "calls", "avg_us", "gap_before_us"))
    for k in dur:
        print("%-44s %6d %10.2f %14.2f" % (k, len(dur[k]), sum(dur[k]) / len(dur[k]) / 1e3, sum(gap[k]) / max(1, len(gap[k])) / 1e3))
    print("span_us %.1f  kernels %d" % ((rows[-1][2] - rows[0][1]) / 1e3, len(rows)))


if __name__ == "__main__":
    if sys.argv[1] == "--timeline":
        timeline(sys.argv[2])
    elif sys.argv[1] == "--pmc":
        pmc(sys.argv[2])
    else:
        stats(sys.argv[1])
