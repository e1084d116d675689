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
    c = sqlite3.connect(db)
    print("%-44s %6s %14s %14s %8s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, tot, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-44s %6d %14.3f %14.3f %8.3f" % (short(name)[:44], calls, tot / 1e3, avg / 1e3, pct))


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


if __name__ == "__main__":
    if sys.argv[1] == "--pmc":
        pmc(sys.argv[2])
    else:
        stats(sys.argv[1])
