#!/usr/bin/env python
"""Turns rocprofv3's rocpd sqlite output (<name>_results.db) into the plain-text summaries kept
under profiles/: per-kernel call count / total / average duration (kernel-trace --stats), and
per-kernel mean PMC counter values (--pmc passes).  Usage:
    rocprof_summary.py stats  <results.db>  > profiles/rNN_<tag>_kernel_stats.txt
    rocprof_summary.py pmc    <results.db>  > profiles/rNN_<tag>_pmc_<counter>.txt
"""
import sqlite3
import sys


def short(name):
    name = name.replace("np1k::", "").replace("np2::(anonymous namespace)::", "np2::").replace("(anonymous namespace)::", "")
    return name.split("(")[0][:60]


def stats(db):
    c = sqlite3.connect(db)
    print("%-62s %8s %14s %12s %7s" % ("kernel", "calls", "total_us", "avg_us", "pct"))
    for name, calls, total, avg, pct in c.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print("%-62s %8d %14.3f %12.3f %7.2f" % (short(name), calls, total / 1e0 if total < 1e9 else total, avg, pct))
    print("# durations in microseconds as reported by rocprofv3 (top_kernels view)")


def pmc(db):
    c = sqlite3.connect(db)
    q = ("select kernel_name, counter_name, count(*), avg(value), sum(value), avg(duration) from counters_collection "
         "group by kernel_name, counter_name order by sum(value) desc")
    print("%-62s %-14s %8s %16s %16s %12s" % ("kernel", "counter", "calls", "mean_value", "sum_value", "avg_ns"))
    for name, ctr, n, mean, tot, dur in c.execute(q):
        print("%-62s %-14s %8d %16.3f %16.3f %12.1f" % (short(name), ctr, n, mean, tot, dur))
    print("# FETCH_SIZE / WRITE_SIZE are in KiB per dispatch; gfx950: FETCH_SIZE under-reports wide coalesced reads by 2x "
          "(MI355X_MICROARCH.md, HBM section)")


if __name__ == "__main__":
    {"stats": stats, "pmc": pmc}[sys.argv[1]](sys.argv[2])
