#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc run (rocpd database): counters per kernel name, summed over dispatches."""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
if 'counters_collection' in tabs:
    cols = [r[1] for r in db.execute("pragma table_info(counters_collection)")]
    q = "select kernel_name, counter_name, sum(value), count(distinct dispatch_id) from counters_collection group by kernel_name, counter_name"
    agg = {}
    for k, c, v, n in db.execute(q):
        agg.setdefault(k.split('(')[0], {})[c] = (v, n)
    for k, d in agg.items():
        if len(sys.argv) > 2 and sys.argv[2] not in k: continue
        print(k, {c: "%.4g" % v for c, (v, n) in d.items()}, "dispatches", max(n for _, n in d.values()))
else:
    print(tabs)
