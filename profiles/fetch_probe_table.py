"""FETCH_SIZE per probe of tools/fetch_size_bench (each reads 256 MiB exactly once) from `rocprofv3 --pmc FETCH_SIZE`.
usage: python profiles/fetch_probe_table.py fetch_results.db"""
import collections
import sqlite3
import sys

con = sqlite3.connect(sys.argv[1])
cur = con.cursor()
view = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")
        if r[0].startswith("counters_collection")][0]
acc = collections.defaultdict(list)
for name, cname, value in cur.execute(f"select kernel_name, counter_name, value from {view}"):
    if cname == "FETCH_SIZE" and "probe_" in name:
        acc[name.split("(")[0]].append(value)
print("%-18s %10s %16s %12s" % ("probe", "dispatches", "FETCH_SIZE_KB", "of 262144"))
for k in sorted(acc):
    v = acc[k]
    print("%-18s %10d %16.0f %12.3f   (each: %s)" % (k, len(v), sum(v) / len(v), sum(v) / len(v) / 262144.0,
                                                      ", ".join("%.0f" % x for x in v)))
