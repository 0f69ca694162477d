#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db on ROCm 7.2) into the per-kernel summary table committed under
profiles/.  Usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/x.txt"""
import sqlite3
import sys


def main(path, top=45):
    cur = sqlite3.connect(path).cursor()
    rows = cur.execute(
        "select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, "
        "max(vgpr_count), max(sgpr_count), max(lds_size) from kernels group by name order by 3 desc").fetchall()
    tot = sum(r[2] for r in rows)
    print("# rocprofv3 --kernel-trace --stats summary of %s" % path)
    print("# total kernel time %.1f us over %d dispatches" % (tot, sum(r[1] for r in rows)))
    print("%-96s %7s %12s %10s %10s %10s %6s %5s %5s %6s" % ("kernel", "calls", "total_us", "avg_us", "min_us",
                                                            "max_us", "pct", "vgpr", "sgpr", "lds"))
    for r in rows[:top]:
        print("%-96s %7d %12.1f %10.2f %10.2f %10.2f %6.2f %5d %5d %6d" % (
            r[0][:96], r[1], r[2], r[3], r[4], r[5], 100 * r[2] / tot, r[6], r[7], r[8]))


if __name__ == "__main__":
    main(sys.argv[1])
