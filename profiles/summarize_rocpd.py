#!/usr/bin/env python3
"""Turn a rocprofv3 rocpd database (`rocprofv3 --kernel-trace --stats -d DIR -o NAME -- cmd`
writes DIR/NAME_results.db on ROCm 7.2) into the per-kernel summary table committed under
profiles/.  Usage: python profiles/summarize_rocpd.py gpurun_out/prof/x_results.db > profiles/x.txt"""
import sqlite3
import sys


import os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import phases as _phases  # noqa: E402  (how pmc_targets.py orders the renderers' launches: name@genre / @dense / @soft rows)


def phase_rows(cur):
    per = {}
    for name, start, end, vg, sg, lds in cur.execute("select name, start, end, vgpr_count, sgpr_count, lds_size from kernels order by start"):
        per.setdefault(name, []).append(((end - start) / 1e3, vg, sg, lds))
    rows = []
    for name, d in per.items():
        parts = [(tag + " " if tag else "", dd) for tag, dd in _phases.split(name, d)]
        for tag, dd in parts:
            us = [x[0] for x in dd]
            rows.append((tag + name, len(us), sum(us), sum(us) / len(us), min(us), max(us), max(x[1] for x in dd),
                         max(x[2] for x in dd), max(x[3] for x in dd)))
    return sorted(rows, key=lambda r: -r[2])


def main(path, top=45, phases=False):
    cur = sqlite3.connect(path).cursor()
    rows = phase_rows(cur) if phases else cur.execute(
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
    main([a for a in sys.argv[1:] if a != "--phases"][0], phases="--phases" in sys.argv)
