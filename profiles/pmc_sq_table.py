"""SQ counters per dispatch from rocprofv3 --pmc passes over profiles/pmc_targets.py (8 SQ slots per pass on gfx950,
MI355X_MICROARCH.md "rocprofv3 PMC slots").
usage: python profiles/pmc_sq_table.py pass1_results.db [pass2_results.db ...] -- kernel_substring [...]
Prints, for every kernel whose name contains one of the substrings, the average of every collected counter over its
dispatches (whole device), and the wave-cycle split ACTIVE / WAIT_ANY / WAIT_INST_ANY in percent."""
import collections
import os
import re
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import phases as _phases  # noqa: E402


def read(db):
    con = sqlite3.connect(db)
    cur = con.cursor()
    view = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")
            if r[0].startswith("counters_collection")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    order = " order by dispatch_id" if "dispatch_id" in cols else ""
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, cname, value in cur.execute(f"select kernel_name, counter_name, value from {view}{order}"):
        acc[name][cname].append(value)
    # the renderers run in phases in pmc_targets.py (profiles/phases.py): split by dispatch order
    out = collections.defaultdict(dict)
    for name, counters in acc.items():
        for c, vals in counters.items():
            for tag, part in _phases.split(name, vals):
                out[name + tag][c] = part
    return out


def short(name):
    name, _, phase = name.partition("@")
    name = re.sub(r"genre::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0] + ("@" + phase if phase else "")


sep = sys.argv.index("--")
dbs, pats = sys.argv[1:sep], sys.argv[sep + 1:]
merged = collections.defaultdict(dict)
for db in dbs:
    for k, counters in read(db).items():
        for c, vals in counters.items():
            if vals:
                merged[k][c] = (sum(vals) / len(vals), len(vals))
for k in sorted(merged, key=short):
    if not any(p in k for p in pats):
        continue
    c = merged[k]
    if not c:
        continue
    print("# %s   (%d dispatches)" % (short(k), max(n for _, n in c.values())))
    wc = c.get("SQ_WAVE_CYCLES", (0, 0))[0]
    for name in sorted(c):
        v = c[name][0]
        note = ""
        if wc and name in ("SQ_ACTIVE_INST_ANY", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY"):
            note = "   (%.0f %% of wave cycles)" % (100.0 * v / wc)
        print("%-22s %14.0f%s" % (name, v, note))
    print("#")
