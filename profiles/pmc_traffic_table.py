"""FETCH_SIZE / WRITE_SIZE per dispatch from two rocprofv3 --pmc passes over profiles/pmc_targets.py.
usage: python profiles/pmc_traffic_table.py fetch_results.db write_results.db [out.json batch]
With out.json: also writes the table bench.py reads for roofline.traffic -- per kernel the corrected HBM bytes per
launch, stamped with the sha256 of the kernel sources it was measured on (bench.py: source_sha) and the batch size.
gfx950 correction: FETCH_SIZE reports HALF the bytes read, whatever the access shape -- settled in round 4 with
tools/fetch_size_bench.hip, four kernels that read the same 256 MiB exactly once (profiles/r04b_fetch_size_probe.txt):
    16 B / lane consecutive 0.500, 4 B / lane consecutive 0.500, scattered 128-byte lines read 4 B / lane 0.500, the same lines
    read 16 B / lane 0.500 of the true byte count
-- so it is doubled for EVERY kernel (rounds 1-3 doubled it only for the wide streams of WIDE_STREAMS below and under-counted
the reads of the gather-type kernels by half); units are KB."""
import collections
import json
import os
import re
import sqlite3
import sys

# (kept for the record: the kernels rounds 1-3 doubled; everything else was taken at face value then)
WIDE_STREAMS = ("fill2_vec4_kernel", "stop_fwd_vec4_kernel", "stop_bwd_vec4_kernel", "render_scan_fwd_kernel",
                "render_scan_bwd_kernel", "bm_sample_kernel")
FETCH_FACTOR = 2


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import phases as _phases  # noqa: E402  (how pmc_targets.py orders the renderers' launches: name@genre / @dense / @soft rows)


def per_kernel(db, counter):
    con = sqlite3.connect(db)
    cur = con.cursor()
    view = [r[0] for r in cur.execute("select name from sqlite_master where type in ('table','view')")
            if r[0].startswith("counters_collection")][0]
    cols = [r[1] for r in cur.execute(f"pragma table_info({view})")]
    order = " order by dispatch_id" if "dispatch_id" in cols else ""
    acc = collections.defaultdict(list)
    for name, cname, value in cur.execute(f"select kernel_name, counter_name, value from {view}{order}"):
        if cname == counter:
            acc[name].append(value)
    out = {}
    for name, vals in acc.items():
        for tag, part in _phases.split(name, vals):
            out[name + tag] = part
    return out


def short(name):
    name, _, phase = name.partition("@")
    name = re.sub(r"genre::\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return name.split("(")[0] + ("@" + phase if phase else "")


fetch = per_kernel(sys.argv[1], "FETCH_SIZE")
write = per_kernel(sys.argv[2], "WRITE_SIZE")
print("%-52s %10s %14s %14s %6s %12s" % ("kernel", "dispatches", "FETCH_SIZE_KB", "WRITE_SIZE_KB", "xF", "HBM_MB"))
table = {}
for k in sorted(set(fetch) | set(write), key=short):
    if "genre" not in k:
        continue
    f = sum(fetch.get(k, [0])) / max(1, len(fetch.get(k, [0])))
    w = sum(write.get(k, [0])) / max(1, len(write.get(k, [0])))
    factor = FETCH_FACTOR
    hbm = (factor * f + w) * 1024
    table[short(k)] = dict(dispatches=len(fetch.get(k, [])), fetch_kb=f, write_kb=w, fetch_factor=factor, hbm_bytes=hbm)
    print("%-52s %10d %14.0f %14.0f %6d %12.1f" % (short(k)[:52], len(fetch.get(k, [])), f, w, factor, hbm / 1e6))
if len(sys.argv) > 3:
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    with open(sys.argv[3], "w") as fh:
        csrc = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "genre-shapehd_amd", "csrc")
        names = sorted(n for n in os.listdir(csrc) if n.endswith((".hip", ".hpp")))
        json.dump(dict(source_sha=bench.source_sha(), source_sha_by_file=bench.source_sha(names), batch=int(sys.argv[4]), kernels=table,
                       how="rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over profiles/pmc_targets.py"),
                  fh, indent=1, sort_keys=True)
