#!/bin/bash
# registers / LDS / scratch of every kernel of one csrc/ source, from the compiler's own remarks
#   usage: tools/kernel_resources.sh sph_render_bm.hip [extra -D flags]
cd "$(dirname "$0")/../genre-shapehd_amd/csrc"
SRC=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-gpu-flush-denormals-to-zero -I../../include "$@" -Rpass-analysis=kernel-resource-usage -c "$SRC" -o /tmp/kres.o 2>&1 |
python3 -c '
import sys, re, subprocess
cur = None
for line in sys.stdin:
    m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
    if not m: continue
    t = m.group(1).strip()
    if t.startswith("Function Name:"):
        name = t.split(":", 1)[1].strip()
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
        name = re.sub(r"\(.*", "", name).replace("genre::(anonymous namespace)::", "").replace("void ", "")
        cur = {"name": name}
    elif cur is not None:
        k, v = t.split(":", 1)
        cur[k.strip()] = v.strip()
        if k.strip().startswith("LDS Size"):
            print("%-58s vgpr %3s agpr %3s sgpr %3s spill v %s s %s scratch %s occ %s lds %s" % (cur["name"][:58], cur.get("VGPRs"), cur.get("AGPRs"), cur.get("TotalSGPRs"), cur.get("VGPR Spill"), cur.get("SGPR Spill"), cur.get("ScratchSize [bytes/lane]"), cur.get("Occupancy [waves/SIMD]"), cur.get("LDS Size [bytes/block]")))
            cur = None
'
