#!/bin/bash
# A/B builds of libgenre_hip.so for tools/ab_round2.py: ONE source of csrc/ recompiled with extra -D macros, the other
# objects taken from the regular build.  Libraries land in tools/variants/ (not tracked; they travel with gpurun).
#   usage: tools/build_variants.sh sph_render_bm.hip name:"-DFOO=1" other:"-DFOO=2" ...
set -e
SRC="$1"; shift
cd "$(dirname "$0")/../genre-shapehd_amd/csrc"
FLAGS="-O3 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -fno-gpu-flush-denormals-to-zero -I../../include"
make -s
mkdir -p ../../tools/variants
OBJS=""
for o in api cam_bp calc_prob nnd nnd_host glue sph_render sph_render_seg sph_render_bm; do
  [ "$o.hip" = "$SRC" ] || OBJS="$OBJS $o.o"
done
for spec in "$@"; do
  name="${spec%%:*}"; defs="${spec#*:}"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 $FLAGS $defs -c "$SRC" -o /tmp/variant_$name.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -pthread -o ../../tools/variants/libgenre_hip_$name.so $OBJS /tmp/variant_$name.o
done
ls -la ../../tools/variants/
