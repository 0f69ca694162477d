#!/bin/bash
# the standard-layout renderer's GPU tests under other table shapes: rows of the backward so short that most bricks are split over
# several (atomics on every voxel, zeroing by the per-ray pass), then no split brick at all with short forward rows
set -u
SEL='tests/test_gpu_render_seg.py tests/test_gpu_render.py tests/test_gpu_render_genre.py -q -m gpu -k "not bm and not minor"'
GENRE_TABLE_CACHE=0 GENRE_SEG_BWD_SPLIT=192 GENRE_SEG_BWD_SPLIT_SMALL=64 eval timeout 2400 python -m pytest $SEL 2>&1 | tail -2
GENRE_TABLE_CACHE=0 GENRE_SEG_BWD_SPLIT=1000000 GENRE_SEG_BWD_SPLIT_SMALL=1000000 GENRE_SEG_SPLIT=128 GENRE_SEG_SPLIT_SMALL=64 eval timeout 2400 python -m pytest $SEL 2>&1 | tail -2
# the whole GPU suite with the host-side shortcut off: the kernels' own zero paths (render_bm_backward / render_seg_backward on volumes the
# clamp blocks) are what runs in the GenRe-chain tests then
GENRE_LAZY_ZERO_GRAD=0 timeout 3000 python -m pytest tests -q -m gpu 2>&1 | tail -2
