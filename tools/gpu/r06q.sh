#!/bin/bash
python tools/debug_seg_bwd.py 2>&1 | grep -v amdgpu.ids
