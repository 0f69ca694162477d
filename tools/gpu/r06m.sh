#!/bin/bash
set -u
echo wide; python tools/time_cam_bwd.py 2>&1 | grep -v amdgpu.ids
echo narrow; GENRE_CAM_BWD_NARROW=1 python tools/time_cam_bwd.py 2>&1 | grep -v amdgpu.ids
