#!/bin/bash
mkdir -p gpurun_out/$1; O=gpurun_out/$1
CURIOUS_LIB=abtest/dwst.so timeout 300 python tools/dw_timeline.py 19 2>&1 | grep -v amdgpu.ids | cut -c1-700 > $O/tl16.txt
cat $O/tl16.txt
timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_round6.py -q -x -k "rank or sixteen or 64x64" 2>&1 | tail -3
