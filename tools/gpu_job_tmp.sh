#!/bin/bash
mkdir -p gpurun_out/$1; O=$PWD/gpurun_out/$1
CURIOUS_LIB=abtest/dwst.so timeout 300 python tools/dw_timeline.py 1 2>&1 | grep -v amdgpu.ids | cut -c1-400 | head -12
CURIOUS_LIB=abtest/dwst.so timeout 300 python tools/dw_timeline.py 19 2>&1 | grep -v amdgpu.ids | cut -c1-400 | head -8
