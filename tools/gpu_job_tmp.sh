#!/bin/bash
mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -6 > $O/tests.txt
cat $O/tests.txt
