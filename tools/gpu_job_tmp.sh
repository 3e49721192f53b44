#!/bin/bash
mkdir -p gpurun_out/$1; O=gpurun_out/$1
timeout 2400 python -m pytest tests -q -x -m gpu 2>&1 | tail -4 > $O/tests.txt
cat $O/tests.txt
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
