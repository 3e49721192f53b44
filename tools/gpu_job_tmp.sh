#!/bin/bash
timeout 900 python -m pytest tests/test_gpu_round6.py -q -x -k "several_rank_paths" 2>&1 | tail -15
