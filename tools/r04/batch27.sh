#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out/r04
python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > gpurun_out/r04/batch27_tests.txt
python tools/time_others.py > gpurun_out/r04/batch27_time_others.txt 2>&1
cat gpurun_out/r04/batch27_tests.txt gpurun_out/r04/batch27_time_others.txt
