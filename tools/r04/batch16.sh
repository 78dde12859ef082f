#!/bin/bash
# round 4, GPU batch 16: host overhead with the bound-argument launcher; whole GPU suite on the new defaults (swizzled exchange)
cd $GRAFT_REPO_ROOT
out=gpurun_out/r04_batch16; mkdir -p $out
timeout 200 python tools/host_overhead.py > $out/host_overhead.txt 2>&1
timeout 200 python tools/host_profile.py > $out/host_profile.txt 2>&1
timeout 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3 > $out/pytest_default.txt
cat $out/host_overhead.txt; head -34 $out/host_profile.txt | cut -c1-160; cat $out/pytest_default.txt
