#!/bin/bash
# host-side cost of one fused call: compiled binding vs ctypes, same box, alternating processes
export TMPDIR=/tmp
mkdir -p gpurun_out/r05
{
for i in 1 2; do
  echo "== compiled binding"; python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids
  echo "== ctypes (TAC_AMD_EXT=0)"; TAC_AMD_EXT=0 python tools/host_overhead.py 2>&1 | grep -v amdgpu.ids
done
python tools/host_profile.py 2>&1 | grep -v amdgpu.ids | head -40
} | tee gpurun_out/r05/host_overhead.txt
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "fused or g2 or deferred or tables or launch or graph" 2>&1 | tail -3
