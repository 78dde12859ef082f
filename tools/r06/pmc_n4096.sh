#!/bin/bash
# Run ON THE GPU BOX: counters of the fft_length-4096 |X| rows (cfg-4 slice) for library builds under gpurun_variants/ (separate --pmc passes).
#   [WHAT=mel4096] bash tools/r06/pmc_n4096.sh name [name ...]     ('default' = the shipped library; WHAT: prof_driver.py's case)
set -u
what=${WHAT:-spec4096}
export TMPDIR=/tmp
for v in "$@"; do
  out=gpurun_out/pmc_${what}_$v
  mkdir -p $out
  if [ $v = default ]; then unset TAC_AMD_LIB; else export TAC_AMD_LIB=$PWD/gpurun_variants/libtac_$v.so; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python tools/prof_driver.py $what 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/stall -o p -- python tools/prof_driver.py $what 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC SQ_BUSY_CU_CYCLES --output-format csv -d $out/mem -o p -- python tools/prof_driver.py $what 3 > /dev/null 2>&1
  echo "== $v"; python tools/pmc_summary.py $out | grep -A30 n4096
done
