#!/bin/bash
# Run ON THE GPU BOX: the stall-bucket and LDS counters of the fused mel kernel (separate --pmc passes).
set -u
out=gpurun_out/${1:-pmc_mel}
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python tools/prof_driver.py mel 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/stall -o p -- python tools/prof_driver.py mel 3 > /dev/null 2>&1
python tools/pmc_summary.py $out
