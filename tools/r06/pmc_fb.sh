#!/bin/bash
# Run ON THE GPU BOX: counters of the dense fp32 MFMA filterbank GEMM (separate --pmc passes).  $1 = output name
set -u
out=gpurun_out/${1:-pmc_fb}
export TMPDIR=/tmp
mkdir -p $out
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python tools/prof_driver.py fb 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/stall -o p -- python tools/prof_driver.py fb 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_MISC SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM --output-format csv -d $out/mfma -o p -- python tools/prof_driver.py fb 3 > /dev/null 2>&1
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum --output-format csv -d $out/tcp -o p -- python tools/prof_driver.py fb 3 > /dev/null 2>&1
python tools/pmc_summary.py $out | grep -A40 gemm_fb
