#!/bin/bash
# Run ON THE GPU BOX: counters of the |X|^2-row kernel in its lab forms (TAC_R3_FORM; separate --pmc passes).
set -u
export TMPDIR=/tmp
for form in base h12 h12b; do
  out=gpurun_out/pmc_spec_$form
  mkdir -p $out
  if [ $form = base ]; then unset TAC_R3_FORM; else export TAC_R3_FORM=$form; fi
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python tools/prof_driver.py spec 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/stall -o p -- python tools/prof_driver.py spec 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC --output-format csv -d $out/mem -o p -- python tools/prof_driver.py spec 3 > /dev/null 2>&1
  echo "== $form"; python tools/pmc_summary.py $out | grep -A30 stft_ring3
done
