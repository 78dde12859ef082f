cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for v in one two; do
  if [ $v = two ]; then export TAC_N4096_TWO_AREAS=1; else unset TAC_N4096_TWO_AREAS; fi
  out=gpurun_out/pmc_4096_$v
  mkdir -p $out
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $out/sq -o p -- python tools/prof_driver.py spec4096 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/stall -o p -- python tools/prof_driver.py spec4096 3 > /dev/null 2>&1
  echo "== $v"; python tools/pmc_summary.py $out
done
