#!/bin/bash
# Run ON THE GPU BOX (via gpurun): rocprofv3 kernel trace of the bench command + separate PMC passes for the
# dominant kernels.  Writes under gpurun_out/$1/; tools/summarize_profiles.py turns it into profiles/$1/.
set -u
tag=${1:-r04}
out=gpurun_out/$tag
export TMPDIR=/tmp
mkdir -p $out
cd ${GRAFT_REPO_ROOT:-.}
python bench.py > $out/bench_N1.json 2> $out/bench_N1.err
# the headline steps alone (the stages launch the same kernel on other sizes: their launches would dilute its average) ...
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-stages --no-live-counters > $out/kt_bench.log 2>&1
# ... and the same command with its stages (every other kernel the bench line quotes)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_stages -o bench -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-live-counters > $out/kt_stages.log 2>&1
for k in mel stft spec spec4096 mel4096; do
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${k}_fetch -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${k}_write -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_${k}_sq -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_${k}_mfma -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
  rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA --output-format csv -d $out/pmc_${k}_stall -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
done
# the filterbank stage on the matrix cores (dense bank): MFMA instruction / busy counters of gemm_fb_kernel
rocprofv3 --pmc SQ_INSTS_VALU_MFMA_F32 SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_fb_mfma -o p -- python tools/prof_driver.py fb 3 > /dev/null 2>&1
# backward of the fused chain: which kernels, how long
# (120 back-to-back training steps each: steady-state clocks, not the five cold launches of round 2)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_grad -o grad -- python tools/prof_driver.py grad 120 > $out/kt_grad.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_gradf -o gradf -- python tools/prof_driver.py gradf 120 > $out/kt_gradf.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_grad400 -o grad400 -- python tools/prof_driver.py grad400h160 60 > $out/kt_grad400.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_gradspec -o gradspec -- python tools/prof_driver.py gradspec 60 > $out/kt_gradspec.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_grad1024 -o grad1024 -- python tools/prof_driver.py grad1024 60 > $out/kt_grad1024.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_grad512 -o grad512 -- python tools/prof_driver.py grad512 60 > $out/kt_grad512.log 2>&1
# counters of the backward kernels (separate passes, like the forward kernels')
k=grad
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_${k}_fetch -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/pmc_${k}_write -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $out/pmc_${k}_sq -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
rocprofv3 --pmc SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $out/pmc_${k}_stall -o p -- python tools/prof_driver.py $k 3 > /dev/null 2>&1
# steady-state timings of everything else
python tools/r06/mel4096_pack_info.py > $out/mel4096_banks.txt 2>&1
python tools/time_steady.py stft spec mel stft4096 spec4096 mel4096 stft512 spec512 stft1024 spec1024 mel512 mel1024 mel400 stft400 spec400 mel256 stft256 spec256 > $out/time_steady.txt 2>&1
python tools/time_others.py > $out/time_others.txt 2>&1
ls $out
