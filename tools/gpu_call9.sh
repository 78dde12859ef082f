cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/grad
timeout 900 python -m pytest tests -m gpu -x -q -k "grad or autograd or backward or strict" 2>&1 | tail -4
cd /tmp && export TMPDIR=/tmp
for k in grad gradspec; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$k -o $k -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $k 60 > /tmp/kt.log 2>&1
f=$(find /tmp/kt_$k -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/grad/kernel_stats_$k.csv
done
TAC_BWD_LDS_RING=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_old -o old -- python $GRAFT_REPO_ROOT/tools/prof_driver.py gradspec 60 > /tmp/kt.log 2>&1
cp $(find /tmp/kt_old -name "*kernel_stats.csv" | head -1) $GRAFT_REPO_ROOT/gpurun_out/grad/kernel_stats_gradspec_ldsring.csv
