cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
cd /tmp && export TMPDIR=/tmp
for k in grad400h160; do
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$k -o $k -- python $GRAFT_REPO_ROOT/tools/prof_driver.py $k 60 > /tmp/kt.log 2>&1
f=$(find /tmp/kt_$k -name "*kernel_stats.csv" | head -1)
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/grad; cp $f $GRAFT_REPO_ROOT/gpurun_out/grad/kernel_stats_$k.csv
done
