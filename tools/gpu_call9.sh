cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/grad
timeout 900 python -m pytest tests -m gpu -x -q -k "grad or autograd or backward or strict" 2>&1 | tail -8
cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_grad -o grad -- python $GRAFT_REPO_ROOT/tools/prof_driver.py grad 60 > /tmp/kt.log 2>&1
f=$(find /tmp/kt_grad -name "*kernel_stats.csv" | head -1)
cp $f $GRAFT_REPO_ROOT/gpurun_out/grad/kernel_stats_grad.csv
head -12 $f | cut -c1-150
