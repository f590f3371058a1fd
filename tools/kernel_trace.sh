cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r01e
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1 || exit 1
python3 $R/tools/step_timeline.py $O/kt 3 > $O/step_timeline.txt 2>&1
rm -f $O/kt/kt_kernel_trace.csv
