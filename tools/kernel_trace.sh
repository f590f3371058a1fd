# usage (on the GPU box): bash tools/kernel_trace.sh <tag> [extra bench args]: kernel trace of the timed steps + one step's timeline
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-kt}
shift
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-step "$@" > $O/kt.log 2>&1 || exit 1
python3 $R/tools/step_timeline.py $O/kt 3 > $O/step_timeline.txt 2>&1
rm -f $O/kt/kt_kernel_trace.csv
cat $O/step_timeline.txt
