# usage (through gpurun): bash tools/gpu_check.sh <tag> [tests|bench|trace ...]   -- every step under its own time limit
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$1
shift
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for what in "$@"; do
  case $what in
    tests) (cd $R && timeout -k 10 500 python3 -u -m pytest tests -m gpu -x -q --timeout 120 --timeout-method thread > $O/test.log 2>&1) || { tail -30 $O/test.log; exit 1; }
           tail -2 $O/test.log ;;
    bench) (cd $R && timeout -k 10 200 python3 bench.py > $O/bench.log 2>&1) || { tail -30 $O/bench.log; exit 2; }
           tail -1 $O/bench.log ;;
    trace) timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $O/kt.log 2>&1 || { tail -30 $O/kt.log; exit 3; }
           python3 $R/tools/step_timeline.py $O/kt 3 > $O/step_timeline.txt 2>&1
           rm -f $O/kt/kt_kernel_trace.csv
           cat $O/step_timeline.txt ;;
  esac
done
