# developer timing: k_step_head's duration in the step with each block range switched off in turn (results are wrong then)
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for k in 0 1 2 4 8 16 32 64 126 127; do
  O=$R/gpurun_out/head_$k
  mkdir -p $O
  CINEMA_HEAD_SKIP=$k timeout -s KILL 200 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 10 --warmup 3 --profile-step > $O/kt.log 2>&1 || { tail -20 $O/kt.log; exit 3; }
  echo "skip=$k $(grep k_step_head $O/kt/kt_kernel_stats.csv | cut -d, -f2-4) $(grep -m1 '\[head\]' $O/kt.log)"
  rm -rf $O/kt
done
