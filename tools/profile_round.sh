# usage (through gpurun): bash tools/profile_round.sh <tag>    e.g. r03
# rocprofv3 passes of `bench.py --profile-step` (the timed steps only): kernel trace + stats, then one --pmc pass per
# counter group (separate runs, as the MI355X guide prescribes).  Leaves under gpurun_out/<tag>/: step_timeline.txt,
# kernel_stats.csv (EVERY row of the stats whose kernel is a library kernel `k_*`, full names), pmc_{fetch,write,tcc,sq}.txt,
# bench_train.json (the default bench line of the same tree).  Copy what is to be judged into profiles/<tag>_*.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-r03}
mkdir -p $O
timeout -s KILL 300 rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 $R/bench.py --steps 20 --warmup 5 --profile-step > $O/kt.log 2>&1 || exit 1
python3 $R/tools/step_timeline.py $O/kt 3 > $O/step_timeline.txt 2>&1
timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --profile-step > $O/fetch.log 2>&1 || exit 2
timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --profile-step > $O/write.log 2>&1 || exit 3
timeout -s KILL 200 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum -d $O/tcc -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --profile-step > $O/tcc.log 2>&1 || exit 4
timeout -s KILL 200 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS -d $O/sq -o p --output-format csv -- python3 $R/bench.py --steps 3 --warmup 2 --profile-step > $O/sq.log 2>&1 || exit 5
for p in fetch write tcc sq; do python3 $R/tools/pmc_summary.py $O/$p > $O/pmc_$p.txt 2>&1; rm -rf $O/$p; done
rm -f $O/kt/kt_kernel_trace.csv
python3 - $O/kt/kt_kernel_stats.csv > $O/kernel_stats.csv <<'PY'
import csv, sys
rows = list(csv.reader(open(sys.argv[1])))
w = csv.writer(sys.stdout)
w.writerow(rows[0])
for r in rows[1:]:
    name = r[0].replace("(anonymous namespace)::", "")
    name = name[5:] if name.startswith("void ") else name
    if name.startswith("k_"):   # every kernel of libcinema_hip.so is named k_*
        w.writerow(r)
PY
(cd $R && timeout -k 10 280 python3 bench.py > $O/bench_train.json 2> $O/bench_train.log) || exit 6
ls -la $O
