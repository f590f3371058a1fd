# usage (through gpurun): bash tools/pmc_bwd.sh <tag>   -- HBM-side bytes of the chain kernels in tools/time_bwd.py
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/${1:-pmcb}
mkdir -p $O
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE -d $O/fetch -o p --output-format csv -- python3 $R/tools/time_bwd.py > $O/fetch.log 2>&1 || exit 2
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE -d $O/write -o p --output-format csv -- python3 $R/tools/time_bwd.py > $O/write.log 2>&1 || exit 3
for p in fetch write; do python3 $R/tools/pmc_summary.py $O/$p > $O/pmc_$p.txt 2>&1; rm -rf $O/$p; done
cat $O/pmc_fetch.txt $O/pmc_write.txt
