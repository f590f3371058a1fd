# usage: bash tools/bg.sh '<gpurun command>' [gpurun timeout]  -- rebuild the library; only if that succeeds, run on the GPU box
cd /root/repo
python -c "from cinema_amd import build as b; b.build()" > /tmp/build.log 2>&1 || { grep -E "error" -A6 /tmp/build.log | head -40; echo BUILD FAILED; exit 1; }
/usr/local/graft/bin/gpurun --timeout ${2:-900} -- "$1"
