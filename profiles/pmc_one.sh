# usage: bash profiles/pmc_one.sh <tag> "<counters>" [kernel-name filters...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
t=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" "$@" > gpurun_out/pmc_$tag.txt; cat gpurun_out/pmc_$tag.txt; rm -f "$t"
