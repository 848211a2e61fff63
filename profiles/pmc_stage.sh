# usage: [ADX_...=..] bash profiles/pmc_stage.sh <tag> "<counters>" [kernel-name filters...]
# One counter pass over profiles/stage_times.py (every stage of a variant-A step launched in isolation).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctr=$2; shift; shift
timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -- python3 profiles/stage_times.py ${ADX_STAGE_PREC:-bf16x3} > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
t=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" "$@" > gpurun_out/pmc_$tag.txt; cat gpurun_out/pmc_$tag.txt; rm -f "$t"
