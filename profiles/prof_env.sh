# usage: VAR=value bash profiles/prof_env.sh <tag> [bench args]   -- like prof_one.sh, the caller's environment reaches the run
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log | cut -c1-200; exit 1; }
tail -1 gpurun_out/prof_$tag.log | cut -c1-140
t=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 8; find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
