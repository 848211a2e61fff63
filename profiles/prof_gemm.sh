# usage: ADX_GEMM_SHAPE=M,N,K [ADX_GEMM_STAGED=1] bash profiles/prof_gemm.sh <tag> <precision>   -- kernel-trace summary of profiles/gemm_times.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 profiles/gemm_times.py "$@" > gpurun_out/prof_$tag.log 2>&1 || exit 1
tail -2 gpurun_out/prof_$tag.log
t=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 6; find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
