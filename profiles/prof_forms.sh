# usage: ADX_GEMM_SHAPE=M,N,K ADX_FORMS=1,2 bash profiles/prof_forms.sh <tag> <precision>   -- kernel-trace summary of profiles/gemm_forms.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; shift
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 profiles/gemm_forms.py "$@" > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }
grep -v amdgpu.ids gpurun_out/prof_$tag.log | tail -3
t=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 8 | tee gpurun_out/prof_$tag.txt; find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
