# usage: ADX_GEMM_SHAPE=M,N,K bash profiles/pmc_gemm.sh <tag> "<counters>" <precision>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctr=$2; prec=$3
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -- python3 profiles/gemm_times.py $prec > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
t=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" linear_big linear_ws linear_rows linear_glds > gpurun_out/pmc_$tag.txt; cat gpurun_out/pmc_$tag.txt; rm -f "$t"
