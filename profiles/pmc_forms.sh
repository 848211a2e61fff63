# usage: ADX_GEMM_SHAPE=M,N,K ADX_FORMS=2 bash profiles/pmc_forms.sh <tag> "<counters>" <precision>   -- one --pmc pass over profiles/gemm_forms.py
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1; ctr=$2; prec=$3
timeout -s KILL 120 rocprofv3 --pmc $ctr --output-format csv -d gpurun_out/pmc_$tag -- python3 profiles/gemm_forms.py $prec > gpurun_out/pmc_$tag.log 2>&1 || { tail -5 gpurun_out/pmc_$tag.log; exit 1; }
t=$(find gpurun_out/pmc_$tag -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" linear_glds > gpurun_out/pmc_$tag.txt; cat gpurun_out/pmc_$tag.txt; rm -f "$t"
