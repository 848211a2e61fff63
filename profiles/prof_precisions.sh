cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for p in bf16x3 bf16 fp32; do
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$p -- python3 bench.py --steps 1 --warmup 1 --no-cpu-baseline --precision $p > gpurun_out/prof_$p.log 2>&1 || exit 1
tail -1 gpurun_out/prof_$p.log | cut -c1-140
t=$(find gpurun_out/prof_$p -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 5; find gpurun_out/prof_$p -name "*kernel_trace.csv" -delete
done
