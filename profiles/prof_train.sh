# usage: bash profiles/prof_train.sh <tag>   -- timing line + kernel-trace summary of the refiner's training step
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
tag=$1
timeout -k 10 200 python3 profiles/train_step.py --cpu > gpurun_out/${tag}_train_step.txt 2> gpurun_out/${tag}_train_step.err || exit 1
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_$tag -- python3 profiles/train_step.py --only fs10 --steps 5 > gpurun_out/prof_$tag.log 2>&1 || exit 1
t=$(find gpurun_out/prof_$tag -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 30 > gpurun_out/${tag}_train_by_grid.txt
cp $(find gpurun_out/prof_$tag -name "*kernel_stats.csv" | head -1) gpurun_out/${tag}_train_kernel_stats.csv
find gpurun_out/prof_$tag -name "*kernel_trace.csv" -delete
