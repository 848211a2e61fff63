# usage (on the GPU box, from the repository root): bash profiles/r03_evidence.sh <part>
# Collects what DESIGN.md quotes for this round under gpurun_out/r03/ (copied into profiles/ afterwards).  Counter passes
# run on their own (no tracing beside --pmc), one rocprofv3 process per pass.
set -u
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r03; mkdir -p $O
part=${1:-all}
trace() {   # trace <tag> <command...>: kernel-trace summary -> $O/<tag>_kernel_stats.csv
    tag=$1; shift
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$tag -- "$@" > $O/prof_$tag.log 2>&1 || { tail -5 $O/prof_$tag.log; return 1; }
    cp $(find $O/prof_$tag -name "*kernel_stats.csv" | head -1) $O/${tag}_kernel_stats.csv
    t=$(find $O/prof_$tag -name "*kernel_trace.csv" | head -1); python3 profiles/summarize_trace.py "$t" 12 > $O/${tag}_by_grid.txt; rm -rf $O/prof_$tag
    head -8 $O/${tag}_by_grid.txt
}
pmc() {     # pmc <tag> "<counters>" <filter> <command...>: one counter pass -> appended to $O/<tag>.txt
    tag=$1; ctr=$2; flt=$3; shift; shift; shift
    timeout -s KILL 240 rocprofv3 --pmc $ctr --output-format csv -d $O/pmc_tmp -- "$@" > $O/pmc_tmp.log 2>&1 || { tail -5 $O/pmc_tmp.log; return 1; }
    t=$(find $O/pmc_tmp -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" $flt >> $O/$tag.txt; rm -rf $O/pmc_tmp
}
if [ $part = all ] || [ $part = bench ]; then
    timeout -k 10 400 python3 bench.py --steps 20 --warmup 5 > $O/bench_b1024.json 2> $O/bench.err && cut -c1-400 $O/bench_b1024.json
    trace refiner_b1024 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    # the same command with every kernel of a step on one stream: the per-kernel averages the events inside bench.py report
    # (its roofline pass runs un-overlapped); in the overlapped trace above the kernels stretch because the tail shares the CUs
    ADX_NO_OVERLAP=1 trace refiner_b1024_one_stream python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
fi
if [ $part = all ] || [ $part = pmcref ]; then
    rm -f $O/pmc_refiner_b1024.txt
    echo "# rocprofv3 --pmc passes over profiles/stage_times.py bf16x3 1024 (B=1024: 828416 conditioning rows), per-launch averages" > $O/pmc_refiner_b1024.txt
    echo "# HBM bytes = TCC_EA0_RDREQ x 128 B (= FETCH_SIZE KB doubled, the gfx950 correction) + TCC_EA0_WRREQ x 64 B" >> $O/pmc_refiner_b1024.txt
    pmc pmc_refiner_b1024 "TCC_EA0_RDREQ_sum FETCH_SIZE" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py bf16x3 1024
    pmc pmc_refiner_b1024 "TCC_EA0_WRREQ_sum WRITE_SIZE" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py bf16x3 1024
    pmc pmc_refiner_b1024 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py bf16x3 1024
    cat $O/pmc_refiner_b1024.txt
fi
if [ $part = all ] || [ $part = gemm ]; then
    python3 profiles/gemm_forms.py bf16x3 bf16 2>&1 | grep -v amdgpu.ids > $O/gemm_times.txt; cat $O/gemm_times.txt
    python3 profiles/gemm_dispatch.py bf16x3 bf16 2>&1 | grep -v amdgpu.ids > $O/gemm_dispatch.txt
        export ADX_GEMM_SHAPE=20480,4096,1024
    ADX_FORMS=2 trace gemm_x3 python3 profiles/gemm_forms.py bf16x3
    ADX_FORMS=2 trace gemm_bf16 python3 profiles/gemm_forms.py bf16
    rm -f $O/pmc_gemm.txt
    echo "# rocprofv3 --pmc passes over profiles/gemm_forms.py (ADX_FORMS=2: the two-workgroups-per-CU staged kernel), M=20480 N=4096 K=1024, per-launch averages" > $O/pmc_gemm.txt
    for p in bf16x3 bf16; do
        echo "## $p" >> $O/pmc_gemm.txt
        ADX_FORMS=2 pmc pmc_gemm "SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_WAIT_INST_LDS" linear_glds python3 profiles/gemm_forms.py $p
        ADX_FORMS=2 pmc pmc_gemm "GRBM_GUI_ACTIVE TCC_HIT_sum TCC_MISS_sum" linear_glds python3 profiles/gemm_forms.py $p
    done
    unset ADX_GEMM_SHAPE
    cat $O/pmc_gemm.txt
fi
if [ $part = all ] || [ $part = dit ]; then
    for w in dit1024 dit256 mmdit256; do
        steps=""; [ $w = dit1024 ] && steps="--dit-steps 20"
        timeout -k 10 400 python3 bench.py --workload $w $steps --steps 2 --warmup 1 > $O/bench_$w.json 2>> $O/bench.err && cut -c1-300 $O/bench_$w.json
    done
    timeout -k 10 400 python3 bench.py --workload dit1024 --precision bf16 --dit-steps 20 --steps 2 --warmup 1 > $O/bench_dit1024_bf16.json 2>> $O/bench.err
    timeout -k 10 400 python3 bench.py --workload dit1024 --mc-num 16 --dit-steps 20 --steps 2 --warmup 1 > $O/bench_dit1024_mc16.json 2>> $O/bench.err && cut -c1-300 $O/bench_dit1024_mc16.json
    timeout -k 10 400 python3 bench.py --workload dit256 --mc-num 16 --steps 2 --warmup 1 > $O/bench_dit256_mc16.json 2>> $O/bench.err && cut -c1-300 $O/bench_dit256_mc16.json
    timeout -k 10 400 python3 bench.py --workload strdiffdit --steps 5 --warmup 2 > $O/bench_strdiffdit.json 2>> $O/bench.err && cut -c1-400 $O/bench_strdiffdit.json
    trace strdiffdit python3 bench.py --workload strdiffdit --steps 2 --warmup 1
    trace dit1024 python3 bench.py --workload dit1024 --dit-steps 10 --steps 1 --warmup 1
    trace dit256 python3 bench.py --workload dit256 --steps 1 --warmup 1
fi
if [ $part = all ] || [ $part = misc ]; then
    python3 profiles/fresh_tensor_probe.py 1024 2>&1 | grep -v amdgpu.ids > $O/fresh_tensor_probe.txt; cat $O/fresh_tensor_probe.txt
    for b in 256 512 2048 4096; do timeout -k 10 300 python3 bench.py --batch $b --steps 4 --warmup 2 --no-cpu-baseline --no-extras; done 2>/dev/null | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print(d[\"config\"][\"batch_per_gpu\"], d[\"value\"], d[\"ms_per_step\"])" > $O/batch_knee.txt; cat $O/batch_knee.txt
    for c in "--frame-stack 80 --n-sample 10 --batch 512" "--n-sample 100 --batch 1024"; do timeout -k 10 300 python3 bench.py $c --steps 4 --warmup 2 --no-cpu-baseline --no-extras; done 2>/dev/null > $O/bench_other_configs.json; cut -c1-330 $O/bench_other_configs.json
fi
if [ $part = all ] || [ $part = train ]; then
    timeout -k 10 300 python3 profiles/train_step.py --cpu > $O/train_step.txt 2> $O/train_step.err && cat $O/train_step.txt
    trace train_step python3 profiles/train_step.py --only fs10 --steps 5
fi
if [ $part = all ] || [ $part = elementwise ]; then
    timeout -k 10 300 python3 profiles/elementwise_bw.py 2>&1 | grep -v amdgpu.ids > $O/elementwise_bw.txt; cat $O/elementwise_bw.txt
    trace elementwise python3 profiles/elementwise_bw.py
    rm -f $O/pmc_elementwise.txt
    echo "# rocprofv3 --pmc passes over profiles/elementwise_bw.py, per-launch averages; HBM bytes = TCC_EA0_RDREQ x 128 B + TCC_EA0_WRREQ x 64 B" > $O/pmc_elementwise.txt
    pmc pmc_elementwise "TCC_EA0_RDREQ_sum FETCH_SIZE" "layernorm sched_step q_sample normalize add_kernel binary4 randn" python3 profiles/elementwise_bw.py
    pmc pmc_elementwise "TCC_EA0_WRREQ_sum WRITE_SIZE" "layernorm sched_step q_sample normalize add_kernel binary4 randn" python3 profiles/elementwise_bw.py
    cat $O/pmc_elementwise.txt
fi
if [ $part = all ] || [ $part = f16 ]; then
    # fp16 operand-plane precisions of the DiT engine: accuracy at the batch whose products all run on planes, product timings, whole passes
    ADX_TWO_TERM_ABLATION=1 timeout -k 10 600 python3 -u profiles/dit_precision.py 24 16 16 200 2>&1 | grep -v amdgpu.ids > $O/dit_precision_b256.txt; cat $O/dit_precision_b256.txt
    timeout -k 10 300 python3 -u profiles/gemm_f16x2.py 2>&1 | grep -v amdgpu.ids > $O/gemm_f16.txt; cat $O/gemm_f16.txt
    for p in f16 f16x2 bf16x3; do
        timeout -k 10 400 python3 bench.py --workload dit1024 --precision $p --mc-num 16 --dit-steps 20 --steps 2 --warmup 1 > $O/bench_dit1024_mc16_$p.json 2>> $O/bench.err && cut -c1-120 $O/bench_dit1024_mc16_$p.json
    done
    timeout -k 10 400 python3 bench.py --workload dit1024 --precision f16 --dit-steps 20 --steps 2 --warmup 1 > $O/bench_dit1024_f16.json 2>> $O/bench.err
    trace dit1024_f16 python3 bench.py --workload dit1024 --precision f16 --mc-num 16 --dit-steps 10 --steps 1 --warmup 1
fi
if [ $part = f16 ]; then     # the refiner's fp16 precision (bench default): bench line, traces, counter passes of its per-step kernels
    timeout -k 10 500 python3 bench.py --steps 20 --warmup 5 > $O/bench_b1024_f16.json 2> $O/bench_f16.err && cut -c1-300 $O/bench_b1024_f16.json
    trace refiner_f16_b1024 python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    ADX_NO_OVERLAP=1 trace refiner_f16_b1024_one_stream python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras
    rm -f $O/pmc_refiner_f16.txt
    echo "# rocprofv3 --pmc passes over profiles/stage_times.py f16 1024 (B=1024: 828416 conditioning rows), per-launch averages" > $O/pmc_refiner_f16.txt
    echo "# HBM bytes = TCC_EA0_RDREQ x 128 B (= FETCH_SIZE KB doubled, the gfx950 correction) + TCC_EA0_WRREQ x 64 B" >> $O/pmc_refiner_f16.txt
    pmc pmc_refiner_f16 "TCC_EA0_RDREQ_sum FETCH_SIZE" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py f16 1024
    pmc pmc_refiner_f16 "TCC_EA0_WRREQ_sum WRITE_SIZE" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py f16 1024
    pmc pmc_refiner_f16 "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES" "enc_rows attn_g tail_mfma" python3 profiles/stage_times.py f16 1024
    cat $O/pmc_refiner_f16.txt
    python3 profiles/stage_times.py f16 1024 2>&1 | grep -v amdgpu.ids > $O/stage_times_f16.txt; cat $O/stage_times_f16.txt
fi
