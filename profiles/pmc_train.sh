# usage (GPU box, repository root): bash profiles/pmc_train.sh   -- counter passes over one training step of the refiner
# (frame stack 10, B = 64), per-launch averages of the attention and weight-gradient kernels -> gpurun_out/pmc_train.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out; rm -f $O/pmc_train.txt
echo "# rocprofv3 --pmc passes over profiles/train_step.py --only fs10 --steps 2 (B = 64, 8 chunks as one batch: 414208 conditioning rows), per-launch averages" > $O/pmc_train.txt
pass() {
    timeout -s KILL 240 rocprofv3 --pmc $1 --output-format csv -d $O/pmc_tmp -- python3 profiles/train_step.py --only fs10 --steps 2 > $O/pmc_tmp.log 2>&1 || { tail -5 $O/pmc_tmp.log; return 1; }
    t=$(find $O/pmc_tmp -name "*counter_collection.csv" | head -1); python3 profiles/summarize_pmc.py "$t" attn_bwd attn_fwd_stats wgrad_partial >> $O/pmc_train.txt; rm -rf $O/pmc_tmp
}
pass "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_LDS" || exit 1
pass "TCC_EA0_RDREQ_sum FETCH_SIZE" || exit 1
pass "GRBM_GUI_ACTIVE" || exit 1
cat $O/pmc_train.txt
