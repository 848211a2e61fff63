# usage: bash profiles/ab_rows.sh   -- same-box A/B of the per-step kernels: autodiffusion_amd/libadx_base.so (an older build put
# there by hand) against libadx.so, two rounds, stage times in isolation plus the whole bench line
cd $GRAFT_REPO_ROOT
for i in 1 2; do
  ADX_LIB_PATH=$GRAFT_REPO_ROOT/autodiffusion_amd/libadx_base.so timeout -k 10 120 python3 profiles/stage_times.py bf16x3 1024 2>/dev/null | grep -v "cache build" | sed 's/^/base: /' || exit 1
  timeout -k 10 120 python3 profiles/stage_times.py bf16x3 1024 2>/dev/null | grep -v "cache build" | sed 's/^/new:  /' || exit 1
done
ADX_LIB_PATH=$GRAFT_REPO_ROOT/autodiffusion_amd/libadx_base.so timeout -k 10 200 python3 bench.py --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-150 | sed 's/^/base: /' || exit 1
timeout -k 10 200 python3 bench.py --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-150 | sed 's/^/new:  /' || exit 1
ADX_LIB_PATH=$GRAFT_REPO_ROOT/autodiffusion_amd/libadx_base.so timeout -k 10 200 python3 bench.py --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-150 | sed 's/^/base: /' || exit 1
timeout -k 10 200 python3 bench.py --steps 10 --no-cpu-baseline --no-extras 2>/dev/null | cut -c1-150 | sed 's/^/new:  /' || exit 1
