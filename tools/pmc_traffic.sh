# HBM traffic per kernel: FETCH_SIZE and WRITE_SIZE in separate passes, kernel-trace only (guide: HBM / rocprofv3)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
