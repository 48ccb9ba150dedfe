# SQ counters of the assembly inside whole ecsim steps, gathering form (fused re-binning 1) against scatter first (0):
# two passes each, kernel-trace only.  usage: tools/pmc_sq_fill_ga.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r04}
for F in 1 0; do
  A="--steps 2 --warmup 1 --no-cpu-baseline --no-probe --fused-rebin $F"
  rm -rf $R/gpurun_out/sqg1_${F}_$T $R/gpurun_out/sqg2_${F}_$T
  rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sqg1_${F}_$T -- python3 $R/bench.py $A > $R/gpurun_out/sqg1_${F}_$T.log 2>&1 || exit 1
  rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/sqg2_${F}_$T -- python3 $R/bench.py $A > $R/gpurun_out/sqg2_${F}_$T.log 2>&1 || exit 1
  { echo "== fused re-binning $F"; for i in 1 2; do python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sqg${i}_${F}_$T "k_ecsim_fill<"; done; } > $R/gpurun_out/pmc_sq_fill_ga_${F}_$T.txt 2>&1
done
cat $R/gpurun_out/pmc_sq_fill_ga_1_$T.txt $R/gpurun_out/pmc_sq_fill_ga_0_$T.txt
