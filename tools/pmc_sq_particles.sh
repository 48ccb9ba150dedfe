# SQ counters of the particle kernels of the ecsim step (k_second_push, k_scatter); three passes, each with kernel-trace only.  usage: tools/pmc_sq_particles.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
ARGS="--steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $R/gpurun_out/sqp1_$T $R/gpurun_out/sqp2_$T $R/gpurun_out/sqp3_$T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sqp1_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqp1_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/sqp2_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqp2_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_THREAD_CYCLES_VALU SQ_INSTS_SMEM SQ_WAIT_INST_ANY SQ_INSTS_GDS --output-format csv -d $R/gpurun_out/sqp3_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqp3_$T.log 2>&1
for k in k_second_push "k_scatter<true"; do
  echo "== $k"
  for i in 1 2 3; do python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sqp${i}_$T "$k"; done
done > $R/gpurun_out/pmc_sq_particles_$T.txt 2>&1
cat $R/gpurun_out/pmc_sq_particles_$T.txt
