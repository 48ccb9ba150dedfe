# SQ counters of the Esirkepov push kernels (three passes, each with kernel-trace only).  usage: tools/pmc_sq_esk.sh <scheme> <dt> [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
S=${1:-basic}; DT=${2:-0.1}; T=${3:-r02}
ARGS="--scheme $S --grid 128 --ppc 32 --dt $DT --steps 2 --warmup 1 --no-cpu-baseline"
rm -rf $R/gpurun_out/sqe1_${S}_$T $R/gpurun_out/sqe2_${S}_$T $R/gpurun_out/sqe3_${S}_$T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sqe1_${S}_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqe1_${S}_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/sqe2_${S}_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqe2_${S}_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_WAVE32_INSTS SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/sqe3_${S}_$T -- python3 $R/bench.py $ARGS > $R/gpurun_out/sqe3_${S}_$T.log 2>&1
for i in 1 2 3; do python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sqe${i}_${S}_$T k_esirkepov_push; done > $R/gpurun_out/pmc_sq_esk_${S}_$T.txt 2>&1
cat $R/gpurun_out/pmc_sq_esk_${S}_$T.txt
