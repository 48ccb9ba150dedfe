# SQ counters of the solve's kernels (k_cheb_bar, k_matA) inside whole steps; two passes, each with kernel-trace only.  usage: tools/pmc_sq_solve.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r05}
rm -rf $R/gpurun_out/sqs1_$T $R/gpurun_out/sqs2_$T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sqs1_$T -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sqs1_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/sqs2_$T -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/sqs2_$T.log 2>&1
{ for k in "k_cheb_bar<false, false" "k_matA<true>" "k_second_push"; do echo "== $k"; for i in 1 2; do python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sqs${i}_$T "$k"; done; done; } > $R/gpurun_out/pmc_sq_solve_$T.txt 2>&1
cat $R/gpurun_out/pmc_sq_solve_$T.txt
