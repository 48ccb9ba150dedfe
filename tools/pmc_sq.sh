# SQ counters of both assembly kernels (classic, warp-specialised); two passes, each with kernel-trace only.  usage: tools/pmc_sq.sh [tag]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
T=${1:-r02}
rm -rf $R/gpurun_out/sq1_$T $R/gpurun_out/sq2_$T
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $R/gpurun_out/sq1_$T -- python3 $R/tools/fill_bench.py 256 64 1 "0 1" > $R/gpurun_out/sq1_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_INSTS_LDS SQ_INST_CYCLES_VMEM SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM --output-format csv -d $R/gpurun_out/sq2_$T -- python3 $R/tools/fill_bench.py 256 64 1 "0 1" > $R/gpurun_out/sq2_$T.log 2>&1 &&
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_F64 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_MISC SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_SCA SQ_WAVE32_INSTS SQ_THREAD_CYCLES_VALU --output-format csv -d $R/gpurun_out/sq3_$T -- python3 $R/tools/fill_bench.py 256 64 1 "0 1" > $R/gpurun_out/sq3_$T.log 2>&1
{ for k in "k_ecsim_fill<" "k_ecsim_fill_ws"; do echo "== $k"; for i in 1 2 3; do python3 $R/tools/pmc_sq_summary.py $R/gpurun_out/sq${i}_$T "$k"; done; done; } > $R/gpurun_out/pmc_sq_$T.txt 2>&1
cat $R/gpurun_out/pmc_sq_$T.txt
