# round-4 slab measurements on one GPU (self-ring): A/B of the overlapped matL ghost-row exchange, k_cheb_bar z-chunks,
# and the kernel trace that shows RCCL kernels beside the assembly.  usage: tools/slab_r04.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
T=${1:-r04}
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for ov in 1 3 1 3 0; do echo "== overlap $ov (bit 0: operator halos, bit 1: matL ghost rows)"; XPIC_SLAB_OVERLAP=$ov timeout -k 10 200 python tools/step_slab.py ecsim 256 256 32 64 2>&1 | tail -3; done
for zc in ${ZCS:-8 16 32}; do echo "== k_cheb_bar z-chunk $zc"; XPIC_CHEB_ZC=$zc timeout -k 10 200 python tools/step_slab.py ecsim 256 256 32 64 2>&1 | tail -3; done
for nz in ${NZS:-64 128 256}; do echo "== $nz planes"; timeout -k 10 300 python tools/step_slab.py ecsim 256 256 $nz 64 2>&1 | tail -3; done
} > gpurun_out/step_slab_$T.txt 2>&1
cat gpurun_out/step_slab_$T.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_slab_$T
export XPIC_SLAB_OVERLAP=3
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/trace_slab_$T -- python3 $R/tools/step_slab.py ecsim 256 256 32 64 > $R/gpurun_out/trace_slab_$T.log 2>&1 || { tail -5 $R/gpurun_out/trace_slab_$T.log; exit 1; }
python3 $R/tools/trace_overlap.py $R/gpurun_out/trace_slab_$T > $R/gpurun_out/trace_overlap_$T.txt 2>&1
cat $R/gpurun_out/trace_overlap_$T.txt
rm -rf $R/gpurun_out/trace_slab_$T
