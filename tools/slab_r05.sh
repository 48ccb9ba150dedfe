# round-5 slab measurements on one GPU (self-ring): the matL ghost rows as a blocking RCCL exchange (overlap 0) against the
# copy-engine path (overlap 4: hipMemcpyAsync into the mapped peer buffer on a copy stream), message counts, the 1/2/4/8
# shares, and the kernel trace of the copy path.  usage: tools/slab_r05.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
T=${1:-r05}
cd $R
export HSA_ENABLE_IPC_MODE_LEGACY=0
{
for ov in 0 4 0 4; do echo "== overlap $ov (0: blocking RCCL exchange of the matL ghost rows, 4: peer copy on the copy stream)"; XPIC_SLAB_PEER=1 XPIC_SLAB_OVERLAP=$ov timeout -k 10 200 python tools/step_slab.py ecsim 256 256 32 64 2>&1 | tail -3; done
for nz in ${NZS:-64 128 256}; do echo "== $nz planes"; timeout -k 10 300 python tools/step_slab.py ecsim 256 256 $nz 64 2>&1 | tail -3; done
echo "== ecsimcorr, one GPU's share of configs[4] as a slab (512 x 512 x 64 x 32 ppc)"; timeout -k 10 400 python tools/step_slab.py ecsimcorr 512 512 64 32 2>&1 | tail -3
} > gpurun_out/step_slab_$T.txt 2>&1
cat gpurun_out/step_slab_$T.txt
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/trace_slab_$T
export XPIC_SLAB_OVERLAP=4 XPIC_SLAB_PEER=1
rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $R/gpurun_out/trace_slab_$T -- python3 $R/tools/step_slab.py ecsim 256 256 32 64 > $R/gpurun_out/trace_slab_$T.log 2>&1 || { tail -5 $R/gpurun_out/trace_slab_$T.log; exit 1; }
python3 $R/tools/trace_overlap.py $R/gpurun_out/trace_slab_$T > $R/gpurun_out/trace_overlap_$T.txt 2>&1
cat $R/gpurun_out/trace_overlap_$T.txt
ls $R/gpurun_out/trace_slab_$T/*/ | head
