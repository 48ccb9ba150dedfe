# HBM bytes of tools/ubench/partial_lines.hip's kernels (two PMC passes, kernel-trace only): tools/ubench/partial_lines.sh
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R/tools/ubench && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o partial_lines.out partial_lines.hip || exit 1
./partial_lines.out > $R/gpurun_out/partial_lines.txt || exit 1
cd /tmp && export TMPDIR=/tmp
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pl_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pl_$c -- $R/tools/ubench/partial_lines.out > $R/gpurun_out/pl_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pl_FETCH_SIZE $R/gpurun_out/pl_WRITE_SIZE >> $R/gpurun_out/partial_lines.txt
cat $R/gpurun_out/partial_lines.txt
