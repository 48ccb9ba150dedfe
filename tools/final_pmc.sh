# call A of the final artifacts: PMC traffic of the three schemes + SQ counters of the fill (copy the summaries into
# profiles/ BEFORE call B, so that bench.py can quote them as roofline.traffic):  tools/final_pmc.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/pmc_traffic_$TAG.txt 2>&1 || exit 1
head -8 $R/gpurun_out/pmc_traffic_$TAG.txt
for S in basic ecsimcorr; do
  DT=1.0; [ $S = basic ] && DT=0.1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/gpurun_out/pmc_${S}_$c
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${S}_$c -- python3 $R/bench.py --scheme $S --grid 128 --ppc 32 --dt $DT --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${S}_$c.log 2>&1 || exit 1
  done
  python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${S}_FETCH_SIZE $R/gpurun_out/pmc_${S}_WRITE_SIZE > $R/gpurun_out/pmc_traffic_${S}_$TAG.txt
  head -6 $R/gpurun_out/pmc_traffic_${S}_$TAG.txt
done
cd $R && bash tools/pmc_sq.sh $TAG > /dev/null || exit 1
head -12 gpurun_out/pmc_sq_$TAG.txt
