# call B of the final artifacts (after the PMC summaries of call A were copied into profiles/): bench lines + rocprofv3
# kernel stats of the headline and the side configurations, the rehearsals.  tools/final_bench.sh <tag>
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r03}
cd $R
timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_256_$TAG.json 2> gpurun_out/bench_256_$TAG.err || { tail gpurun_out/bench_256_$TAG.err; exit 1; }
cut -c1-300 gpurun_out/bench_256_$TAG.json
for S in basic ecsimcorr; do
  DT=1.0; [ $S = basic ] && DT=0.1
  timeout -k 10 300 python bench.py --scheme $S --grid 128 --ppc 32 --dt $DT --steps 5 --warmup 2 > gpurun_out/bench_${S}_$TAG.json 2> gpurun_out/bench_${S}_$TAG.err || { tail gpurun_out/bench_${S}_$TAG.err; exit 1; }
  cut -c1-300 gpurun_out/bench_${S}_$TAG.json
done
A4="--scheme ecsimcorr --grid-xyz 512 512 64 --ppc 32 --steps 5 --warmup 2"
timeout -k 10 600 python bench.py $A4 > gpurun_out/bench_cfg4_$TAG.json 2> gpurun_out/bench_cfg4_$TAG.err || { tail gpurun_out/bench_cfg4_$TAG.err; exit 1; }
cut -c1-300 gpurun_out/bench_cfg4_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG $R/gpurun_out/prof_basic_$TAG $R/gpurun_out/prof_ecsimcorr_$TAG $R/gpurun_out/prof_cfg4_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_basic_$TAG -- python3 $R/bench.py --scheme basic --grid 128 --ppc 32 --dt 0.1 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_basic_$TAG.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_ecsimcorr_$TAG -- python3 $R/bench.py --scheme ecsimcorr --grid 128 --ppc 32 --steps 5 --warmup 2 --no-cpu-baseline > $R/gpurun_out/prof_ecsimcorr_$TAG.log 2>&1 || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4_$TAG -- python3 $R/bench.py $A4 --no-cpu-baseline > $R/gpurun_out/prof_cfg4_$TAG.log 2>&1 || exit 1
cd $R
bash tools/official_run.sh $TAG rehearsal
