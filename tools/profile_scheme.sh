# rocprofv3 kernel stats + PMC traffic (separate passes) of a side scheme: tools/profile_scheme.sh <scheme> <grid> <ppc> <tag> <dt>
set -o pipefail
R=$GRAFT_REPO_ROOT
S=${1:-basic}; G=${2:-128}; P=${3:-32}; TAG=${4:-r02}; DT=${5:-1.0}
ARGS="--scheme $S --grid $G --ppc $P --dt $DT --steps 5 --warmup 2"
cd $R && timeout -k 10 300 python bench.py $ARGS > gpurun_out/bench_${S}_$TAG.json 2> gpurun_out/bench_${S}_$TAG.err || { tail gpurun_out/bench_${S}_$TAG.err; exit 1; }
cat gpurun_out/bench_${S}_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_${S}_$TAG
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_${S}_$TAG -- python3 $R/bench.py $ARGS > $R/gpurun_out/prof_${S}_$TAG.log 2>&1 || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf $R/gpurun_out/pmc_${S}_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_${S}_$c -- python3 $R/bench.py --scheme $S --grid $G --ppc $P --dt $DT --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_${S}_$c.log 2>&1 || exit 1
done
python3 $R/tools/pmc_summary.py $R/gpurun_out/pmc_${S}_FETCH_SIZE $R/gpurun_out/pmc_${S}_WRITE_SIZE > $R/gpurun_out/pmc_traffic_${S}_$TAG.txt
head -12 $R/gpurun_out/pmc_traffic_${S}_$TAG.txt
