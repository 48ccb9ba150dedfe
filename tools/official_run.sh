# official artifacts of a build: GPU tests, bench lines, rocprofv3 kernel stats, PMC traffic (separate passes)
# usage: tools/official_run.sh <tag> [part]   part: all | tests | ecsim | side | cfg4 | sq | rehearsal
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-r02}
PART=${2:-all}
cd $R
if [ $PART = all ] || [ $PART = tests ]; then
  timeout -k 10 1100 python -m pytest tests -x -q -m gpu > gpurun_out/tests_gpu_$TAG.log 2>&1 || { tail -20 gpurun_out/tests_gpu_$TAG.log; exit 1; }
  tail -2 gpurun_out/tests_gpu_$TAG.log
fi
if [ $PART = all ] || [ $PART = ecsim ]; then
  timeout -k 10 600 python bench.py --steps 10 --warmup 3 > gpurun_out/bench_256_$TAG.json 2> gpurun_out/bench_256_$TAG.err || { tail gpurun_out/bench_256_$TAG.err; exit 1; }
  cut -c1-400 gpurun_out/bench_256_$TAG.json
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
  rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
  bash $R/tools/pmc_traffic.sh > $R/gpurun_out/pmc_traffic_$TAG.txt 2>&1 || exit 1
  head -8 $R/gpurun_out/pmc_traffic_$TAG.txt
  cd $R
fi
if [ $PART = all ] || [ $PART = side ]; then
  bash tools/profile_scheme.sh basic 128 32 $TAG 0.1 || exit 1
  bash tools/profile_scheme.sh ecsimcorr 128 32 $TAG 1.0 || exit 1
fi
if [ $PART = all ] || [ $PART = cfg4 ]; then
  # one GPU's real share of BASELINE configs[4]: ecsimcorr, 512 x 512 x 64 cells x 32 ppc (537 M particles) as one periodic box
  A="--scheme ecsimcorr --grid-xyz 512 512 64 --ppc 32 --steps 5 --warmup 2"
  timeout -k 10 600 python bench.py $A > gpurun_out/bench_cfg4_$TAG.json 2> gpurun_out/bench_cfg4_$TAG.err || { tail gpurun_out/bench_cfg4_$TAG.err; exit 1; }
  cut -c1-300 gpurun_out/bench_cfg4_$TAG.json
  cd /tmp && export TMPDIR=/tmp
  rm -rf $R/gpurun_out/prof_cfg4_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_cfg4_$TAG -- python3 $R/bench.py $A --no-cpu-baseline > $R/gpurun_out/prof_cfg4_$TAG.log 2>&1 || exit 1
  cd $R
fi
if [ $PART = all ] || [ $PART = sq ]; then
  bash tools/pmc_sq.sh $TAG > /dev/null || exit 1
  head -30 gpurun_out/pmc_sq_$TAG.txt
fi
if [ $PART = all ] || [ $PART = rehearsal ]; then
  XPIC_BENCH_COMM=gloo timeout -k 10 600 python bench.py --gpus 2 --grid 128 --ppc 32 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_n2_rehearsal_$TAG.json 2> gpurun_out/bench_n2_rehearsal_$TAG.err || { tail gpurun_out/bench_n2_rehearsal_$TAG.err; exit 1; }
  cut -c1-300 gpurun_out/bench_n2_rehearsal_$TAG.json
  # BASELINE configs[3] as it is cut: 8 z-slabs of 32 planes of the 256^3 x 64 box, one thread per rank in one process
  XPIC_BENCH_COMM=threads timeout -k 10 600 python bench.py --gpus 8 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/bench_n8_threads_$TAG.json 2> gpurun_out/bench_n8_threads_$TAG.err || { tail gpurun_out/bench_n8_threads_$TAG.err; exit 1; }
  cut -c1-300 gpurun_out/bench_n8_threads_$TAG.json
fi
