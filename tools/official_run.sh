# official artifacts of a build: GPU tests, bench line, rocprofv3 kernel stats, PMC traffic (separate passes)
set -o pipefail
R=$GRAFT_REPO_ROOT
TAG=${1:-v5}
cd $R && timeout -k 10 900 python -m pytest tests -x -q -m gpu > gpurun_out/tests_gpu.log 2>&1 || { tail -20 gpurun_out/tests_gpu.log; exit 1; }
tail -2 gpurun_out/tests_gpu.log
timeout -k 10 600 python bench.py > gpurun_out/bench_256_$TAG.json 2> gpurun_out/bench_256_$TAG.err || { tail gpurun_out/bench_256_$TAG.err; exit 1; }
cat gpurun_out/bench_256_$TAG.json
cd /tmp && export TMPDIR=/tmp
rm -rf $R/gpurun_out/prof_$TAG && rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/prof_$TAG -- python3 $R/bench.py --no-cpu-baseline > $R/gpurun_out/prof_$TAG.log 2>&1 || exit 1
rm -rf $R/gpurun_out/pmc_FETCH_SIZE $R/gpurun_out/pmc_WRITE_SIZE
bash $R/tools/pmc_traffic.sh > $R/gpurun_out/pmc_traffic_$TAG.txt 2>&1 || exit 1
head -8 $R/gpurun_out/pmc_traffic_$TAG.txt
