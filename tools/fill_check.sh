# parity tests that exercise k_ecsim_fill, then the assembly alone timed at 256^3 x 64 (tools/fill_bench.py)
set -o pipefail
cd $GRAFT_REPO_ROOT
timeout -k 10 500 python -m pytest tests/test_gpu_ecsim.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/fill_check_tests.log 2>&1 || { tail -20 gpurun_out/fill_check_tests.log; exit 1; }
tail -1 gpurun_out/fill_check_tests.log
timeout -k 10 300 python tools/fill_bench.py 256 64 3 2> gpurun_out/fill_check.err | tail -3
