# runs the given pytest arguments on the GPU box with the log under gpurun_out/: tools/run_tests.sh <tag> <pytest args...>
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=$1; shift
timeout -k 10 1100 python -m pytest "$@" -x -q -m gpu > gpurun_out/tests_$TAG.log 2>&1 || { tail -30 gpurun_out/tests_$TAG.log | cut -c1-300; exit 1; }
tail -2 gpurun_out/tests_$TAG.log
