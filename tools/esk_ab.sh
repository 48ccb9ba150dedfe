# Esirkepov pushes of the working tree: tests, then the side benches with phases (one box): tools/esk_ab.sh <tag> [cfg4]
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-x}
timeout -k 10 900 python -m pytest tests/test_gpu_schemes.py tests/test_gpu_edges.py -x -q -m gpu > gpurun_out/tests_esk_$TAG.log 2>&1 || { tail -30 gpurun_out/tests_esk_$TAG.log | cut -c1-300; exit 1; }
tail -1 gpurun_out/tests_esk_$TAG.log
show() { python3 -c "
import json,sys; l=json.load(open(sys.argv[1])); p=l['phase_ms_per_step']
print(sys.argv[2], 'ms/step %.2f' % l['ms_per_step'], {k: round(p[k], 2) for k in ('basic_push','corr_first_push','corr_second_push','fill_current','scatter','index','move_bin','solve_matA','solve_matM') if p.get(k,0) > 0})" $1 $2; }
for rep in 1 2; do
  timeout -k 10 300 python bench.py --scheme basic --steps 6 --warmup 2 --no-cpu-baseline --no-probe > gpurun_out/esk_basic_$TAG.json 2> gpurun_out/esk.err || { tail -3 gpurun_out/esk.err; exit 1; }
  show gpurun_out/esk_basic_$TAG.json basic128
  timeout -k 10 300 python bench.py --scheme ecsimcorr --steps 5 --warmup 2 --no-cpu-baseline --no-probe > gpurun_out/esk_corr_$TAG.json 2> gpurun_out/esk.err || { tail -3 gpurun_out/esk.err; exit 1; }
  show gpurun_out/esk_corr_$TAG.json ecsimcorr128
done
if [ "$2" = cfg4 ]; then
  timeout -k 10 500 python bench.py --scheme ecsimcorr --grid-xyz 512 512 64 --ppc 32 --steps 4 --warmup 2 --no-cpu-baseline --no-probe > gpurun_out/esk_cfg4_$TAG.json 2> gpurun_out/esk.err || { tail -3 gpurun_out/esk.err; exit 1; }
  show gpurun_out/esk_cfg4_$TAG.json cfg4
fi
