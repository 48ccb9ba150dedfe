# the blob load with the density-scaled surrogate and its ratios capped: tools/rcap.sh "<caps>" [loader]
set -o pipefail
cd $GRAFT_REPO_ROOT
for R in $1; do
  XPIC_RCAP=$R timeout -k 10 500 python bench.py --steps 3 --warmup 1 --loader ${2:-blob} --precond 4 --no-cpu-baseline --no-probe > gpurun_out/bench_rcap.json 2> gpurun_out/bench_rcap.err || { tail -5 gpurun_out/bench_rcap.err; exit 1; }
  python3 -c "
import json; l=json.load(open('gpurun_out/bench_rcap.json')); p=l['phase_ms_per_step']; o=l['occupancy']
print('${2:-blob} cap $R: ms/step %.1f its %.1f solve %.1f (matA %.1f precond %.1f) stencil steps/it %.1f fallbacks %.1f' % (l['ms_per_step'], l['ksp_iterations_per_step'], p['solve_matA'], p['matA_apply'], p['precond'], l['stencil_steps_per_iteration'], o['precond_fallbacks_per_step']))"
done
