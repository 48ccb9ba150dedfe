# the non-uniform loads with the density-scaled preconditioner (kind 4) beside the default: tools/nonuniform_pc.sh "<kinds>" "<loaders>"
set -o pipefail
cd $GRAFT_REPO_ROOT
for L in ${2:-gradient blob}; do
 for K in ${1:-3 4}; do
  timeout -k 10 500 python bench.py --steps 4 --warmup 1 --loader $L --precond $K --no-cpu-baseline --no-probe > gpurun_out/bench_nu_${L}_pc$K.json 2> gpurun_out/bench_nu.err || { tail -5 gpurun_out/bench_nu.err; exit 1; }
  python3 -c "
import json; l=json.load(open('gpurun_out/bench_nu_${L}_pc$K.json')); p=l['phase_ms_per_step']; o=l['occupancy']
print('$L kind $K: ms/step %.1f its %.1f fill %.1f solve %.1f (matA %.1f precond %.1f setup %.1f) stencil steps/it %.1f fallbacks %.1f' % (l['ms_per_step'], l['ksp_iterations_per_step'], p['fill_current'], p['solve_matA'], p['matA_apply'], p['precond'], p['precond_setup'], l['stencil_steps_per_iteration'], o['precond_fallbacks_per_step']))"
 done
done
