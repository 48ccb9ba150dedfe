# the headline step on plasmas that are not uniform (review item 5): tools/nonuniform.sh <tag>
# uniform, gradient (4 : 1 along x) and blob (1 % in a Gaussian clump) loads of the same particle total on ONE box
set -o pipefail
cd $GRAFT_REPO_ROOT
TAG=${1:-r05}
for L in poisson gradient blob; do
  timeout -k 10 500 python bench.py --steps 5 --warmup 2 --loader $L --no-cpu-baseline > gpurun_out/bench_nonuniform_${L}_$TAG.json 2> gpurun_out/bench_nonuniform_$L.err || { tail -5 gpurun_out/bench_nonuniform_$L.err; exit 1; }
  python3 -c "
import json; l=json.load(open('gpurun_out/bench_nonuniform_${L}_$TAG.json')); p=l['phase_ms_per_step']; o=l['occupancy']
print('$L: ms/step %.1f its %.1f fill %.1f solve %.1f push2 %.1f scatter %.1f index %.1f move_bin %.1f | max cell %d >64 %d >128 %d >bucket %d max/mean pencil %.2f idx/step %.1f rebuilds %.1f fallbacks %.1f' % (l['ms_per_step'], l['ksp_iterations_per_step'], p['fill_current'], p['solve_matA'], p['second_push'], p['scatter'], p['index'], p['move_bin'], o['max_cell'], o['cells_over_64'], o['cells_over_128'], o['cells_over_bucket'], o['max_pencil']/o['mean_pencil'], o['index_passes_per_step'], o['key_rebuilds_per_step'], o['precond_fallbacks_per_step']))"
done
