# bench.py under several compile-time configurations in ONE gpurun call, selected phases printed per configuration:
#   tools/bench_cfg.sh "<bench args>" "<make EXTRA flags 1>" "<flags 2>" ...      (the default build is restored at the end)
set -o pipefail
export XPIC_ALLOW_EXPERIMENT=1  # (flag sets that set a kernel switch are experiment builds: -DXPIC_EXPERIMENT in the flags, common.h)
cd $GRAFT_REPO_ROOT
ARGS=$1; shift
for f in "$@"; do
  rm -f xpic_amd/csrc/*.o
  make -s -j8 xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/bench_cfg_build.log 2>&1 || { tail gpurun_out/bench_cfg_build.log; exit 1; }
  echo -n "[$f]: "
  timeout -k 10 400 python bench.py $ARGS --no-cpu-baseline 2> gpurun_out/bench_cfg.err | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); p=l['phase_ms_per_step']
print('ms/step %.1f its %.1f | fill %.1f solve %.1f matA %.1f precond %.1f setup %.2f scatter %.1f index %.1f push2 %.1f mdot %.2f maxpy %.2f' % (l['ms_per_step'], l['ksp_iterations_per_step'], p['fill_current'], p['solve_matA'], p['matA_apply'], p['precond'], p.get('precond_setup',0), p['scatter'], p.get('index', 0), p['second_push'], p['mdot'], p['maxpy']), '| basic %.2f corr1 %.2f corr2 %.2f solveM %.1f' % (p['basic_push'], p['corr_first_push'], p['corr_second_push'], p['solve_matM']))" || { tail -3 gpurun_out/bench_cfg.err; exit 1; }
done
rm -f xpic_amd/csrc/*.o
make -s -j8 xpic_amd/libxpic_hip.so > gpurun_out/bench_cfg_build.log 2>&1 || { tail gpurun_out/bench_cfg_build.log; exit 1; }
