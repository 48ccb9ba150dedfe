# error bound (hence degree) of the Chebyshev preconditioner of ecsimcorr's solve ON matM ("correct"): iterations and solve times
cd $GRAFT_REPO_ROOT
export XPIC_ALLOW_EXPERIMENT=1  # (an experiment build: -DXPIC_EXPERIMENT, common.h)
for d in 0.04 0.005 0.00125 0.0001; do
  A="--scheme ecsimcorr --grid 128 --ppc 32 --steps 4 --warmup 1 --no-cpu-baseline"
  rm -f xpic_amd/csrc/api.o; make -s xpic_amd/libxpic_hip.so EXTRA="-DXPIC_EXPERIMENT -DXPIC_CHEB_M_BOUND=$d" > gpurun_out/deg_build.log 2>&1 || exit 1
  echo -n "[bound $d] "
  timeout -k 10 300 python bench.py $A 2> gpurun_out/deg.err | python3 -c "
import json,sys
l=json.loads(sys.stdin.readline()); p=l['phase_ms_per_step']
print('ms/step %.1f its %.1f | solveA %.2f solveM %.2f precond %.2f matA %.2f allreduce %.1f' % (l['ms_per_step'], l['ksp_iterations_per_step'], p['solve_matA'], p['solve_matM'], p['precond'], p['matA_apply'], l['allreduces_per_step']))"
done
rm -f xpic_amd/csrc/api.o; make -s xpic_amd/libxpic_hip.so > gpurun_out/deg_build.log 2>&1
