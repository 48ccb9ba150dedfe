# times the Esirkepov push kernels with different compile-time flags: tools/esk_cfg.sh "<flags1>" "<flags2>" ...
# (a flag set that sets a kernel switch needs -DXPIC_EXPERIMENT; the default build is restored at the end)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
export XPIC_ALLOW_EXPERIMENT=1
for f in "$@"; do
  rm -f xpic_amd/csrc/esirkepov.o xpic_amd/csrc/api.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/esk_cfg_build.log 2>&1 || { tail gpurun_out/esk_cfg_build.log; exit 1; }
  for s in "basic --drift 0" "basic" "ecsimcorr" ${ESK_CFG4:+"ecsimcorr --grid-xyz 512 512 64 --ppc 32"}; do
    timeout -k 10 400 python bench.py --scheme $s --steps 4 --warmup 1 --no-cpu-baseline --no-probe > gpurun_out/esk_run.json 2> gpurun_out/esk_cfg.err || { tail -3 gpurun_out/esk_cfg.err; }
    python -c "
import json; l=json.load(open('gpurun_out/esk_run.json')); p=l['phase_ms_per_step']; print('[$f] $s:', {k:round(p[k],2) for k in ('basic_push','corr_first_push','corr_second_push') if p[k]>0})"
  done
done
rm -f xpic_amd/csrc/esirkepov.o xpic_amd/csrc/api.o
make -s xpic_amd/libxpic_hip.so > gpurun_out/esk_cfg_build.log 2>&1
