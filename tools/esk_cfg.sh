# times the Esirkepov push kernels with different compile-time flags: tools/esk_cfg.sh "<flags1>" "<flags2>" ...
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for f in "$@"; do
  rm -f xpic_amd/csrc/esirkepov.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/esk_cfg_build.log 2>&1 || { tail gpurun_out/esk_cfg_build.log; exit 1; }
  for s in basic ecsimcorr; do
    timeout -k 10 300 python bench.py --scheme $s --steps 4 --warmup 1 --no-cpu-baseline > gpurun_out/esk_$s.json 2> gpurun_out/esk_cfg.err || { tail -3 gpurun_out/esk_cfg.err; }
    python -c "
import json; l=json.load(open('gpurun_out/esk_$s.json')); p=l['phase_ms_per_step']; print('[$f] $s:', {k:round(p[k],2) for k in ('basic_push','corr_first_push','corr_second_push') if p[k]>0})"
  done
done
rm -f xpic_amd/csrc/esirkepov.o
