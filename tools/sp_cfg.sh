# times the ecsim step phases with different compile-time flags of particles.hip: tools/sp_cfg.sh "<flags1>" ...
set -o pipefail
export XPIC_ALLOW_EXPERIMENT=1  # (flag sets that set a kernel switch are experiment builds: -DXPIC_EXPERIMENT in the flags, common.h)
R=$GRAFT_REPO_ROOT
cd $R
for f in "$@"; do
  rm -f xpic_amd/csrc/particles.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/sp_cfg_build.log 2>&1 || { tail gpurun_out/sp_cfg_build.log; exit 1; }
  timeout -k 10 300 python bench.py --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/sp.json 2> gpurun_out/sp_cfg.err || { tail -3 gpurun_out/sp_cfg.err; }
  python -c "
import json; l=json.load(open('gpurun_out/sp.json')); p=l['phase_ms_per_step']; print('[$f]', l['ms_per_step'], {k:round(v,2) for k,v in p.items() if v>1})"
done
rm -f xpic_amd/csrc/particles.o
