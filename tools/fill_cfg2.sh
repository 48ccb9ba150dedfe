# several compile-time configurations of ecsim.hip timed with tools/fill_bench.py in ONE gpurun call, then the default
# build's parity tests:  tools/fill_cfg2.sh "<flags1>" "<flags2>" ...
set -o pipefail
export XPIC_ALLOW_EXPERIMENT=1  # (flag sets that set a kernel switch are experiment builds: -DXPIC_EXPERIMENT in the flags, common.h)
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  rm -f xpic_amd/csrc/ecsim.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/fill_cfg_build.log 2>&1 || { tail gpurun_out/fill_cfg_build.log; exit 1; }
  echo -n "[$f]: "
  timeout -k 10 300 python tools/fill_bench.py 256 64 3 2> gpurun_out/fill_cfg.err | tail -1 || { tail -3 gpurun_out/fill_cfg.err; exit 1; }
done
rm -f xpic_amd/csrc/ecsim.o
make -s xpic_amd/libxpic_hip.so > gpurun_out/fill_cfg_build.log 2>&1 || { tail gpurun_out/fill_cfg_build.log; exit 1; }
