# times k_ecsim_fill builds with different compile-time configurations: tools/fill_cfg.sh "<flags1>" "<flags2>" ...
set -o pipefail
export XPIC_ALLOW_EXPERIMENT=1  # (flag sets that set a kernel switch are experiment builds: -DXPIC_EXPERIMENT in the flags, common.h)
R=$GRAFT_REPO_ROOT
cd $R
for f in "$@"; do
  rm -f xpic_amd/csrc/ecsim.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/fill_cfg_build.log 2>&1 || { tail gpurun_out/fill_cfg_build.log; exit 1; }
  echo -n "[$f]: "
  timeout -k 10 300 python tools/fill_bench.py 256 64 3 "${FILL_KINDS:-1}" 2> gpurun_out/fill_cfg.err | tail -${FILL_TAIL:-3} || { tail -3 gpurun_out/fill_cfg.err; }
done
rm -f xpic_amd/csrc/ecsim.o
