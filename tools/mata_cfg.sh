# k_matA under several compile-time configurations in ONE gpurun call: tools/mata_cfg.sh "<flags 1>" "<flags 2>" ...
set -o pipefail
export XPIC_ALLOW_EXPERIMENT=1
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  rm -f xpic_amd/csrc/fields.o
  make -s -j8 xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/mata_cfg_build.log 2>&1 || { tail gpurun_out/mata_cfg_build.log; exit 1; }
  echo -n "[$f]: "
  timeout -k 10 200 python tools/mata_time.py 2> gpurun_out/mata_cfg.err || { tail -3 gpurun_out/mata_cfg.err; exit 1; }
done
rm -f xpic_amd/csrc/fields.o
make -s -j8 xpic_amd/libxpic_hip.so > gpurun_out/mata_cfg_build.log 2>&1
