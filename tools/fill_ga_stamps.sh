# in-kernel section timers of the gathering assembly inside whole steps (experiment build; the default build is restored):
# tools/fill_ga_stamps.sh "<extra flags>" ...   one run per flag set
set -o pipefail
cd $GRAFT_REPO_ROOT
export XPIC_ALLOW_EXPERIMENT=1
for f in "$@"; do
  rm -f xpic_amd/csrc/ecsim.o xpic_amd/csrc/api.o
  make -s -j8 xpic_amd/libxpic_hip.so EXTRA="-DXPIC_EXPERIMENT -DFILL_STAMPS $f" > gpurun_out/fill_stamps_build.log 2>&1 || { tail gpurun_out/fill_stamps_build.log; exit 1; }
  echo "[$f]"
  for mode in ${FILL_MODES:-1 0}; do
    echo "fused_rebin $mode:"
    timeout -k 10 400 python tools/fill_ga_stamps.py 256 64 3 $mode 2> gpurun_out/fill_stamps.err || { tail -3 gpurun_out/fill_stamps.err; exit 1; }
  done
done
rm -f xpic_amd/csrc/ecsim.o xpic_amd/csrc/api.o
make -s -j8 xpic_amd/libxpic_hip.so > gpurun_out/fill_stamps_build.log 2>&1
