# the assembly on one slab of an 8-slab run, per compile-time configuration: tools/slab_cfg.sh "<flags1>" ...
cd $GRAFT_REPO_ROOT
for f in "$@"; do
  rm -f xpic_amd/csrc/ecsim.o
  make -s xpic_amd/libxpic_hip.so EXTRA="$f" > gpurun_out/slab_cfg_build.log 2>&1 || { tail gpurun_out/slab_cfg_build.log; exit 1; }
  echo "[$f]"
  timeout -k 10 200 python tools/fill_slab.py 256 256 32 64 2>&1 | tail -1
  timeout -k 10 200 python tools/fill_slab.py 256 256 64 64 2>&1 | tail -1
  timeout -k 10 200 python tools/fill_slab.py 512 512 64 32 2>&1 | tail -1
done
rm -f xpic_amd/csrc/ecsim.o; make -s xpic_amd/libxpic_hip.so > gpurun_out/slab_cfg_build.log 2>&1
