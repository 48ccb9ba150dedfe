# breakdown experiments of k_ecsim_fill on the GPU box: rebuilds ecsim.o with -DFILL_EXP=<n> and times the assembly
# usage: tools/fill_exp.sh "0 1 3 4 s" [grid] [ppc]     (s = in-kernel section stamps)
set -o pipefail
R=$GRAFT_REPO_ROOT
cd $R
for e in $1; do
  rm -f xpic_amd/csrc/ecsim.o
  F="-DXPIC_EXPERIMENT -DFILL_EXP=$e"; [ "$e" = s ] && F="-DXPIC_EXPERIMENT -DFILL_STAMPS"; export XPIC_ALLOW_EXPERIMENT=1
  make -s xpic_amd/libxpic_hip.so EXTRA="$F" > gpurun_out/fill_exp_build_$e.log 2>&1 || { tail gpurun_out/fill_exp_build_$e.log; exit 1; }
  echo -n "FILL_EXP=$e: "
  timeout -k 10 300 python tools/fill_bench.py ${2:-256} ${3:-64} 3 2> gpurun_out/fill_exp_$e.err | tail -40 || { tail -3 gpurun_out/fill_exp_$e.err; }
done
rm -f xpic_amd/csrc/ecsim.o
