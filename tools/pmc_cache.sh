# L1 / L2 request counters per kernel (one counter per pass, kernel-trace only): how often a coefficient line of k_matA is
# fetched from L2.  usage: tools/pmc_cache.sh [counters...]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
CS=${@:-TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TOTAL_CACHE_ACCESSES_sum}
for c in $CS; do
  rm -rf $R/gpurun_out/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/gpurun_out/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/pmc_$c.log 2>&1 || { tail -3 $R/gpurun_out/pmc_$c.log; continue; }
  python3 - $R/gpurun_out/pmc_$c $c <<'P'
import collections, csv, glob, re, sys
f = glob.glob(sys.argv[1] + "/*/*counter_collection.csv")[0]
agg = collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    m = re.search(r"(k_[A-Za-z0-9_]+(<[^>]*>)?)", r["Kernel_Name"])
    if m:
        agg[m.group(1)].append(float(r["Counter_Value"]))
for k in ("k_matA<true>", "k_ecsim_fill<true, true>", "k_scatter<true, true>", "k_second_push<true, false, true>", "k_cheb_bar<false, false, true>"):
    if k in agg:
        print("%-28s %-34s %14.4g per launch (%d launches)" % (sys.argv[2], k, sum(agg[k]) / len(agg[k]), len(agg[k])))
P
done
