# copies the artifacts of tools/final_bench.sh <tag> from gpurun_out/ into profiles/${P}_*:  tools/copy_final.sh <tag>
T=${1:-r04}; P=${2:-r04}
cd /root/repo
cp gpurun_out/bench_256_$T.json profiles/${P}_bench_256.json
cp gpurun_out/bench_basic_$T.json profiles/${P}_bench_basic_128.json
cp gpurun_out/bench_ecsimcorr_$T.json profiles/${P}_bench_ecsimcorr_128.json
cp gpurun_out/bench_cfg4_$T.json profiles/${P}_bench_cfg4_512x512x64.json
cp gpurun_out/bench_n2_rehearsal_$T.json profiles/${P}_bench_n2_gloo_rehearsal_128.json
cp gpurun_out/bench_n8_threads_$T.json profiles/${P}_bench_n8_threads_rehearsal_256.json
cp "$(ls -t gpurun_out/prof_$T/runc/*_kernel_stats.csv | head -1)" profiles/${P}_rocprofv3_kernel_stats_256.csv  # (the merged directory keeps older runs: the newest file)
cp "$(ls -t gpurun_out/prof_basic_$T/runc/*_kernel_stats.csv | head -1)" profiles/${P}_rocprofv3_kernel_stats_basic_128.csv  # (the merged directory keeps older runs: the newest file)
cp "$(ls -t gpurun_out/prof_ecsimcorr_$T/runc/*_kernel_stats.csv | head -1)" profiles/${P}_rocprofv3_kernel_stats_ecsimcorr_128.csv  # (the merged directory keeps older runs: the newest file)
cp "$(ls -t gpurun_out/prof_cfg4_$T/runc/*_kernel_stats.csv | head -1)" profiles/${P}_rocprofv3_kernel_stats_cfg4.csv  # (the merged directory keeps older runs: the newest file)
python3 - $P <<'PYEOF'
import json, sys
for f in ('256', 'basic_128', 'ecsimcorr_128', 'cfg4_512x512x64'):
    l = json.load(open('/root/repo/profiles/%s_bench_%s.json' % (sys.argv[1], f)))
    r = l['roofline']
    print(f, round(l['ms_per_step'], 2), 'its', l['ksp_iterations_per_step'], 'roofline', r['kernel'][:28], 'frac %.4f' % r['frac'], 'traffic', r['traffic'])
PYEOF
