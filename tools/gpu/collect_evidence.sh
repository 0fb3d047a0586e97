#!/bin/bash
# copy the summaries of a tools/gpu/r5_evidence.sh visit from gpurun_out/ (scratch) into profiles/ (tracked)
cd "$(dirname "$0")/../.." || exit 1
TAG=${1:-r05}
for f in pytest_gpu.log smoke.log launch_table.txt train_profile_b2.txt config5_b2.txt config5_gan_b2.txt train_py.txt gan_step1.txt gan_traj.txt pmc_traffic.json; do
  cp gpurun_out/${TAG}_$f profiles/${TAG}_$f
done
cp gpurun_out/${TAG}_prof/${TAG}_kernel_stats.csv profiles/${TAG}_kernel_stats.csv
cp gpurun_out/${TAG}_prof1/${TAG}s1_kernel_stats.csv profiles/${TAG}_kernel_stats_1stream.csv
cp gpurun_out/${TAG}_prof_train/${TAG}t_kernel_stats.csv profiles/${TAG}_kernel_stats_train_config5.csv
grep '^{' gpurun_out/${TAG}_bench.log > profiles/${TAG}_bench.json
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json
python - <<PY
import json
d=json.load(open('profiles/${TAG}_bench.json'))
r=d['roofline']
print('ms',d['ms_per_step'],'one-at-a-time',d['ms_per_step_one_at_a_time'],'frac',r['frac'],'traffic',r['traffic'],'avg_launch_ms',r['avg_launch_ms'])
print('stack',r['conv_stack']); print('stft',r['stft']['frac'],r['stft']['avg_launch_ms'],'istft',r['istft']['frac'],r['istft']['avg_launch_ms'])
print('step_mfma_frac',r['step_mfma_frac'],'kernel sum',r['kernel_sum_ms_per_step']); print(d['config']['other_configs']); print(d['cpu_baseline']['value'], d['cpu_baseline']['value_1_thread'])
ks=d['kernels_ms_per_step']
print('lstm', round(sum(v for k,v in ks.items() if 'lstm' in k),3), 'attn', round(sum(v for k,v in ks.items() if 'attn' in k),3))
PY
