#!/bin/bash
# training-side evidence only (the inference kernels and their PMC stamp are untouched by a k_train.h change)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 200 python tools/profile_train.py 2 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/${TAG}_train_profile_b2.txt
timeout 300 python tools/config5.py 2 6 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_b2.txt
timeout 300 python tools/config5.py 2 6 --gan 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_gan_b2.txt
timeout 300 python train.py experiment=aero_11-44_512_256 experiment.batch_size=2 steps=5 2>&1 | grep "^{" > gpurun_out/${TAG}_train_py.txt
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_train" -o ${TAG}t -- python "$GRAFT_REPO_ROOT/tools/config5.py" 2 6 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof_train.log" 2>&1
cd "$GRAFT_REPO_ROOT"
timeout 400 python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/${TAG}_bench_train_side.log 2>/dev/null
tail -2 gpurun_out/${TAG}_config5_b2.txt; tail -1 gpurun_out/${TAG}_config5_gan_b2.txt; tail -1 gpurun_out/${TAG}_train_py.txt
grep '^{' gpurun_out/${TAG}_bench_train_side.log | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['config']['other_configs'])"
