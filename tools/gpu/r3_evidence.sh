#!/bin/bash
# round-3 evidence visit: GPU parity tests, smoke, rocprofv3 kernel stats of the bench and of a config-5 training step, PMC traffic passes
# (separate --pmc runs, no tracing domains next to them), the bench line (+ CPU baseline, extra configs), launch table, training profile
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r03}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o ${TAG} -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_train" -o ${TAG}t -- python "$GRAFT_REPO_ROOT/tools/config5.py" 2 6 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof_train.log" 2>&1
AERO_STREAMS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch.log" 2>&1
AERO_STREAMS=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_traffic.py gpurun_out/${TAG}_fetch/f_counter_collection.csv gpurun_out/${TAG}_write/w_counter_collection.csv gpurun_out/${TAG}_pmc_traffic.json
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads it for roofline.traffic (stamped with the kernel-source fingerprint)
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
timeout 200 python tools/profile_train.py 2 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/${TAG}_train_profile_b2.txt
timeout 200 python tools/profile_train.py 16 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/${TAG}_train_profile_b16.txt
timeout 300 python tools/config5.py 2 6 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_b2.txt
timeout 300 python tools/config5.py 16 4 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_b16.txt
timeout 300 python tools/config5.py 2 6 --gan 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_gan_b2.txt
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_tl" -o tl -- python "$GRAFT_REPO_ROOT/tools/config5.py" 2 6 > /dev/null 2>&1 )
python tools/step_timeline.py $(find gpurun_out/${TAG}_tl -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_step_timeline_b2.txt 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_tlg" -o tl -- python "$GRAFT_REPO_ROOT/tools/config5.py" 2 6 --gan > /dev/null 2>&1 )
python tools/step_timeline.py $(find gpurun_out/${TAG}_tlg -name "*kernel_trace.csv" | head -1) > gpurun_out/${TAG}_step_timeline_gan_critic_b2.txt 2>&1
rm -rf gpurun_out/${TAG}_tl gpurun_out/${TAG}_tlg
timeout 200 python tools/torch_glue.py 2 2>&1 | grep "torch kernels" > gpurun_out/${TAG}_torch_glue_b2.txt
tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-700
head -3 gpurun_out/${TAG}_train_profile_b2.txt; cat gpurun_out/${TAG}_config5_gan_b2.txt | tail -1
ls gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof_train | head
