#!/bin/bash
# round-5 evidence visit: GPU parity suite, smoke, rocprofv3 kernel stats of the bench in BOTH schedules (the timed region's: three batches in
# flight on three streams; and one batch at a time on one stream: VERDICT r3 item 2), of a config-5 training step; PMC traffic passes (separate --pmc runs, no trace domains next to them); the bench line
# (+ CPU baseline, extra configs); launch table; training profile; train.py on one GPU
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r05}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > gpurun_out/${TAG}_pytest_gpu.log
echo "pytest exit ${PIPESTATUS[0]}" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
cd /tmp
BARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o ${TAG} -- python "$GRAFT_REPO_ROOT/bench.py" $BARGS > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.log" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof1" -o ${TAG}s1 -- python "$GRAFT_REPO_ROOT/bench.py" $BARGS > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof_train" -o ${TAG}t -- python "$GRAFT_REPO_ROOT/tools/config5.py" 2 6 > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof_train.log" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch.log" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_traffic.py gpurun_out/${TAG}_fetch/f_counter_collection.csv gpurun_out/${TAG}_write/w_counter_collection.csv gpurun_out/${TAG}_pmc_traffic.json
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads it for roofline.traffic (stamped with the kernel-source fingerprint)
timeout 700 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
timeout 200 python tools/profile_train.py 2 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/${TAG}_train_profile_b2.txt
timeout 300 python tools/config5.py 2 6 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_b2.txt
timeout 300 python tools/config5.py 2 6 --gan 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_gan_b2.txt
timeout 300 python train.py experiment=aero_11-44_512_256 experiment.batch_size=2 steps=5 2>&1 | grep "^{" > gpurun_out/${TAG}_train_py.txt
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-900
tail -2 gpurun_out/${TAG}_train_py.txt; tail -1 gpurun_out/${TAG}_config5_gan_b2.txt
find gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof1 -name "*kernel_stats.csv" | head
timeout 200 python tools/dbg/gan_step1.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > gpurun_out/${TAG}_gan_step1.txt
timeout 200 python tools/dbg/gan_traj.py 2>&1 | grep -v "amdgpu.ids\|Warn\|warn" > gpurun_out/${TAG}_gan_traj.txt
