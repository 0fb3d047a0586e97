#!/bin/bash
# round-6 evidence visit: GPU parity suite (under rocprofv3 --kernel-trace --stats: which kernels the suite launches), smoke, kernel stats of the
# bench in both schedules and of a config-5 training step, PMC traffic passes and the ring tiles' counter passes (separate --pmc runs, no trace
# domains next to them), the bench line (+ CPU baseline, extra configs), launch tables, training profile, kernel coverage
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r06}
mkdir -p gpurun_out
export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"
cd /tmp
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_suite" -o suite -- python -m pytest "$R/tests" -m gpu -q --maxfail=60 -p no:cacheprovider 2>&1 | grep -v "^HIP version\|^ROCm version\|^Hostname\|^Librccl" > "$R/gpurun_out/${TAG}_pytest_gpu.log"
echo "pytest exit ${PIPESTATUS[0]}" >> "$R/gpurun_out/${TAG}_pytest_gpu.log"
cd "$R"
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
cd /tmp
BARGS="--steps 6 --warmup 3 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference"
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof" -o ${TAG} -- python "$R/bench.py" $BARGS > "$R/gpurun_out/${TAG}_rocprof.log" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof1" -o ${TAG}s1 -- python "$R/bench.py" $BARGS > "$R/gpurun_out/${TAG}_rocprof1.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_train" -o ${TAG}t -- python "$R/tools/config5.py" 2 6 > "$R/gpurun_out/${TAG}_rocprof_train.log" 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$R/gpurun_out/${TAG}_prof_c4" -o ${TAG}c4 -- python "$R/tools/launch_table.py" --config4 > "$R/gpurun_out/${TAG}_launch_table_config4.txt" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$R/gpurun_out/${TAG}_fetch" -o f -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference > "$R/gpurun_out/${TAG}_fetch.log" 2>&1
AERO_PIPELINE=1 AERO_STREAMS=1 timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$R/gpurun_out/${TAG}_write" -o w -- python "$R/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events --no-extra-configs --no-serial-reference > "$R/gpurun_out/${TAG}_write.log" 2>&1
run() { n=$1; shift; timeout 200 rocprofv3 --pmc "$@" --output-format csv -d "$R/gpurun_out/${TAG}_ringpmc$n" -o p -- python "$R/tools/bench_conv.py" --layers d0,d1,d2,d3 --iters 3 > "$R/gpurun_out/${TAG}_ringpmc$n.log" 2>&1; }
run 1 GRBM_GUI_ACTIVE SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_INSTS_LDS
cd "$R"
python tools/pmc_traffic.py $(find gpurun_out/${TAG}_fetch -name "*counter_collection.csv" | head -1) $(find gpurun_out/${TAG}_write -name "*counter_collection.csv" | head -1) gpurun_out/${TAG}_pmc_traffic.json
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads it for roofline.traffic (stamped with the kernel-source fingerprint)
python tools/pmc_ring_table.py $(find gpurun_out/${TAG}_ringpmc1 -name "*counter_collection.csv") --json gpurun_out/${TAG}_pmc_clock.json > gpurun_out/${TAG}_pmc_clock_table.txt 2>&1
cp gpurun_out/${TAG}_pmc_clock.json profiles/pmc_clock.json       # bench.py: roofline.clock_ghz
timeout 900 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>gpurun_out/${TAG}_bench.err
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
timeout 200 python tools/profile_train.py 2 2>&1 | grep -v "amdgpu.ids\|Warn" > gpurun_out/${TAG}_train_profile_b2.txt
timeout 300 python tools/config5.py 2 6 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_b2.txt
timeout 300 python tools/config5.py 2 6 --gan 2>&1 | grep "^step" > gpurun_out/${TAG}_config5_gan_b2.txt
timeout 300 python train.py experiment=aero_11-44_512_256 experiment.batch_size=2 steps=5 2>&1 | grep "^{" > gpurun_out/${TAG}_train_py.txt
python tools/kernel_coverage.py --suite $(find gpurun_out/${TAG}_prof_suite -name "*kernel_stats.csv" | head -1) --bench $(find gpurun_out/${TAG}_prof -name "*kernel_stats.csv" | head -1) \
   --bench $(find gpurun_out/${TAG}_prof1 -name "*kernel_stats.csv" | head -1) --config5 $(find gpurun_out/${TAG}_prof_train -name "*kernel_stats.csv" | head -1) \
   --config4 $(find gpurun_out/${TAG}_prof_c4 -name "*kernel_stats.csv" | head -1) > gpurun_out/${TAG}_kernel_coverage.txt 2>&1
for d in prof prof1 prof_train; do f=$(find gpurun_out/${TAG}_$d -name "*kernel_stats.csv" | head -1); cp "$f" gpurun_out/${TAG}_kernel_stats_$d.csv; done
rm -rf gpurun_out/${TAG}_prof_suite gpurun_out/${TAG}_prof gpurun_out/${TAG}_prof1 gpurun_out/${TAG}_prof_train gpurun_out/${TAG}_prof_c4 gpurun_out/${TAG}_fetch gpurun_out/${TAG}_write gpurun_out/${TAG}_ringpmc1
tail -4 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; grep '^{' gpurun_out/${TAG}_bench.log | cut -c1-600
tail -2 gpurun_out/${TAG}_train_py.txt; tail -1 gpurun_out/${TAG}_config5_gan_b2.txt; head -3 gpurun_out/${TAG}_kernel_coverage.txt
