#!/bin/bash
# full evidence visit: GPU parity tests, smoke, bench (+cpu baseline), rocprofv3 kernel stats, PMC traffic passes
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2ev}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/${TAG}_smoke.log 2>&1
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_prof" -o ${TAG} -- python "$GRAFT_REPO_ROOT/bench.py" --steps 5 --warmup 2 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_rocprof.log" 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch" -o f -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_fetch.log" 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write" -o w -- python "$GRAFT_REPO_ROOT/bench.py" --steps 2 --warmup 1 --no-cpu-baseline --no-kernel-events > "$GRAFT_REPO_ROOT/gpurun_out/${TAG}_write.log" 2>&1
cd "$GRAFT_REPO_ROOT"
python tools/pmc_traffic.py gpurun_out/${TAG}_fetch/f_counter_collection.csv gpurun_out/${TAG}_write/w_counter_collection.csv gpurun_out/${TAG}_pmc_traffic.json
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json   # bench.py reads it for roofline.traffic
timeout 400 python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench.log 2>&1
tail -3 gpurun_out/${TAG}_pytest_gpu.log; tail -2 gpurun_out/${TAG}_smoke.log; tail -1 gpurun_out/${TAG}_bench.log | cut -c1-600
timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_launch_table.txt
tail -1 gpurun_out/${TAG}_launch_table.txt
ls gpurun_out/${TAG}_prof/*/ 2>/dev/null | head; ls gpurun_out/${TAG}_prof | head
