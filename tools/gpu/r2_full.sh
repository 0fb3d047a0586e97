#!/bin/bash
# full GPU test suite + bench (with kernel events, no cpu baseline)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=${1:-r2m}
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q --maxfail=60 -p no:cacheprovider -s > gpurun_out/${TAG}_pytest_gpu.log 2>&1
echo "pytest exit $?" >> gpurun_out/${TAG}_pytest_gpu.log
timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/${TAG}_bench.log 2>&1
grep -E "rel-L2|passed|failed|FAILED|Error" gpurun_out/${TAG}_pytest_gpu.log | tail -30
python - "$TAG" <<'PY'
import json, sys
l=[x for x in open(f'gpurun_out/{sys.argv[1]}_bench.log') if x.startswith('{')]
if l:
    d=json.loads(l[-1]); print(d['ms_per_step'], d['value'], d['roofline'])
    for k,v in d['kernels_ms_per_step'].items(): print(f'  {v:7.3f}  {k}  {d["kernels_achieved"][k]}')
else: print(open(f'gpurun_out/{sys.argv[1]}_bench.log').read()[-2000:])
PY
