#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_train.py -m gpu -q -x -p no:cacheprovider -k "stft or frozen" 2>&1 | tail -2
for i in 1 2 3; do timeout 200 python tools/launch_table.py 2>&1 | grep "stft\|sum of"; done
for i in 1 2; do timeout 200 python tools/dbg/pipeline_fill_drain.py 2>&1 | grep '^K=' | cut -c1-30 | tr '\n' ' '; echo; done
