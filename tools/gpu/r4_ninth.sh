#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
export TMPDIR=/tmp
T="tests/test_gpu_model.py::test_two_stream_forward_equals_one_stream tests/test_gpu_model.py::test_full_model_batch64_matches_golden_and_is_batch_invariant"
echo default; timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo AERO_PW=0; AERO_PW=0 timeout 300 python -m pytest $T -m gpu -q -p no:cacheprovider 2>&1 | tail -3
echo comm; timeout 100 python -m pytest tests/test_gpu_distrib.py -m gpu -q -p no:cacheprovider 2>&1 | tail -3
