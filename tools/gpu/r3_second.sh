#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -q -k "train or music or config5 or full_model or stft_loss_value or localstate_bwd or lstm_bwd or ftb_autograd or freqfc_wgrad or frames_op or gate_bwd" 2>&1 | tail -25 > gpurun_out/r3b_pytest.txt
cd /tmp && export TMPDIR=/tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3b_prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/config5.py 16 3 > $GRAFT_REPO_ROOT/gpurun_out/r3b_prof_c5.log 2>&1
cd $GRAFT_REPO_ROOT
tail -8 gpurun_out/r3b_pytest.txt
f=$(find gpurun_out/r3b_prof_c5 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && head -40 "$f" | cut -c1-220
