#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_gpu_train.py tests/test_gpu_ops.py -q -k "full_model or lstm_bwd or ftb_autograd" 2>&1 | grep -E "AssertionError|passed|failed|Error" | cut -c1-3000 > gpurun_out/r3c_pytest.txt
python tools/config5.py 16 3 2>&1 | grep step > gpurun_out/r3c_config5_b16.txt
cat gpurun_out/r3c_pytest.txt; cat gpurun_out/r3c_config5_b16.txt
