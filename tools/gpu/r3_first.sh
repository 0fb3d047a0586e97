#!/bin/bash
# round 3, first visit: the whole GPU suite, the new training-step op tests, the whole-model gradient check, config 5 fwd+bwd
mkdir -p gpurun_out
python -m pytest tests -m gpu -x -q 2>&1 | tail -15 > gpurun_out/r3_pytest.txt
python tools/train_check.py small 800 --vjp > gpurun_out/r3_train_small_vjp.txt 2>&1
python tools/train_check.py small 800 > gpurun_out/r3_train_small.txt 2>&1
python tools/train_check.py full 8000 --vjp > gpurun_out/r3_train_full_vjp.txt 2>&1
python tools/config5.py > gpurun_out/r3_config5.txt 2>&1
python bench.py --steps 20 --warmup 5 > gpurun_out/r3_bench.txt 2>&1
tail -3 gpurun_out/r3_pytest.txt; tail -4 gpurun_out/r3_train_small_vjp.txt; tail -3 gpurun_out/r3_train_full_vjp.txt; tail -12 gpurun_out/r3_config5.txt; tail -1 gpurun_out/r3_bench.txt | cut -c1-400
