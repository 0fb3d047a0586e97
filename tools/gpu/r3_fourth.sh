#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests -m gpu -q 2>&1 | tail -25 > gpurun_out/r3d_pytest.txt
python tools/config5.py 2 4 2>&1 | grep step > gpurun_out/r3d_config5_b2.txt
python tools/config5.py 16 3 2>&1 | grep step > gpurun_out/r3d_config5_b16.txt
python bench.py --steps 20 --warmup 5 > gpurun_out/r3d_bench.txt 2>&1
tail -12 gpurun_out/r3d_pytest.txt; cat gpurun_out/r3d_config5_b2.txt gpurun_out/r3d_config5_b16.txt; tail -1 gpurun_out/r3d_bench.txt | cut -c1-300
