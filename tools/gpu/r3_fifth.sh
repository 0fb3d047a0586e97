#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_disc.py tests/test_gpu_train.py -m gpu -q 2>&1 | tail -12 > gpurun_out/r3e_pytest.txt
python tools/config5.py 2 3 --gan 2>&1 | grep step > gpurun_out/r3e_config5_gan_b2.txt
cd /tmp && export TMPDIR=/tmp && timeout 400 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_c5 -o c5 -- python $GRAFT_REPO_ROOT/tools/config5.py 2 4 > $GRAFT_REPO_ROOT/gpurun_out/r3e_prof_c5.log 2>&1
cd $GRAFT_REPO_ROOT
tail -5 gpurun_out/r3e_pytest.txt; cat gpurun_out/r3e_config5_gan_b2.txt
head -45 gpurun_out/r3e_prof_c5/c5_kernel_stats.csv | cut -c1-160
