#!/bin/bash
# second visit: stage count chosen per instantiation (release) vs two stages everywhere (nst2); AERO_CONV_KMIN192=192 on top
# the variant libraries are experiment builds of part 2 (not tracked): python -c "import __graft_entry__ as g; g.build_library(out='tools/dbg/libaero_glds_nst2.so',
#   defines=['AERO_GLDS_NST=2'], objdir='aero_amd/csrc/build/nst2', only_parts=[2], force=True)"  (nst4: AERO_GLDS_NST=4; at the time of the first visit a second
#   macro, AERO_GLDS_NST64, set the stage count of the 64-channel-chunk tiles separately: nst3_64x2 = 3 / 2)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for rep in 1 2; do
for v in release nst2 kmin192; do
  lib=aero_amd/libaero_hip.so; [ $v = nst2 ] && lib=tools/dbg/libaero_glds_$v.so
  k=384; [ $v = kmin192 ] && k=192
  echo "== $v (pass $rep)"
  AERO_CONV_KMIN192=$k AERO_HIP_LIB=$PWD/$lib timeout 200 python tools/launch_table.py 2>&1 | grep "glds\|sum of"
done
done
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_model.py -m gpu -q -x -p no:cacheprovider -k "conv or golden or batch64" 2>&1 | tail -3
for v in release nst2 release nst2; do
  lib=aero_amd/libaero_hip.so; [ $v != release ] && lib=tools/dbg/libaero_glds_$v.so
  AERO_HIP_LIB=$PWD/$lib timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-configs --no-kernel-events 2>/dev/null | grep '^{' | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('bench $v', d['ms_per_step'], d['config'].get('ms_per_step_one_at_a_time'))"
done
