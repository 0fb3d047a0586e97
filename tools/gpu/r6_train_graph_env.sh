#!/bin/bash
# round 6: the config-5 training step as a HIP graph under the runtime's graph knobs (eager | graph ms per step)
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
out=gpurun_out/r06_train_graph_env.txt
: > $out
run() {
  echo "$*: $(env "$@" timeout 300 python bench.py --extra-configs-worker --steps 5 --warmup 2 2>/dev/null | python -c "
import sys, json
for l in sys.stdin:
    if l.startswith('['):
        r = json.loads(l)
        print('config4', r[0].get('ms_per_step'), '| train eager', r[1].get('ms_per_step'), 'graph', r[1].get('ms_per_step_hip_graph'), '| adversarial', r[2].get('ms_per_step'))
")" >> $out
}
run X=0
run DEBUG_HIP_FORCE_GRAPH_QUEUES=1
run DEBUG_HIP_FORCE_GRAPH_QUEUES=2
run DEBUG_HIP_FORCE_GRAPH_QUEUES=4
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1024
run GPU_MAX_HW_QUEUES=8
cat $out
