#!/bin/bash
# per-launch tables under environment variants:  r2_tables.sh TAG "VAR=a VAR2=b" "VAR=c" ...   ('-' = default environment)
cd "$GRAFT_REPO_ROOT" || exit 1
TAG=$1; shift
mkdir -p gpurun_out
i=0
for cfg in "$@"; do
  if [ "$cfg" = "-" ]; then envs=""; else envs="$cfg"; fi
  env $envs timeout 200 python tools/launch_table.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_table$i.txt
  echo "== [$cfg] $(tail -1 gpurun_out/${TAG}_table$i.txt)"
  i=$((i+1))
done
