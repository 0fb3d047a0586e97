#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu/r2_ring2.sh r2i
bash tools/gpu/r2_abl.sh
