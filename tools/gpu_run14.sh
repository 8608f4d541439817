#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
bash tools/gpu_run12.sh 2>&1 | tail -8
bash tools/gpu_run13.sh 2>&1 | head -8
