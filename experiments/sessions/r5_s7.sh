#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
bash tools/ab_multi.sh s7 ADK_RB16_FEW128 2 128 256
