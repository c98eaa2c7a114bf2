#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
for v in "new" "ADK_CONV_MAX_SPLIT=0" "ADK_RL16_FEW=0" "ADK_RL16_FEW=0 ADK_CONV_MAX_SPLIT=0"; do
  tag=$(echo "$v" | tr ' =' '__')
  if [ "$v" = "new" ]; then e=""; else e="$v"; fi
  env $e ADK_SPLIT16=1 python tools/op_profile.py libritts_sym 1 80 > gpurun_out/r2u_ops_$tag.txt 2>/dev/null
  echo "$v: $(tail -1 gpurun_out/r2u_ops_$tag.txt)"
done
paste <(cut -c1-92 gpurun_out/r2u_ops_new.txt) <(cut -c70-92 gpurun_out/r2u_ops_ADK_RL16_FEW_0_ADK_CONV_MAX_SPLIT_0.txt)
