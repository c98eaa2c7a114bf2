#!/bin/bash
# SQ counter passes over single-kernel micro-benchmarks (kbench), to see what bounds the stream-K split kernel
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
K=$R/tools/bin/kbench
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA GRBM_GUI_ACTIVE SQ_WAIT_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INSTS_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS" \
           "SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_WAVES"; do
  i=$((i+1))
  for s in s0 e3 s1; do
    timeout 120 rocprofv3 --kernel-trace --pmc $set -d $R/gpurun_out/pmcH_${s}_$i -o p --output-format csv -- $K conv $s 4 256 20 > $R/gpurun_out/pmcH_${s}_$i.log 2>&1
    echo "$s set $i rc=$?"
  done
done
python3 - <<'PY'
import csv, glob, os, collections
R=os.environ["GRAFT_REPO_ROOT"]
out=[]
for d in sorted(glob.glob(R+"/gpurun_out/pmcH_*_[0-9]")):
    f=glob.glob(d+"/**/*counter_collection.csv", recursive=True)
    if not f: print("no csv in", d); continue
    acc=collections.defaultdict(lambda: [0.0,0])
    for r in csv.DictReader(open(f[0])):
        if "conv_sk_kernel" not in r["Kernel_Name"]: continue
        k=(r["Kernel_Name"][:60], r["Grid_Size"], r["Counter_Name"])
        acc[k][0]+=float(r["Counter_Value"]); acc[k][1]+=1
    for k,v in sorted(acc.items()):
        out.append(f"{os.path.basename(d)},{k[0]},{k[1]},{k[2]},{v[0]/v[1]:.1f},{v[1]}")
open(R+"/gpurun_out/r2h_pmc_summary.csv","w").write("run,kernel,grid,counter,avg_per_launch,launches\n"+"\n".join(out)+"\n")
print("\n".join(out))
PY
