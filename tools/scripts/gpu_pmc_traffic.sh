cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 150 rocprofv3 --kernel-trace --pmc $c -d $R/gpurun_out/pmc17_$c -o p --output-format csv -- python $R/bench.py --steps 6 --warmup 2 --serial --no-cpu-baseline --no-other-precision --no-op-profile > $R/gpurun_out/pmc17_$c.log 2>&1
  echo "$c rc=$?"
done
ls $R/gpurun_out | grep pmc17
