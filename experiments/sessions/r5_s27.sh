#!/bin/bash
# round 5, session 27: the named kernel (conv_up16) by itself under rocprofv3 at 5 and 10 frames per call, idle GPU vs right after matrix-core load
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/s27
( cd $R && timeout 300 python -m pytest tests/test_gpu_b256.py -q -m gpu -x -k "upsampling_streamer" 2>&1 | tail -2 )
for spec in 256:5 256:10; do
  tag=${spec/:/x}
  timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/s27/$tag -o p --output-format csv -- python $R/tools/up16_time.py $spec > $R/gpurun_out/s27_$tag.log 2>&1; echo "rc=$?"
  grep "streams x" $R/gpurun_out/s27_$tag.log
  f=$(find $R/gpurun_out/s27/$tag -name "p_kernel_stats.csv" | head -1); head -1 $f; grep conv_up16 $f
  cp $f $R/gpurun_out/s27_up16_alone_${tag}_kernel_stats.csv
done
python $R/tools/up16_time.py hot 256:5 2>&1 | grep "streams x"
python $R/tools/up16_time.py 256:5 256:1 2>&1 | grep "streams x"
find $R/gpurun_out/s27 -name "*.db" -delete 2>/dev/null; find $R/gpurun_out/s27 -name "p_kernel_trace.csv" -delete 2>/dev/null
