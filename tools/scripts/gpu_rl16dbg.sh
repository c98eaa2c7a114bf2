echo base; timeout 200 python tools/conv_bench.py --shape s3,s2 --impl 5 2>&1 | grep -v amdgpu
for d in 1 2 4 8 3 15; do echo "DBG=$d"; ADK_LIB_PATH=$PWD/tools/bin/libadk_dbg$d.so timeout 200 python tools/conv_bench.py --shape s3,s2 --impl 5 2>&1 | grep -v amdgpu; done
