echo base; timeout 200 python tools/conv_bench.py --shape s0,s1,e2 --impl 6 2>&1 | grep -v amdgpu
for d in 1 4 7; do echo "DBG=$d"; ADK_LIB_PATH=$PWD/tools/bin/libadk_sk$d.so timeout 200 python tools/conv_bench.py --shape s0,s1,e2 --impl 6 2>&1 | grep -v amdgpu; done
