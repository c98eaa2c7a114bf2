for tt in 0 160 128 96 64; do echo "TT=$tt"; ADK_RL16_TT=$tt timeout 200 python tools/conv_bench.py --shape s3,s2,e0 --impl 5 2>&1 | grep -v amdgpu; done
