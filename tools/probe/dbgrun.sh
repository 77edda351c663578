for d in 0 4 1; do echo "SG_DBG=$d"; SG_DBG=$d python tools/bench_conv.py res3x3 mask3x3 D3_ 2>&1 | grep -v amdgpu | cut -c1-132; done
