#!/bin/bash
# A/B of gemm_x3p.hip variants on the same box
echo "current:"; python tools/x3p_ab.py 2>&1 | grep -v amdgpu | tail -10
cp speech2text_amd/csrc/gemm_x3p.hip /tmp/cur.hip
cp tools/debug/gemm_x3p_0211.hip.txt speech2text_amd/csrc/gemm_x3p.hip
python -m speech2text_amd.csrc.build 2>&1 | tail -1
echo "commit 0211:"; python tools/x3p_ab.py 2>&1 | grep -v amdgpu | tail -1
cp /tmp/cur.hip speech2text_amd/csrc/gemm_x3p.hip
