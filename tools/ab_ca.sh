#!/bin/bash
# same-box A/B of the fused cross-attention kernel: the committed library against lidar-vision-vqa_amd/liblvq_hip_exp.so (an experimental build
# of the same ABI, selected through LVQ_LIB_PATH): kernel tests on the experimental build, then the in-kernel phase timeline of both, interleaved.
# On the GPU box: bash tools/ab_ca.sh
EXP=$(pwd)/lidar-vision-vqa_amd/liblvq_hip_exp.so
LVQ_LIB_PATH=$EXP python3 -m pytest tests/test_gpu_ca_fused.py -x -q -m gpu 2>&1 | tail -3
for i in 1 2; do
  echo "== base"; WARM=50 python3 tools/stamps_ca.py | grep -v "^per-phase"
  echo "== exp";  WARM=50 LVQ_LIB_PATH=$EXP python3 tools/stamps_ca.py | grep -v "^per-phase"
done
