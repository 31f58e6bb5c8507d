#!/bin/bash
# Does a second group of a workgroup run faster than its first?  Single-net gradient kernels (PIME_PPO_DUAL=0):
# B = 65536 on 256 workgroups (1 group each), B = 131072 / 196608 on 256 workgroups (2 / 3 groups each, the whole chip busy throughout),
# and B = 65536 on 128 workgroups (PIME_FUSED_GRID=128: 2 groups each on half of the chip)
OUT=gpurun_out; mkdir -p $OUT
for b in 65536 131072 196608; do
  echo "== PPO single kernels, B=$b on 256 workgroups"
  GRAD_AB_B=$b PIME_PPO_DUAL=0 timeout -k 10 200 bash tools/kstats.sh tools/grad_ab.py modular 128 | head -3
done
echo "== PPO dual kernel, B=65536 / 131072"
for b in 65536 131072; do GRAD_AB_B=$b timeout -k 10 200 bash tools/kstats.sh tools/grad_ab.py modular 128 | head -2; done
