#!/bin/bash
# 32x4 against 16x8 diagonals-per-wave x waves at K = 128 over system sizes (chain length = N / 256 rows)
mkdir -p gpurun_out/r3
L=gpurun_out/r3/shapes_n.log
: > $L
for n in 1572864 2097152 2621440 3145728 3670016 4194304 4194368 5242880 6291456 8388608; do
  echo "== N=$n" >> $L
  timeout -k 10 300 python tools/ab_apply.py $n 128 0 "32x4_pf2:" "16x8_pf2:ENV.SPIKE_SWEEP_SHAPE=16;8;2" >> $L 2>&1 || exit 1
done
grep -v amdgpu $L | cut -c1-100
