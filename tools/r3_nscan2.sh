#!/bin/bash
# chain-count sweep of the k_nscan_* path at several sizes
mkdir -p gpurun_out/r3
L=gpurun_out/r3/nscan_chains.log
: > $L
for k in 1 2 3; do
for n in 1048576 4194304 16777216; do
  echo "== K=$k N=$n" >> $L
  timeout -k 10 300 python tools/ab_apply.py $n $k 0 "c512:ENV.SPIKE_AUTO_CHAINS=512" "c1024:ENV.SPIKE_AUTO_CHAINS=1024" "c2048:ENV.SPIKE_AUTO_CHAINS=2048" "c4096:ENV.SPIKE_AUTO_CHAINS=4096" "c8192:ENV.SPIKE_AUTO_CHAINS=8192" >> $L 2>&1 || exit 1
done; done
grep -v amdgpu.ids $L | cut -c1-150
