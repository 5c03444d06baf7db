#!/bin/bash
# four-rows-per-lane scan (k_nscan_*) against the paths it replaces, one process per K; chain-count sweep
mkdir -p gpurun_out/r3
L=gpurun_out/r3/nscan_ab.log
: > $L
timeout -k 10 300 python tools/ab_apply.py 8388608 2 0 "nscan:" "tiles:narrow_scan_kmax=1" "ns2048:ENV.SPIKE_AUTO_CHAINS=2048" "ns4096:ENV.SPIKE_AUTO_CHAINS=4096" "ns16384:ENV.SPIKE_AUTO_CHAINS=16384" >> $L 2>&1 &&
timeout -k 10 300 python tools/ab_apply.py 8388608 3 0 "nscan:" "tiles:narrow_scan_kmax=1" "ns2048:ENV.SPIKE_AUTO_CHAINS=2048" "ns4096:ENV.SPIKE_AUTO_CHAINS=4096" >> $L 2>&1 &&
timeout -k 10 300 python tools/ab_apply.py 16777216 1 0 "rows4:" "rows1:narrow_scan_rows=1" "r4c4096:ENV.SPIKE_AUTO_CHAINS=4096" "r4c16384:ENV.SPIKE_AUTO_CHAINS=16384" >> $L 2>&1
grep -v amdgpu.ids $L
