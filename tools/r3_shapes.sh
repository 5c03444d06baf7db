#!/bin/bash
# sweep shapes for K > 64 re-measured with the round-3 kernels: diagonals per wave x waves x bundles in flight (one process per case)
mkdir -p gpurun_out/r3
L=gpurun_out/r3/shapes.log
: > $L
run() { echo "== N=$1 K=$2" >> $L; shift 0; timeout -k 10 300 python tools/ab_apply.py "$@" >> $L 2>&1 || exit 1; }
run 4194304 128 0 "32x4_pf2:" "16x8_pf2:ENV.SPIKE_SWEEP_SHAPE=16;8;2" "32x4_pf4:ENV.SPIKE_SWEEP_SHAPE=32;4;4"
run 524288 128 0 "32x4_pf2:" "16x8_pf2:ENV.SPIKE_SWEEP_SHAPE=16;8;2" "16x8_pf4:ENV.SPIKE_SWEEP_SHAPE=16;8;4"
run 1048576 128 0 "32x4_pf2:" "16x8_pf2:ENV.SPIKE_SWEEP_SHAPE=16;8;2"
run 4194304 96 0 "32x3_pf2:" "16x6_pf2:ENV.SPIKE_SWEEP_SHAPE=16;6;2" "16x6_pf4:ENV.SPIKE_SWEEP_SHAPE=16;6;4"
run 4194304 192 0 "32x6_pf2:" "16x12_pf2:ENV.SPIKE_SWEEP_SHAPE=16;12;2" "16x12_pf3:ENV.SPIKE_SWEEP_SHAPE=16;12;3"
run 4194304 256 0 "32x8_pf2:" "16x16_pf2:ENV.SPIKE_SWEEP_SHAPE=16;16;2" "16x16_pf3:ENV.SPIKE_SWEEP_SHAPE=16;16;3"
grep -v amdgpu $L | cut -c1-110
