#!/bin/bash
# factorisation kernel A/B: round-1 schedule (SPIKE_FACTOR_OLD=1) vs the look-ahead schedule, K = 64 / 96 / 128
mkdir -p gpurun_out/r2
for k in 64 96 128; do
  for old in 0 1; do
    if [ $old = 1 ]; then export SPIKE_FACTOR_OLD=1; else unset SPIKE_FACTOR_OLD; fi
    echo "K=$k old=$old: $(SPIKE_SETUP_TRACE=1 python tools/setup_once.py $k 2>&1 | grep 'factor' | tail -1)"
  done
done
unset SPIKE_FACTOR_OLD
python -m pytest tests -x -q -m gpu 2>&1 | tail -2
