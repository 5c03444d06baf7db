#!/bin/bash
# full GPU suite, then the headline at caller-chosen partition counts (subsplit picks the chains)
set -e
mkdir -p gpurun_out/r2
python -m pytest tests -x -q -m gpu > gpurun_out/r2/full_pytest.log 2>&1 || { tail -40 gpurun_out/r2/full_pytest.log; exit 1; }
tail -2 gpurun_out/r2/full_pytest.log
B="python bench.py --no-cpu --no-ksp --k 128 --steps 20 --warmup 3"
for p in 0 8 64 96 384 640; do
  $B --partitions $p > gpurun_out/r2/full_p$p.json 2> gpurun_out/r2/full_err.txt
done
echo done
