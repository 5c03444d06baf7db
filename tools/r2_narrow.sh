#!/bin/bash
# narrow and medium bands with auto partitions: N = 8M and N = 4M, K = 2..64 (+ BASELINE config 2)
mkdir -p gpurun_out/r2
for n in 8388608 4194304; do for k in 2 4 8 16 32 64; do
  python bench.py --n $n --k $k --steps 100 --warmup 10 --no-cpu --no-ksp > gpurun_out/r2/nb_n${n}_k$k.json 2>/dev/null
done; done
python bench.py --n 1048576 --k 32 --partitions 64 --steps 200 --warmup 20 --no-cpu --no-ksp > gpurun_out/r2/nb_c2.json 2>/dev/null
