#!/bin/bash
# narrow bands, auto partitions: one A/B process per K (twisted + fp32 tail vs round-2 behaviour), then plain bench lines
mkdir -p gpurun_out/r3
python tools/ab_apply.py 16777216 1 0 "k1:" > gpurun_out/r3/narrow_ab.log 2>&1
for k in 2 3 4 8 16; do
  python tools/ab_apply.py 8388608 $k 0 "tw:" "r2:twist=off,spike_fp32=off,spike_tol=1e-16" >> gpurun_out/r3/narrow_ab.log 2>&1
done
python bench.py --n 16777216 --k 1 --steps 100 --warmup 10 --no-cpu --no-ksp > gpurun_out/r3/nb_k1.json 2>/dev/null
for k in 2 4 8 16; do python bench.py --n 8388608 --k $k --steps 100 --warmup 10 --no-cpu --no-ksp > gpurun_out/r3/nb_k$k.json 2>/dev/null; done
grep -v amdgpu.ids gpurun_out/r3/narrow_ab.log
python tools/show_bench.py gpurun_out/r3/nb_k*.json
