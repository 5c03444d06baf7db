#!/bin/bash
set -e
mkdir -p gpurun_out/r2
python -m pytest tests/test_multirank_gpu.py tests/test_rccl_selftest_gpu.py -x -q -m gpu > gpurun_out/r2/st_pytest.log 2>&1 || { tail -40 gpurun_out/r2/st_pytest.log; exit 1; }
tail -2 gpurun_out/r2/st_pytest.log
B="python bench.py --no-cpu --no-ksp --k 128 --steps 50 --warmup 5"
for n in 524288 4194304; do
$B --n $n > gpurun_out/r2/st_n${n}_nocomm.json 2> gpurun_out/r2/st_err.txt
$B --n $n --rccl-selftest overlap > gpurun_out/r2/st_n${n}_overlap.json 2>> gpurun_out/r2/st_err.txt
$B --n $n --rccl-selftest serial > gpurun_out/r2/st_n${n}_serial.json 2>> gpurun_out/r2/st_err.txt
done
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r2/st_trace -- python $GRAFT_REPO_ROOT/bench.py --no-cpu --no-ksp --k 128 --steps 5 --warmup 2 --n 524288 --rccl-selftest overlap > /dev/null 2>&1
echo done
