#!/bin/bash
set -e
ROOT=$GRAFT_REPO_ROOT
OUT=$ROOT/gpurun_out/r2/pmc_setup
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD --output-format csv -d $OUT/sq -- python3 $ROOT/tools/setup_once.py > $OUT/sq.log 2>&1 || tail -5 $OUT/sq.log
rocprofv3 --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/tcc -- python3 $ROOT/tools/setup_once.py > $OUT/tcc.log 2>&1 || tail -5 $OUT/tcc.log
rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum --output-format csv -d $OUT/tcp -- python3 $ROOT/tools/setup_once.py > $OUT/tcp.log 2>&1 || tail -5 $OUT/tcp.log
find $OUT -name "*counter_collection.csv" | head
echo done
