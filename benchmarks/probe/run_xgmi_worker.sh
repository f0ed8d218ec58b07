#!/bin/bash
# usage: run_xg.sh <world> <order>
export NAF_ROOT=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 NAF_XGMI_TEST_ORDER=$2
python -m torch.distributed.run --nnodes=1 --nproc-per-node=$1 --master-addr 127.0.0.1 --master-port $((20000 + RANDOM % 20000)) tests/xgmi_worker.py 2>&1 | grep -E "DIFF|XGMI_OK|AssertionError" | head -14
