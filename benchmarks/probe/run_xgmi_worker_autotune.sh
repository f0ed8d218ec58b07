export NAF_ROOT=$PWD HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2 NAF_DP_EXCHANGE=auto NAF_AUTOTUNE_DEBUG=1
python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29533 tests/xgmi_worker.py 2>&1 | grep "autotune\|naf\]\|XGMI_OK\|Error" | head -40
