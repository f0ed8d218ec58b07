run() { echo "== $*"; env "$@" timeout -k 10 120 python benchmarks/host_api_steps.py 64 2>/dev/null | grep -o "host-API path: [0-9]* timesteps/s"; env "$@" timeout -k 10 120 python bench.py --steps 600 --warmup 50 --no-extras --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('bench', d['value'])"; }
run X=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=1
run DEBUG_CLR_GRAPH_PACKET_CAPTURE=0
run HIP_FORCE_DEV_KERNARG=1
run HIP_FORCE_DEV_KERNARG=0
run AMD_OPT_FLUSH=0
run AMD_OPT_FLUSH=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=1
run DEBUG_HIP_GRAPH_BATCH_SIZE=64
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run GPU_FLUSH_ON_EXECUTION=1
run X=1
