run() { python bench.py --steps 4 --warmup 2 --no-cpu-baseline --schedule pipelined 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$1', round(d['value'],1), round(d['ms_per_step']), {k:round(v) for k,v in d['stage_ms_per_step'].items()})"; }
python -c "import torch; print('priority range', torch.cuda.Stream.priority_range() if hasattr(torch.cuda.Stream,'priority_range') else 'n/a')"
run base
CCVS_PIPELINE_PRIORITIES=0,0 run prio_0_0
CCVS_PIPELINE_PRIORITIES=-1,-1 run prio_-1_-1
CCVS_PIPELINE_PRIORITIES=0,-1 run prio_0_-1
GPU_MAX_HW_QUEUES=8 run hwq8
GPU_MAX_HW_QUEUES=2 run hwq2
GPU_MAX_HW_QUEUES=1 run hwq1
HIP_FORCE_DEV_KERNARG=1 run dev_kernarg
HSA_ENABLE_SDMA=0 run nosdma
