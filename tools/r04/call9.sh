#!/bin/bash
# round 4, GPU call 9: warp_fuse_blend variants; packed (P8) activations in the pipelined run; Kinetics / Drums lines with the dense
# prefill GEMM and more token chains
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O; rm -f $O/fuse9.txt $O/p8_9.txt $O/kin9.txt
for cch in 8 4; do for pf in 0 1; do
  echo -n "CCH=$cch PREF=$pf  " >> $O/fuse9.txt
  CCVS_FUSE_CCH=$cch CCVS_FUSE_PREF=$pf timeout 200 python tools/mem_bench.py 2>&1 | grep warp_fuse >> $O/fuse9.txt
done; done
cat $O/fuse9.txt
line() { python - "$1" <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split('/')[-1], "fps",round(d["value"],2),"stages",{k:round(v) for k,v in d["stage_ms_per_step"].items()},"batch",d["config"].get("batch_per_gpu"),"steps",d["steps"])
except Exception as e: print(sys.argv[1],"failed",e)
PY
}
for p8 in 0 1; do
  CCVS_CONV_P8=$p8 timeout 400 python bench.py --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/bench9_p8_$p8.json 2> $O/bench9_p8_$p8.err
  line $O/bench9_p8_$p8.json | tee -a $O/p8_9.txt
done
for ch in 2 3 4; do
  timeout 500 python bench.py --config kinetics --batch 64 --steps 6 --warmup 2 --chains $ch --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/bench9_kin_c$ch.json 2> $O/bench9_kin_c$ch.err
  line $O/bench9_kin_c$ch.json | tee -a $O/kin9.txt
done
for b in 4 8; do
  timeout 600 python bench.py --config drums --batch $b --steps 3 --warmup 1 --no-cpu-baseline --no-strict-f32 --no-encode-cond-leg > $O/bench9_drums_b$b.json 2> $O/bench9_drums_b$b.err
  line $O/bench9_drums_b$b.json | tee -a $O/kin9.txt
done
