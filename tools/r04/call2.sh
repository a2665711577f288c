#!/bin/bash
# round 4, GPU call 2: (a) what a convolution loses to beside ONE kind of background load (tools/conv_contention_probe.py);
# (b) where a tile's time goes by shape: full / no MFMA / no staging no MFMA / + no epilogue (CCVS_CONV_ABLATE)
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O
timeout 600 python tools/conv_contention_probe.py 120 > $O/contention.txt 2>&1
tail -80 $O/contention.txt
for shape in "195 128 3 256 120" "49 128 3 256 120" "99 128 3 256 120" "128 64 3 256 120" "64 32 3 256 120"; do
  for ab in 0 2 3 16387 131 16515 128 256; do
    echo -n "ablate=$ab  " >> $O/ablate.txt
    CCVS_CONV_ABLATE=$ab timeout 120 python tools/conv_one.py $shape 2>&1 | tail -1 >> $O/ablate.txt
  done
done
cat $O/ablate.txt
