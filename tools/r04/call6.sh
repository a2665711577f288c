#!/bin/bash
# round 4, GPU call 6: where the per-tile fixed cost goes (dispatch alone / + set-up and first chunk / + step barriers), WPC2 again
cd "${GRAFT_REPO_ROOT:-.}"; O=gpurun_out/r04; mkdir -p $O; rm -f $O/ablate6.txt $O/wpc2b.txt
for shape in "195 128 3 256 120" "49 128 3 256 120" "128 64 3 256 120" "64 32 3 256 120"; do
  for ab in 0 32768 4096 16387 3; do
    echo -n "ablate=$ab  " >> $O/ablate6.txt
    CCVS_CONV_ABLATE=$ab timeout 120 python tools/conv_one.py $shape 2>&1 | tail -1 >> $O/ablate6.txt
  done
done
cat $O/ablate6.txt
for w in 0 112; do
  for shape in "49 128 3 256 120" "99 128 3 256 120" "49 128 3 256 240" "99 128 3 256 240"; do
    echo -n "WPC2=$w  " >> $O/wpc2b.txt
    CCVS_CONV_WPC2=$w timeout 120 python tools/conv_one.py $shape 2>&1 | tail -1 >> $O/wpc2b.txt
  done
done
cat $O/wpc2b.txt
