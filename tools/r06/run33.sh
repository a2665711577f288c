#!/bin/bash
# round 6, GPU call 33: the whole GPU suite and smoke() on the round's final commit
cd /root/repo
O=gpurun_out/r06ag; mkdir -p $O
timeout 600 python3 -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?" >> $O/smoke.log; tail -2 $O/smoke.log
timeout 1500 python3 -m pytest tests -x -q -m gpu 2>&1 | tail -6 > $O/gpu_tests.log; cat $O/gpu_tests.log
