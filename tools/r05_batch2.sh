#!/bin/bash
# round 5, GPU batch 2: VOP2->VOP3 select peephole A/B on every workload family
O=gpurun_out/r05_pp_ab.txt; : > $O
for W in "C3" "C3 --content random" "C3s" "C3r" "C2" "C2 --content random" "C5" "C2r" "C4"; do
  echo "== workload $W" | tee -a $O
  tools/abn.sh 2 ab/base.so ab/pp.so -- --workload $W 2>&1 | tee -a $O
done
