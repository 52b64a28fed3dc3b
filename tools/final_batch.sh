#!/bin/bash
# final measurements of the round (run from the repo root on the GPU box)
python bench.py > gpurun_out/r05_bench_default.json 2> gpurun_out/r05_bench_default.err; tail -c 600 gpurun_out/r05_bench_default.json
tools/bench_all.sh gpurun_out/r05_bench_all_workloads.txt > /dev/null 2>&1; cat gpurun_out/r05_bench_all_workloads.txt | cut -c1-200
for W in C2 C2r C3 C4 C5 C2sbsr; do tools/profile.sh r05_final_$(echo $W | tr A-Z a-z) --workload $W --steps 10 --warmup 3 > /dev/null 2>&1; done
ls gpurun_out/profile/
python -m pytest tests/test_gpu_parity_report.py -q 2>&1 | tail -2
