#!/bin/bash
# round 5, GPU batch 1: issue-rate microbenchmark (new rows) + interleaved A/B of the EASU scheduling variants
mkdir -p gpurun_out
timeout 120 tools/ubench/valu_rates > gpurun_out/r05_valu_rates.txt 2>&1
tail -25 gpurun_out/r05_valu_rates.txt
timeout 900 tools/abn.sh 3 ab/base.so ab/fsb.so ab/fsb5.so ab/mme.so > gpurun_out/r05_sched_ab_1.txt 2>&1
cat gpurun_out/r05_sched_ab_1.txt
