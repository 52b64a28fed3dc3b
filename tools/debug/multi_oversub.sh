#!/bin/bash
# tools/debug/multi_oversub.sh -- the round driver's N = 2 commands on real shards of a 1-GPU box (both on device 0, --oversubscribe): direct and under torchrun
echo "== direct --gpus 2 --oversubscribe"; python bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 2>gpurun_out/m1.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['per_device_ms_per_step'], d['parity_check'].get('ok'), d['parity_check'].get('per_shard'), d['config'].get('oversubscribed'))"
tail -3 gpurun_out/m1.err
echo "== torchrun 2 ranks"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --oversubscribe --steps 5 --warmup 2 2>gpurun_out/m2.err | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['n_gpus'], d['config']['per_device_ms_per_step'], d['parity_check'].get('ok'), d['parity_check'].get('per_shard'), d['config'].get('oversubscribed'), 'cpu', d.get('cpu_baseline'))"
tail -3 gpurun_out/m2.err
echo "== torchrun 2 ranks, short line"; python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29518 bench.py --gpus 2 --oversubscribe --steps 2 --warmup 1 --no-cpu --no-extras --pmc off 2>gpurun_out/m3.err | tail -1 | cut -c1-300
