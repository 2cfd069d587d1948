#!/bin/bash
# round 6: the other bench shapes on the final code (one GPU): one rank's configs[3] share through a 1-rank RCCL gather, configs[4] (1,024 single-block proofs),
# configs[1] (64 blocks), --chunk 4, and the lone-call latency tool.  ~12 minutes.
cd /tmp && export TMPDIR=/tmp && cd - >/dev/null
O=gpurun_out
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode strong --blocks 8192 --steps 4 --warmup 1 --no-cpu-baseline > $O/r06_bench_strong_1rank_8192.json 2> $O/r06_bench_strong.err
timeout 900 python bench.py --mode batch --proofs 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/r06_bench_batch_1024.json 2> $O/r06_bench_batch.err
timeout 600 python bench.py --blocks 64 --steps 2 --warmup 1 --no-cpu-baseline > $O/r06_bench_64_blocks.json 2> $O/r06_bench_64.err
timeout 900 python bench.py --chunk 4 --blocks 1024 --steps 4 --warmup 1 --no-cpu-baseline > $O/r06_bench_chunk4.json 2> $O/r06_bench_chunk4.err
timeout 600 python tools/latency.py > $O/r06_latency.json 2> $O/r06_latency.err
for f in r06_bench_strong_1rank_8192 r06_bench_batch_1024 r06_bench_64_blocks r06_bench_chunk4; do python -c "import json,sys; d=json.load(open('gpurun_out/$f.json')); print('$f', d['value'], d['unit'], d['proofs_verified'])"; done; cat $O/r06_latency.json
