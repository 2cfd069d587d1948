set -x
mkdir -p gpurun_out
( time timeout 2400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 1 --mode strong --blocks 65536 --steps 20 --warmup 1 --no-cpu-baseline --serial-probe 0 ) > gpurun_out/r02_bench_strong_65536_1rank.json 2> gpurun_out/r02_bench_strong_65536_1rank.err; tail -c 900 gpurun_out/r02_bench_strong_65536_1rank.json; tail -4 gpurun_out/r02_bench_strong_65536_1rank.err
