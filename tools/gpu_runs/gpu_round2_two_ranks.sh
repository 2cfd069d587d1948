# rehearsal of the N = 2 rank path with the REAL prover on a one-GPU box: two gloo ranks sharing device 0 (bench.py's ZKAES_BENCH_BACKEND / ZKAES_BENCH_ONE_GPU hooks)
set -x
mkdir -p gpurun_out
export ZKAES_BENCH_BACKEND=gloo ZKAES_BENCH_ONE_GPU=1
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 2 --blocks 512 --steps 4 --warmup 1 --contexts 6 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_two_ranks_weak.json 2> gpurun_out/r02_bench_two_ranks_weak.err; tail -c 400 gpurun_out/r02_bench_two_ranks_weak.json; tail -3 gpurun_out/r02_bench_two_ranks_weak.err
timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 2 --mode strong --blocks 1024 --steps 4 --warmup 1 --contexts 6 --no-cpu-baseline --serial-probe 0 > gpurun_out/r02_bench_two_ranks_strong.json 2> gpurun_out/r02_bench_two_ranks_strong.err; tail -c 400 gpurun_out/r02_bench_two_ranks_strong.json; tail -3 gpurun_out/r02_bench_two_ranks_strong.err
