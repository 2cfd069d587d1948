# round 3: the -m gpu suite, smoke() and the latency table on the final commit (after the bucket-order change that followed r03_final.sh)
mkdir -p gpurun_out/r03_final_check
O=gpurun_out/r03_final_check
timeout 2400 python -m pytest tests -m gpu -x -q --durations=8 > $O/gputest.log 2>&1; echo "pytest rc=$?" >> $O/gputest.log; tail -4 $O/gputest.log
python -c "import __graft_entry__ as g; g.build(); g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 600 python tools/latency.py > $O/latency.json 2> $O/latency.err; tail -c 300 $O/latency.json
python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_command.json 2> $O/bench_driver_command.err; tail -c 600 $O/bench_driver_command.json
