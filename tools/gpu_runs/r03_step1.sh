# round 3, first GPU call: balanced table windows + sleep-polling waits -- parity subset, bench with host CPU time, knock-in study
mkdir -p gpurun_out/r03_step1
O=gpurun_out/r03_step1
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_marlin.py -m gpu -x -q -k "table or aes96 or ops_proofs or chunked_message or seeded or msm_matches or full_size" > $O/pytest_subset.log 2>&1
tail -3 $O/pytest_subset.log
# the driver's command shape, host CPU time beside it
/usr/bin/time -v -o $O/time_default.txt timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline > $O/bench_default.json 2> $O/bench_default.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r03_step1/bench_default.json').read().strip().splitlines()[-1])
print('default', d['value'], d['proofs_verified'], d['roofline']['avg_launch_ms'], d['roofline']['frac'], d['roofline']['one_context_probe'])
PY
grep -E "User time|System time|Elapsed" $O/time_default.txt
ZKAES_WAIT=spin /usr/bin/time -v -o $O/time_spin.txt timeout 900 python bench.py --gpus 1 --steps 8 --warmup 2 --no-cpu-baseline --serial-probe 0 > $O/bench_spin.json 2> $O/bench_spin.err
python -c "
import json;d=json.loads(open('$O/bench_spin.json').read().strip().splitlines()[-1]);print('spin', d['value'], d['proofs_verified'])"
grep -E "User time|System time|Elapsed" $O/time_spin.txt
# knock-in study (each part of the MSM pipeline once more: the drop in blocks/s is its cost in the saturated run); 2048-block message
for k in 0 1 2 4 8; do
  ZKAES_KNOCKIN=$k timeout 600 python bench.py --blocks 2048 --steps 4 --warmup 1 --no-cpu-baseline --serial-probe 0 > $O/bench_knockin_$k.json 2>/dev/null
  python -c "
import json;d=json.loads(open('$O/bench_knockin_$k.json').read().strip().splitlines()[-1]);print('knockin=$k', d['value'], d['proofs_verified'])" | tee -a $O/knockin.txt
done
