# GPU session r04/17: batched staging loads in the pyramid and the matcher kernels
export TMPDIR=/tmp
O=gpurun_out/r04r; mkdir -p $O
(timeout 600 python -m pytest tests/test_gpu_orb.py tests/test_gpu_match.py tests/test_gpu_golden_ref.py tests/test_gpu_bench_step.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench.err | tail -1) > $O/bench.json
python -c "import json; j=json.load(open('$O/bench.json')); s=j['roofline']['stage_ms_per_batch']; print(j['value'], j['ms_per_step'], 'stable:', j['other_seed_order'], 'verified', j['verified_frames']); print(s)" || tail -2 $O/bench.err
