# GPU session r04/16: the blur tiles issue all their staging loads before the first wait (k_blur7, k_blur_sobel, k_blur_half)
export TMPDIR=/tmp
O=gpurun_out/r04p; mkdir -p $O
(timeout 400 python -m pytest tests/test_gpu_orb.py tests/test_gpu_line.py tests/test_gpu_golden_ref.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(timeout 200 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/bench.err | tail -1) > $O/bench.json
python -c "import json; j=json.load(open('$O/bench.json')); s=j['roofline']['stage_ms_per_batch']; print(j['value'], j['ms_per_step'], 'stable:', j['other_seed_order'], 'verified', j['verified_frames']); print(s)" || tail -2 $O/bench.err
