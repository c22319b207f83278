O=gpurun_out/r03o; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_orb.py tests/test_gpu_golden_ref.py tests/test_gpu_facade.py tests/test_gpu_bench_step.py -m gpu -x -q 2>&1 | tail -3) > $O/tests.log; cat $O/tests.log
(timeout 90 python tools/fuzz_gpu.py --only orb --seconds 60 --seed 58 2>&1 | grep "orb:") >> $O/tests.log; tail -1 $O/tests.log
timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 16 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print(j['value'], j['ms_per_step'], j['verified_frames'], s)" | tee -a $O/tests.log
