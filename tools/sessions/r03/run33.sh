# GPU session 33: final library -- matcher / step tests once more, then a soak of the host-pointer entry points (the call pattern of round 2's page fault)
export TMPDIR=/tmp
O=gpurun_out/r03x13; mkdir -p $O
(timeout 200 python -m pytest tests/test_gpu_match.py tests/test_gpu_bench_step.py tests/test_gpu_guard_page.py -q -p no:cacheprovider -x 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
(timeout 260 python tools/fuzz_gpu.py --soak-calls 30000 --seed 27 2>&1 | tail -3) > $O/soak.log; cat $O/soak.log
