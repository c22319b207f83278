# GPU session r04/55: reduce_region_radius as a rank pairing in k_lsd_grow_mw too (single-frame path): parity, fuzz through 1 / 3 / 5 / 8 waves, latency
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04t; mkdir -p $O
(timeout 200 python -m pytest tests/test_gpu_line.py tests/test_gpu_golden_ref.py tests/test_gpu_facade.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(PLP_LSD_RING=64 timeout 200 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/pytest_ring64.log; cat $O/pytest_ring64.log
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 40 --seed 95 2>&1 | grep "lines:") > $O/fuzz.log; cat $O/fuzz.log
(timeout 100 python tools/experiments/latency_profile.py 2>&1 | grep -v amdgpu.ids | tail -4) > $O/latency.log; cat $O/latency.log
