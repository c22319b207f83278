# GPU session r06/25: ORB at K = 2000 on ONE level (quota 2000: smaller radix / key block in k_quadtree), the ORB tests
export TMPDIR=/tmp
O=gpurun_out/r06y; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_orb.py tests/test_gpu_golden_ref.py -q -x -p no:cacheprovider 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
