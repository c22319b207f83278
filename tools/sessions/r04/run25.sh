# GPU session r04/25: the seed sort tests with the extra size thresholds on the shipped kernel; a lines-only fuzz sweep (> 2 000 frames in the reference's seed order)
export TMPDIR=/tmp
O=gpurun_out/r04y; mkdir -p $O
(timeout 240 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/pytest_sort2.log; cat $O/pytest_sort2.log
(timeout 400 python tools/fuzz_gpu.py --only lines --seconds 200 --seed 71 2>&1 | tail -6) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
