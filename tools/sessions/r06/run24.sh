# GPU session r06/24: 1920 x 1080 frames through the line front end (65 KB of LDS for the USED bitmap), the line tests
export TMPDIR=/tmp
O=gpurun_out/r06x; mkdir -p $O
(timeout 1200 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
