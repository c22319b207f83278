# GPU session r06/47: the 4-wave sort configuration at the other BASELINE geometries (batches of 264 frames at 752 x 480, 1241 x 376, 320 x 240 against the oracle)
export TMPDIR=/tmp
O=gpurun_out/r06geo; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider -k "geometries or large_batch" 2>&1 | tail -6) | tee $O/pytest.log
