# GPU session r06/36: bag-of-words transform up to 8192 descriptors per frame; the BoW tests
export TMPDIR=/tmp
O=gpurun_out/r06bow; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_bow.py -q -x -p no:cacheprovider 2>&1 | tail -6) | tee $O/pytest.log
