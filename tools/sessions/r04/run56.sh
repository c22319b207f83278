# GPU session r04/56: the full GPU suite on the final binary (both growers with the parallel reduce_region_radius)
export TMPDIR=/tmp
ulimit -c 0
mkdir -p gpurun_out/r04s
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > gpurun_out/r04s/pytest.log; cat gpurun_out/r04s/pytest.log
