# GPU session 38: full GPU suite on the final tree
export TMPDIR=/tmp
O=gpurun_out/r03x18; mkdir -p $O
(timeout 400 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -4) > $O/pytest.log; cat $O/pytest.log
