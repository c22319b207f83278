# GPU session r05/final2: the closing measurements again on the final tree (after the late changes: acceptance block, blur-tile and FAST staging, pyramid offsets, launch-constant divisions as multipliers) -- full GPU suite, round profile
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest2.log; cat $O/pytest2.log
bash tools/run_prof.sh r05z > $O/run_prof.log 2>&1; tail -2 $O/run_prof.log | cut -c1-300
(timeout 200 python tools/fuzz_gpu.py --seconds 120 --seed 93 2>&1 | tail -4) > $O/fuzz2.log; cat $O/fuzz2.log
