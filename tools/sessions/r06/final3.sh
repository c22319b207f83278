# GPU session r06/final3: the long soaks again on the tree whose every barrier carries the hard LDS wait -- final2.sh's three, plus the concurrent single-frame pairs (10 000 pairs)
bash tools/sessions/r06/final2.sh
O=gpurun_out/r06z
(timeout 900 python tools/soak_concurrent_pairs.py --pairs 10000 2>&1 | tail -6) > $O/soak_concurrent_pairs.log; cat $O/soak_concurrent_pairs.log
