# GPU session r06/final4: the long form of the soaks on the final tree -- 20 minutes of overlapped steps over the eight step shapes, 15 minutes of the mixed fuzz sweep
# (ORB / lines / matchers / aux families), 30 000 concurrent single-frame pairs
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
timeout 1500 python tools/soak_seed_sort.py --minutes 20 --seed 7 > $O/soak_seed_sort_long.log 2>&1; tail -3 $O/soak_seed_sort_long.log
(timeout 1100 python tools/fuzz_gpu.py --seconds 780 --aux-seconds 120 --seed 104 2>&1 | tail -8) > $O/fuzz_long.log; cat $O/fuzz_long.log
(timeout 1500 python tools/soak_concurrent_pairs.py --pairs 30000 --minutes 14 2>&1 | tail -4) > $O/soak_concurrent_pairs_long.log; cat $O/soak_concurrent_pairs_long.log
