# GPU session r06/final2: long soaks on the final tree -- the shipped sort inside the overlapped step over eight step shapes (tools/soak_seed_sort.py, 8 minutes), a line-only fuzz sweep (5 minutes, random shapes / both growers), ORB host-entry soak (30 000 calls)
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
timeout 700 python tools/soak_seed_sort.py --minutes 8 --seed 6 > $O/soak_seed_sort.log 2>&1; tail -10 $O/soak_seed_sort.log
(timeout 400 python tools/fuzz_gpu.py --seconds 300 --seed 102 --only lines 2>&1 | tail -4) > $O/fuzz_lines.log; cat $O/fuzz_lines.log
(timeout 900 python tools/fuzz_gpu.py --soak-calls 30000 --seed 103 2>&1 | tail -4) > $O/soak_orb.log; cat $O/soak_orb.log
