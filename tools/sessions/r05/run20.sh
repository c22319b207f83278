# GPU session r05/20: long soak of the shipped seed sort over eight step shapes (tools/soak_seed_sort.py, 10 minutes)
export TMPDIR=/tmp
O=gpurun_out/r05t; mkdir -p $O
(timeout 900 python tools/soak_seed_sort.py --minutes 10 --seed 5 2>&1 | grep -v amdgpu.ids | tail -12) > $O/soak_seed_sort.log; cat $O/soak_seed_sort.log
