# GPU session 2 of round 3: co-run priority A/B, occupancy experiment at 4096 frames per step, staging-slack soak.
O=gpurun_out/r03b; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
timeout 200 python -m pytest tests/test_gpu_bench_step.py -m gpu -q 2>&1 | tail -3 > $O/step_test.log
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 $2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 $3', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
{
for pass in 1 2; do for v in base prio2 prio3; do B $v; done; done
B base "--batch 4096"
PLP_LSD_RING=256 B occ4 "--batch 4096" ring256
PLP_LSD_RING=128 B occ4 "--batch 4096" ring128
PLP_LSD_RING=128 B base "--batch 4096" ring128
PLP_LSD_RING=256 B occ4 "" ring256
} > $O/ab.log 2>&1
cp build_exp/noslack.so $L
(timeout 200 python tools/fuzz_gpu.py --soak-calls 30000 --seed 31 2>&1 | tail -6) > $O/soak_noslack.log
cp build_exp/.orig.so $L
cat $O/step_test.log $O/ab.log $O/soak_noslack.log
