# GPU session r05/4: seed sort after the swap pass was reordered (own entry loaded beside the partner position) and v_mbcnt ranks; frames per step 2048 / 3072 / 4096
# with the 113-VGPR grower; the round's phase clocks again (rprof build)
export TMPDIR=/tmp
O=gpurun_out/r05d; mkdir -p $O
(timeout 300 python -m pytest tests/test_gpu_seed_sort.py tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -1) > $O/pytest.log; cat $O/pytest.log
B() {
  (timeout 300 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 $2 2> $O/bench_$1.err | tail -1) > $O/bench_$1.json
  python -c "import json; j=json.load(open('$O/bench_$1.json')); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'order', round(s['lsd_order'],2), 'grow', round(s['lsd_grow'],2))" || tail -2 $O/bench_$1.err
}
B b2048 "--batch 2048"
B b3072 "--batch 3072"
B b4096 "--batch 4096"
B b2048b "--batch 2048"
(PLP_FRONT_LIB=build_exp/rprof.so timeout 120 python tools/grow_profile.py 2048 2>&1 | grep -v amdgpu.ids | tail -2) > $O/rprof.log; cat $O/rprof.log
