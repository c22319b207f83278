# GPU session r04/36: k_lsd_grow at 143 VGPRs (fit from the ring for regions up to 128 points instead of 256): a third 72-VGPR wave fits beside two growers
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04x; mkdir -p $O
for v in r04z rp2 r04z rp2; do
  export PLP_FRONT_LIB=build_exp/$v.so
  (timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/rp_$v.err | tail -1) > $O/rp_$v.json
  python -c "import json; j=json.load(open('$O/rp_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -3 $O/rp_$v.err
done
export PLP_FRONT_LIB=build_exp/rp2.so
(timeout 300 python -m pytest tests/test_gpu_line.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/rp2_pytest.log; cat $O/rp2_pytest.log
