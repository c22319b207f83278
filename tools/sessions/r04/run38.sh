# GPU session r04/38: k_lsd_grow at 128 VGPRs again (60 B of scratch, all of it kernel-lifetime values reloaded in the rectangle fit)
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04x; mkdir -p $O
for v in r04z g128 g128rp2 r04z g128; do
  export PLP_FRONT_LIB=build_exp/$v.so
  (timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/g128_$v.err | tail -1) > $O/g128_$v.json
  python -c "import json; j=json.load(open('$O/g128_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -3 $O/g128_$v.err
done
