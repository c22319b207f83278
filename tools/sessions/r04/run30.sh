# GPU session r04/30: seed sort workgroups of 1 / 2 / 3 waves (register-time held per frame) -- experiment builds, some with scratch
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
for v in r04z w3t4k w2t4k w2t4km2 w2t2k w1t4k; do
  export PLP_FRONT_LIB=build_exp/$v.so
  (timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/ssw_$v.err | tail -1) > $O/ssw_$v.json
  python -c "import json; j=json.load(open('$O/ssw_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'seed order alone', s.get('lsd_order'))" || tail -3 $O/ssw_$v.err
done
