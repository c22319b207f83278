# GPU session r04/29: wave priority of the long-lived register holders (region growing, seed sort) RAISED above the co-runners (round 3 had raised the co-runners: worse)
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
for v in r04z gp3 gp3sp3 gp3sp1 sp3 gp1 r04z; do
  export PLP_FRONT_LIB=build_exp/$v.so
  (timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/prio_$v.err | tail -1) > $O/prio_$v.json
  python -c "import json; j=json.load(open('$O/prio_$v.json')); s=j['roofline']['stage_ms_per_batch']; print('$v', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -3 $O/prio_$v.err
done
