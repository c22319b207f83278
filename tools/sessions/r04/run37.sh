# GPU session r04/37: how much does the LDS the growers HOLD cost the step?  (padding on top of their 10.6 KB per wave, never touched)
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04x; mkdir -p $O
run() { name=$1; shift
  (env "$@" timeout 200 python bench.py --steps 12 --warmup 4 --no-cpu-baseline --no-extras --verify 8 2> $O/pad_$name.err | tail -1) > $O/pad_$name.json
  python -c "import json; j=json.load(open('$O/pad_$name.json')); s=j['roofline']['stage_ms_per_batch']; print('$name', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['ms_per_step'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -3 $O/pad_$name.err
}
run pad0 X=1
run pad2600 PLP_LSD_LDS_PAD=2600
run pad5200 PLP_LSD_LDS_PAD=5200
run pad0b X=1
run ring128 PLP_LSD_RING=128
run ring64 PLP_LSD_RING=64
