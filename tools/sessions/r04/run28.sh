# GPU session r04/28: frames per step (region growing is one wave per frame: 2048 frames = 2 waves per SIMD, 3072 = 3, the most 147 VGPRs allow)
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
run() { # name, batch, env...
  name=$1; b=$2; shift; shift
  (env "$@" timeout 300 python bench.py --batch $b --steps 10 --warmup 3 --no-cpu-baseline --no-extras --verify 8 2> $O/batch_$name.err | tail -1) > $O/batch_$name.json
  python -c "import json; j=json.load(open('$O/batch_$name.json')); s=j['roofline']['stage_ms_per_batch']; print('$name', j['value'], j['ms_per_step'], 'stable:', j['other_seed_order']['value'], 'verified', j['verified_frames'], 'grow alone', s['lsd_grow'])" || tail -3 $O/batch_$name.err
}
run b2048 2048 X=1
run b3072 3072 X=1
run b3072_s1 3072 PLP_BENCH_LINE_SPLIT=1
run b3072_s3 3072 PLP_BENCH_LINE_SPLIT=3
run b2560 2560 X=1
run b4096 4096 X=1
run b1536 1536 X=1
run b1024 1024 X=1
