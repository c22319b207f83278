# Same-box A/B of several builds of libplp_front.so (boxes differ by ~1 %, kernels of one round by less: compare on ONE box).
# Build each variant here, copy it to build_exp/<name>.so (git-ignored, travels with gpurun), then on the GPU box:
#   bash tools/ab_libs.sh base cand1 cand2        # bench line per variant, round robin twice; parity of the LAST variant named
# Prints: name, frames/s, ms per step, isolated k_lsd_grow ms, isolated matcher ms.
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 90 python bench.py --no-cpu-baseline --no-extras --verify 0 $BENCH_ARGS 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
for pass in 1 2; do for v in "$@"; do B $v; done; done
last="${@: -1}"; cp build_exp/$last.so $L
timeout 120 python -m pytest tests -m gpu -x -q 2>&1 | tail -1
timeout 40 python tools/fuzz_gpu.py --seconds 15 --seed 41 2>&1 | grep -i "mismatch" | tail -3
cp build_exp/.orig.so $L
