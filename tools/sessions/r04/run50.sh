# GPU session r04/50: the final binary (scoped-access / perturbation knobs compiled out): full GPU suite and the bench line once more
export TMPDIR=/tmp
ulimit -c 0
O=gpurun_out/r04v3; mkdir -p $O
(timeout 700 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extras 2> $O/bench.err | tail -1) > $O/bench.json; python -c "import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j['other_seed_order'], j['verified_frames'], j['roofline']['stage_ms_per_batch']['lsd_order'])" || tail -3 $O/bench.err
