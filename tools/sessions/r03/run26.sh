# GPU session 26: full GPU suite, round profile (bench line, kernel trace, PMC passes), the other BASELINE configs
export TMPDIR=/tmp
O=gpurun_out/r03w; mkdir -p $O
(timeout 800 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -6) > $O/pytest.log; cat $O/pytest.log
bash tools/run_prof.sh r03w > $O/run_prof.log 2>&1
tail -3 $O/run_prof.log
(timeout 300 python tools/bench_configs.py --batch 1024 --steps 3 2>&1 | grep "^{" ) > $O/other_configs.jsonl; cat $O/other_configs.jsonl
(timeout 200 python tools/fuzz_gpu.py --only lines --seconds 150 --seed 62 2>&1 | grep "lines:") > $O/fuzz.log
(timeout 260 python tools/fuzz_gpu.py --only orb --seconds 200 --seed 25 2>&1 | grep "orb:") >> $O/fuzz.log
cat $O/fuzz.log
