# GPU session r04/final: the round's closing checks and measurements on the final tree
#   full GPU suite; randomised parity sweep (ORB / lines in both seed orders / matchers + the steps either side); 30 000 soak calls of the host entry
#   (VERDICT r03 item 8); the round profile (bench line with the CPU leg, rocprofv3 kernel trace, PMC traffic, SQ counters); the other BASELINE configs, verified
export TMPDIR=/tmp
O=gpurun_out/r04z; mkdir -p $O
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 300 python tools/fuzz_gpu.py --seconds 150 --seed 61 --aux-seconds 40 2>&1 | tail -8) > $O/fuzz.log; cat $O/fuzz.log
(timeout 600 python tools/fuzz_gpu.py --soak-calls 30000 --seed 62 2>&1 | tail -2) > $O/soak.log; cat $O/soak.log
bash tools/run_prof.sh r04z > $O/run_prof.log 2>&1; tail -2 $O/run_prof.log | cut -c1-300
(timeout 600 python tools/bench_configs.py --batch 1024 --steps 3 --verify 8 2> $O/configs.err) > $O/configs.jsonl; cut -c1-200 $O/configs.jsonl
