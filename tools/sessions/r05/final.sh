# GPU session r05/final: the round's closing checks and measurements on the final tree
#   full GPU suite; randomised parity sweep (ORB / lines in both seed orders / matchers + the steps either side); 30 000 soak calls of the host entry; the round profile
#   (bench line with the CPU leg, rocprofv3 kernel trace, PMC traffic, SQ counters); bench.py at N = 2 on this one GPU (gloo), verified on every rank
export TMPDIR=/tmp
O=gpurun_out/r05z; mkdir -p $O
(timeout 900 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest.log; cat $O/pytest.log
(timeout 300 python tools/fuzz_gpu.py --seconds 150 --seed 91 --aux-seconds 40 2>&1 | tail -8) > $O/fuzz.log; cat $O/fuzz.log
(timeout 600 python tools/fuzz_gpu.py --soak-calls 30000 --seed 92 2>&1 | tail -2) > $O/soak.log; cat $O/soak.log
bash tools/run_prof.sh r05z > $O/run_prof.log 2>&1; tail -2 $O/run_prof.log | cut -c1-300
(PLP_BENCH_SHARE_GPU=1 MASTER_ADDR=127.0.0.1 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29617 bench.py --gpus 2 --steps 5 --warmup 2 --batch 1024 --no-cpu-baseline --no-extras --verify 8 2> $O/two_rank.err | grep '^{' | tail -1) > $O/r05_two_rank_bench.json; cut -c1-400 $O/r05_two_rank_bench.json
