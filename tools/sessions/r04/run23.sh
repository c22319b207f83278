# GPU session r04/23: the N > 1 plumbing of bench.py on one GPU (two ranks over gloo, PLP_BENCH_SHARE_GPU), with the seed-order side pass
export TMPDIR=/tmp
O=gpurun_out/r04x; mkdir -p $O
export PLP_BENCH_SHARE_GPU=1
(timeout 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus 2 --steps 4 --warmup 2 --batch 512 --no-cpu-baseline 2> $O/bench2.err | tail -1) > $O/bench2.json
python -c "import json; j=json.load(open('$O/bench2.json')); print(j['value'], j['n_gpus'], j['ms_per_step'], j['other_seed_order'], j['scaling'])" || tail -5 $O/bench2.err
