# GPU session r06/34: where the single-frame line call's time goes (stage events, 64 replay frames, both seed orders)
export TMPDIR=/tmp
O=gpurun_out/r06lat; mkdir -p $O
timeout 300 python tools/experiments/latency_stages.py 2>&1 | grep "order" | tee $O/stages.log
