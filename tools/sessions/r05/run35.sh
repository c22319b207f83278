# GPU session r05/35: bench.py's latency pass (256 synchronous host-pointer calls) on one box: the build before the latency series (build_exp/agentscope.so), the shipped
# one, the shipped one with the old claim policy
export TMPDIR=/tmp
O=gpurun_out/r05y; mkdir -p $O
run() { (PLP_FRONT_LIB=$1 PLP_LSD_MW_POLICY=$2 timeout 200 python bench.py --steps 2 --warmup 1 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1 policy $2', d['latency_ms_median_mean']['line_extract'], d['latency_ms_median_mean']['orb_par_line_extract'])") >> $O/lat_ab.log; }
run build_exp/agentscope.so 0
run "" 3
run "" 0
run build_exp/agentscope.so 0
run "" 3
cat $O/lat_ab.log
