# GPU session r06/10: seed sort as TWO kernels for large batches (16 waves per frame: array + global-memory partitions, list of window segments; 4 waves per frame: the windows); tests, A/B against the single kernel (PLP_SS_SPLIT=0) on this box
export TMPDIR=/tmp
O=gpurun_out/r06j; mkdir -p $O
(timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -5) > $O/ss.log; cat $O/ss.log
(PLP_SS_SPLIT=0 timeout 900 python -m pytest tests/test_gpu_seed_sort.py -q -x -p no:cacheprovider 2>&1 | tail -2) > $O/ss_fused.log; cat $O/ss_fused.log
for c in 1 2048; do PLP_SEED_SORT_DBG_COPIES=$c timeout 300 python tools/experiments/seed_sort_prof.py > $O/prof_$c.log 2>&1; echo "copies $c"; tail -1 $O/prof_$c.log; done
(timeout 900 python -m pytest tests/test_gpu_line.py tests/test_gpu_seed_sort_soak.py -q -x -p no:cacheprovider 2>&1 | tail -3) > $O/line.log; cat $O/line.log
for rep in 1 2; do for sp in 1 0; do
PLP_SS_SPLIT=$sp timeout 600 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline --verify 16 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('split $sp', j['value'], j['ms_per_step'], 'order', s['lsd_order'], 'verified', j['verified_frames'])"
done; done | tee $O/ab.log
