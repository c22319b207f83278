# GPU session 23: LBD direction from k_keylines (k_lbd 76 -> 59 VGPRs), blur5 + Sobel with the blurred tile in place of the staged rows (18.5 -> 14.4 KB);
# timeline of a steady-state step (which stream is the long one now)
export TMPDIR=/tmp
O=gpurun_out/r03x5; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
(timeout 300 python -m pytest tests/test_gpu_line.py tests/test_gpu_bench_step.py tests/test_gpu_golden_ref.py tests/test_gpu_facade.py -q -p no:cacheprovider -x 2>&1 | tail -2) > $O/pytest.log; cat $O/pytest.log
(timeout 60 python tools/fuzz_gpu.py --only lines --seconds 30 --seed 81 2>&1 | grep "lines:" | tail -1) > $O/fuzz.log; cat $O/fuzz.log
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'sobel', s['lbd_blur5_sobel'], 'lbd', s['lbd'], 'match_4x', s['match_4x'])"; }
{
for pass in 1 2 3; do
B cur2
B cur3
done
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/cur3.so $L
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d $GRAFT_REPO_ROOT/$O/kt -o kt -- python $GRAFT_REPO_ROOT/bench.py --steps 8 --warmup 2 --no-cpu-baseline --no-extras --verify 0 > $GRAFT_REPO_ROOT/$O/kt.log 2>&1
cd $GRAFT_REPO_ROOT
python tools/rocpd_timeline.py $O/kt/kt_results.db 6 > $O/timeline_step6.md
python tools/rocpd_summary.py $O/kt/kt_results.db "session 23 (bench.py --steps 8 --warmup 2)" > $O/kernel_stats.md
rm -rf $O/kt
head -70 $O/timeline_step6.md
cp build_exp/.orig.so $L
