# GPU session 19: FAST (batched staging, wave-aggregated queue pushes, packed pre-test), orientation + rBRIEF (46 VGPRs, 25.9 KB), quadtree (62 VGPRs,
# 25.5 KB), line matcher lanes kernel (targets broadcast from registers), matcher kernels per family of modes; region-grower wave priority; PCIe prefetch
export TMPDIR=/tmp
O=gpurun_out/r03y; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
(timeout 600 python -m pytest tests -m gpu -q -p no:cacheprovider -x 2>&1 | tail -8) > $O/pytest.log; cat $O/pytest.log
(timeout 100 python tools/fuzz_gpu.py --only orb --seconds 60 --seed 71 2>&1 | grep "orb:" | tail -1) > $O/fuzz.log
(timeout 80 python tools/fuzz_gpu.py --only match --seconds 45 --seed 72 2>&1 | grep -i "match" | tail -2) >> $O/fuzz.log
cat $O/fuzz.log
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 |', j['value'], j['ms_per_step'], '| grow', s['lsd_grow'], 'fast', s['fast_cells'], 'quadtree', s['quadtree'], 'rbrief', s['orient_rbrief'], 'match_4x', s['match_4x'])"; }
{
B r03z
B base
B new
B prio1
B prio3
B new
B r03z
} > $O/ab.log 2>&1
cat $O/ab.log
cp build_exp/.orig.so $L
(timeout 300 python bench.py --verify 64 2> $O/bench.err) > $O/bench.json; python -c "
import json; j=json.load(open('$O/bench.json')); print(j['value'], j['ms_per_step'], j.get('verified_frames'), j.get('pcie_inclusive_value'), j.get('pcie_inclusive_ms_per_step'), j['roofline']['stage_ms_per_batch'])"
