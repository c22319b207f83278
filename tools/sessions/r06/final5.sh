# GPU session r06/final5: the driver's sequence on the closing tree -- whole GPU suite, smoke(), python bench.py with no flags
export TMPDIR=/tmp
O=gpurun_out/r06z; mkdir -p $O
(timeout 1500 python -m pytest tests -m gpu -q -p no:cacheprovider 2>&1 | tail -3) > $O/pytest_final5.log; cat $O/pytest_final5.log
( time python -c "import __graft_entry__ as g; g.smoke()" ) 2>&1 | tail -5
( time python bench.py > $O/bench_default_final5.json 2> $O/bench_default_final5.err ) 2>&1 | tail -4; cut -c1-220 $O/bench_default_final5.json
python -c "
import json;j=json.load(open('$O/bench_default_final5.json'));r=j['roofline'];print(j['value'],j['ms_per_step'],j['verified_frames'],j['verified_halo_rows'],r['frac'],r['occupancy_bound']['packing'],r['occupancy_bound']['lds']['packing'],j['cpu_baseline']['value'],j['config']['distinct_frames'])"
