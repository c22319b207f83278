# GPU session r06/58: which columns does rocprofv3's counters_collection view carry (LDS block size, workgroup size, VGPR counts per dispatch)?  For tools/step_profile.py's LDS-time bound.
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r06schema; mkdir -p $O
cd /tmp
timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES -d $O/sq -o sq -- python -c "import sys; sys.path.insert(0,'$R'); import __graft_entry__ as g; g.smoke()" > $O/log 2>&1
python - <<PY
import sqlite3,glob
db=glob.glob('$O/sq/*.db')[0]
c=sqlite3.connect(db).cursor()
for t in ('counters_collection','kernels'):
    print(t, [r[1] for r in c.execute(f'pragma table_info({t})')])
print(list(c.execute("select * from counters_collection limit 2")))
PY
rm -rf $O/sq
