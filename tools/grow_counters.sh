# Diagnostic: latency counters of k_lsd_grow with 256 and 2048 frames in flight.  Usage (GPU box): bash tools/grow_counters.sh
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/growctr
mkdir -p $O
cd /tmp
for B in 256 2048; do
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VMEM SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_IFETCH SQ_IFETCH_LEVEL -d $O/a$B -o a -- python $R/tools/grow_scaling.py --same-frame --frames $B > $O/a$B.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM -d $O/b$B -o b -- python $R/tools/grow_scaling.py --same-frame --frames $B > $O/b$B.log 2>&1
  timeout 300 rocprofv3 --kernel-trace --pmc SQC_ICACHE_REQ SQC_ICACHE_MISSES SQC_DCACHE_REQ SQC_DCACHE_MISSES SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY -d $O/c$B -o c -- python $R/tools/grow_scaling.py --same-frame --frames $B > $O/c$B.log 2>&1
done
cd $R
python - <<'PY' > $O/summary.txt 2>&1
import sqlite3, glob, os
from collections import defaultdict
O = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "gpurun_out", "growctr")
for B in (256, 2048):
    acc = defaultdict(float); n = 0
    for t in "abc":
        for db in glob.glob(f"{O}/{t}{B}/*.db"):
            cur = sqlite3.connect(db).cursor()
            for name, cn, val, dur in cur.execute("select kernel_name,counter_name,value,duration from counters_collection"):
                if "k_lsd_grow" not in name: continue
                acc[(t, cn)] += val
                if cn in ("SQ_WAVES", "SQC_ICACHE_REQ") : acc[(t, "n")] += 1; acc[(t, "dur")] += dur
    print("B =", B)
    for k in sorted(acc): print("  ", k, acc[k])
PY
cat $O/summary.txt
rm -rf $O/a256 $O/b256 $O/c256 $O/a2048 $O/b2048 $O/c2048
