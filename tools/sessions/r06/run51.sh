# GPU session r06/51: (a) the other BASELINE configs on the final tree; (b) the second-dispatch failure, two more questions on the hand-written vector-address form of the masks' HBM accesses:
# which SIDE fails (loads only / stores only in that form), and does the cache policy of those instructions matter (sc0, sc0 sc1, nt).  Two line sub-blocks on two streams, three processes per build.
export TMPDIR=/tmp
O=gpurun_out/r06nb2; mkdir -p $O
(timeout 900 python tools/bench_configs.py --batch 1024 --steps 3 --verify 8 2>&1 | grep -v amdgpu.ids | tail -12) > $O/other_configs.log; cat $O/other_configs.log
(timeout 300 python bench.py --keypoints 2000 --steps 5 --warmup 2 2>&1 | tail -1) > $O/k2000.json; cut -c1-300 $O/k2000.json
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.cand.so
for v in asm0 ldonly stonly sc1 sc3 nt; do
  cp build_exp/$v.so $L
  for i in 1 2 3; do FLN_CASES="lines:2" timeout 300 python tools/experiments/flat_neighbours.py > $O/${v}_$i.log 2>&1; echo "$v run $i: $(grep '^parts' $O/${v}_$i.log || echo 'process died (memory fault)')"; done
done
cp build_exp/.cand.so $L
