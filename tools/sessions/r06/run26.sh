# GPU session r06/26: long form of the concurrent single-frame test -- >= 30 000 pairs (plp_orb_extract || plp_line_extract in two host threads + a matcher thread), every result against the oracle
export TMPDIR=/tmp
O=gpurun_out/r06soak; mkdir -p $O
timeout 1500 python tools/soak_concurrent_pairs.py --pairs 30000 --minutes 20 > $O/soak_concurrent_pairs.log 2>&1; tail -16 $O/soak_concurrent_pairs.log
