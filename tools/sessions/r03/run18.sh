# GPU session 18: the 147-register grower, waves per workgroup (how the 2048 waves spread over the SIMDs)
O=gpurun_out/r03x; mkdir -p $O
L=structure-plp-slam_amd/libplp_front.so
cp $L build_exp/.orig.so
B() { cp build_exp/$1.so $L; timeout 120 python bench.py --no-cpu-baseline --no-extras --verify 0 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read()); s=j['roofline']['stage_ms_per_batch']; print('$1 $2 |', j['value'], j['ms_per_step'], s['lsd_grow'], s['match_4x'])"; }
{
B rect170 wpb1
PLP_LSD_WPB=4 B rect170 wpb4
B rect147 wpb1
PLP_LSD_WPB=2 B rect147 wpb2
PLP_LSD_WPB=4 B rect147 wpb4
PLP_LSD_WPB=4 B rect147 wpb4
B rect170 wpb1
} > $O/ab.log 2>&1
cp build_exp/.orig.so $L
cat $O/ab.log
