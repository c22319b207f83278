set -x
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out/r01f
cd $R && timeout 900 python bench.py > gpurun_out/r01f/bench.json 2> gpurun_out/r01f/bench.err
cd /tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r01f/kt -o kt -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/kt.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $R/gpurun_out/r01f/pmc_fetch -o f -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/pmc_fetch.log 2>&1
timeout 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $R/gpurun_out/r01f/pmc_write -o w -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline > $R/gpurun_out/r01f/pmc_write.log 2>&1
ls -la $R/gpurun_out/r01f/*
cat $R/gpurun_out/r01f/bench.json
