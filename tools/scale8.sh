# The exact launch lines of the scaling curve (an 8-GPU MI355X node; one process per GPU over RCCL).  Not run in this project's rounds (one GPU per lease):
# the driver launches the same lines.  Every line carries verified_frames / verified_halo_rows (bench.py --verify at N > 1: every rank checks its block and its halo).
# The step makes ONE packed exchange per rank (replay.halo_exchanger): a neighbour shift by default, the all-gather north_star names with PLP_BENCH_HALO=allgather --
# both are timed here, so that the curve shows what the collective costs against the point-to-point form.
export HSA_ENABLE_IPC_MODE_LEGACY=0
python bench.py --gpus 1 --steps 20 --warmup 5
for HALO in ring allgather; do
  for N in 2 4 8; do
    PLP_BENCH_HALO=$HALO python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port $((29500 + N)) bench.py --gpus $N --steps 20 --warmup 5
  done
done
