# round 4: the driver's own launcher at two ranks rehearsed on one GPU (gloo), a soak of the rebuilt noise generator, an fp16 soak
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 20 --warmup 5 --rehearse-on-one-gpu --no-cpu-baseline > $O/r4_torchrun_rehearsal.json 2> $O/r4_torchrun_rehearsal.err; echo "torchrun rehearsal rc=$?"; tail -c 600 $O/r4_torchrun_rehearsal.json
timeout -k 10 500 python tools/soak.py 1000000 > $O/r4_soak_noise.log 2>&1; echo "soak rc=$?"; tail -3 $O/r4_soak_noise.log
