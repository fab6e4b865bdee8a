export PTI_DIST_BACKEND=gloo PTI_SHARE_GPU=1 HSA_ENABLE_IPC_MODE_LEGACY=0 OMP_NUM_THREADS=2
timeout -k 10 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node=2 --master-addr 127.0.0.1 --master-port 29611 bench.py --gpus 2 --steps 3 --warmup 3 --batch 2 --size 64 --no-cpu-baseline > gpurun_out/dp2.out 2> gpurun_out/dp2.err
echo "rc=$?"; grep -v "amdgpu.ids" gpurun_out/dp2.err | head -80
