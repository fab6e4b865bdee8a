"""Started by tests/conftest.py before the pytest process touches the GPU; never initialises the GPU itself.  Runs, ONE AFTER
THE OTHER (so that at most two of these ranks share the card with the pytest process and the single-rank RCCL job: the
box allows six GPU processes):
  1. the two-rank data-parallel worker (tests/dp_gpu_worker.py, gloo, both ranks on cuda:0);
  2. ``bench.py --gpus 2`` exactly as the driver launches it (``python -m torch.distributed.run --nnodes=1
     --nproc-per-node 2 --master-addr 127.0.0.1 --master-port P bench.py --gpus 2 --steps K --warmup W``) in its
     single-GPU rehearsal form (PTI_DIST_BACKEND=gloo PTI_SHARE_GPU=1), tiny workload; its stdout goes to bench_dp2.out.
usage: dp_jobs_launcher.py <outdir> <port1> <port2>"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    outdir, p1, p2 = sys.argv[1], sys.argv[2], sys.argv[3]
    run = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1"]
    rc = subprocess.run(run + ["--master-port", p1, os.path.join(ROOT, "tests", "dp_gpu_worker.py"), outdir], cwd=ROOT).returncode
    if rc != 0:
        sys.exit(rc)
    env = dict(os.environ, PTI_DIST_BACKEND="gloo", PTI_SHARE_GPU="1")
    with open(os.path.join(outdir, "bench_dp2.out"), "w") as out, open(os.path.join(outdir, "bench_dp2.err"), "w") as err:
        rc = subprocess.run(run + ["--master-port", p2, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "3",
                                   "--batch", "2", "--size", "64", "--no-cpu-baseline", "--detail-out", os.path.join(outdir, "bench_dp2_detail.json")],
                            cwd=ROOT, env=env, stdout=out, stderr=err).returncode
    with open(os.path.join(outdir, "bench_dp2.rc"), "w") as f:
        f.write(str(rc))


if __name__ == "__main__":
    main()
