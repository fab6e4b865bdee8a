import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run on the GPU box with -m gpu)")


# ---- the two-process data-parallel job of tests/test_gpu_data_parallel.py --------------------------------------------
# It must be started by a process that has NOT initialised the GPU (on this pool a GPU-initialised process must not
# start other programs), so it is launched here, right after collection and before the first test touches torch.cuda, and runs in the
# background beside the other tests (3 processes on the card in total).  torch.cuda.device_count() does not initialise
# the GPU on this image.
_DP_JOB = {}
_RCCL_JOB = {}


def _gpu_tests_selected(config):
    m = config.getoption("-m") or ""
    return "not gpu" not in m


def pytest_collection_finish(session):
    """After collection (test modules only import torch, nothing has touched torch.cuda yet), before the first test."""
    import subprocess
    import tempfile
    if not _gpu_tests_selected(session.config) or os.environ.get("PTI_SKIP_DP_JOB") == "1" or _DP_JOB:
        return
    if not any("test_gpu_data_parallel" in it.nodeid for it in session.items):
        return
    if "torch" in sys.modules and sys.modules["torch"].cuda.is_initialized():
        return      # too late to start other programs safely; the tests that need the job will skip
    try:
        import torch
        if torch.cuda.device_count() < 1:
            return
    except Exception:
        return
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    outdir = tempfile.mkdtemp(prefix="pti_dp_job_")
    log = open(os.path.join(outdir, "job.log"), "w")
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="2")
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port_b = s.getsockname()[1]
    s.close()
    # the launcher runs the two-rank worker, then `bench.py --gpus 2` in its one-GPU rehearsal form (tests/dp_jobs_launcher.py)
    cmd = [sys.executable, os.path.join(ROOT, "tests", "dp_jobs_launcher.py"), outdir, str(port), str(port_b)]
    _DP_JOB.update(proc=subprocess.Popen(cmd, stdout=log, stderr=subprocess.STDOUT, env=env, cwd=ROOT), outdir=outdir, log=log)
    # the single-rank job on the REAL backend (nccl = RCCL): tests/rccl_one_rank_worker.py
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port2 = s.getsockname()[1]
    s.close()
    outdir2 = tempfile.mkdtemp(prefix="pti_rccl_job_")
    log2 = open(os.path.join(outdir2, "job.log"), "w")
    env2 = dict(env, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port2))
    cmd2 = [sys.executable, os.path.join(ROOT, "tests", "rccl_one_rank_worker.py"), outdir2]
    _RCCL_JOB.update(proc=subprocess.Popen(cmd2, stdout=log2, stderr=subprocess.STDOUT, env=env2, cwd=ROOT), outdir=outdir2,
                     log=log2)


def pytest_sessionfinish(session, exitstatus):
    for job in (_DP_JOB, _RCCL_JOB):
        proc = job.get("proc")
        if proc is not None and proc.poll() is None:
            proc.terminate()          # the exact process started above


@pytest.fixture(scope="session")
def dp_job():
    """-> callable returning the output directory of the finished two-rank job (waits for it, at most 10 minutes)."""
    def wait():
        proc = _DP_JOB.get("proc")
        if proc is None:
            pytest.skip("the two-rank job was not started (no GPU at session start)")
        try:
            rc = proc.wait(timeout=600)
        except Exception:
            proc.terminate()
            pytest.fail("two-rank data-parallel job did not finish within 10 minutes")
        _DP_JOB["log"].close()
        if rc != 0:
            pytest.fail("two-rank data-parallel job failed:\n" + open(os.path.join(_DP_JOB["outdir"], "job.log")).read()[-4000:])
        return _DP_JOB["outdir"]
    return wait


@pytest.fixture(scope="session")
def rccl_job():
    """-> callable returning the output directory of the finished single-rank RCCL job."""
    def wait():
        proc = _RCCL_JOB.get("proc")
        if proc is None:
            pytest.skip("the RCCL job was not started (no GPU at session start)")
        try:
            rc = proc.wait(timeout=600)
        except Exception:
            proc.terminate()
            pytest.fail("single-rank RCCL job did not finish within 10 minutes")
        _RCCL_JOB["log"].close()
        if rc != 0:
            pytest.fail("single-rank RCCL job failed:\n" + open(os.path.join(_RCCL_JOB["outdir"], "job.log")).read()[-4000:])
        return _RCCL_JOB["outdir"]
    return wait


@pytest.fixture(scope="session")
def dev():
    import torch

    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")
