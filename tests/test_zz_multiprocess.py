"""Multi-PROCESS GPU tests of the tiled path (SURVEY.md §8 e): one process per rank through torch.distributed.run.  The file name sorts after every parity and
property test on purpose: under `pytest -x` a failure of a process rehearsal can no longer hide the tests behind it.  Every launcher keeps each
rank's stdout / stderr (torch.distributed.run --log-dir / --tee) and a failing assertion prints every rank's stderr tail, so that a rank that dies on a signal
explains itself (round 4's SIGABRT of one rank left only the launcher's summary)."""
import glob
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _rank_logs(log_dir, lines=60):
    """Tail of every rank's stderr (and stdout when it holds an error) below a torch.distributed.run --log-dir."""
    out = []
    for f in sorted(glob.glob(os.path.join(str(log_dir), "**", "stderr.log"), recursive=True)):
        try:
            with open(f, errors="replace") as fh:
                tail = fh.read().splitlines()[-lines:]
        except OSError:
            continue
        out.append(f"===== rank {os.path.basename(os.path.dirname(f))}: {f}\n" + "\n".join(tail))
    return "\n".join(out) if out else "(no per-rank logs found)"


def _launch_worker(n_ranks, port, log_dir, one_gpu, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    if one_gpu:
        env["LUMEN_WORKER_ONE_GPU"] = "1"
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n_ranks}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--log-dir", str(log_dir), "--tee", "3", os.path.join(ROOT, "tests", "multigpu_worker.py")]
    res = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)      # children start fresh: nothing GPU-side is inherited
    assert res.returncode == 0 and "MULTIGPU OK" in res.stdout, res.stdout[-2000:] + "\n" + res.stderr[-2000:] + "\n" + _rank_logs(log_dir)


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_tiled_path_on_real_gpus_over_rccl(n_ranks, tmp_path):
    """The multi-GPU path as it runs in production: one process per GPU (torch.distributed.run, backend nccl = RCCL over xGMI),
    tiles + 60-px halo, seam exchange of reservoir history after every TraceFrame (odd depth), one RCCL gather per displayed frame;
    rank 0 compares the stitched blended frames bit for bit with a single-GPU render (tests/multigpu_worker.py).  Needs n_ranks GPUs
    on the box: skipped on the single-GPU boxes of the development pool, runs wherever the driver has a multi-GPU node."""
    import torch
    if torch.cuda.device_count() < n_ranks:
        pytest.skip(f"{torch.cuda.device_count()} GPU(s) on this box, {n_ranks} needed")
    _launch_worker(n_ranks, 29540 + n_ranks, tmp_path, one_gpu=False)


@pytest.mark.parametrize("n_ranks", [2, 4, 8])
def test_tiled_worker_processes_share_the_one_gpu(n_ranks, tmp_path):
    """The production worker as N real processes on the ONE GPU of the development boxes: every rank renders its window on GPU 0, the wave-count agreement, the seam
    exchange of the halo rings' reservoirs and the tile gather run between the processes (gloo, staged through the host — tests/host_staged_dist.py: RCCL does not
    accept two ranks on one device), and rank 0 compares every blended frame of the stitched image bit for bit with its own full-frame render.  What this does not
    cover is RCCL itself over xGMI (test_tiled_path_on_real_gpus_over_rccl, skipped without N GPUs)."""
    _launch_worker(n_ranks, 29560 + n_ranks, tmp_path, one_gpu=True)


def test_tiled_worker_single_rank_communicator(tmp_path):
    """The same worker with one rank: the nccl communicator, the preallocated gather buffers and the frame loop on the GPU that is there
    (the 2 / 4 / 8-rank forms above need a multi-GPU box)."""
    _launch_worker(1, 29539, tmp_path, one_gpu=False, timeout=600)


@pytest.mark.parametrize("n_ranks,workload", [(2, "sandbox"), (8, "c2"), (8, "c4")])      # c4 = BASELINE's 8-GPU configuration (4K, 8 spp, depth 8)
def test_bench_multi_rank_path_rehearsed_on_the_one_gpu(n_ranks, workload, tmp_path):
    """`python bench.py --gpus N` end to end where only one GPU exists (LUMEN_BENCH_ONE_GPU=1: every rank on GPU 0, collectives over gloo with host staging): the
    self-launch, the windows and tiles, the seam exchange after every TraceFrame (sandbox: odd depth), the gather, the barrier-bracketed timing with the maximum over
    ranks and the per-rank statistics all execute, and rank 0 prints ONE well-formed line that names itself a rehearsal.  The rate is not looked at.
    (Round 4: one rank of the 8-rank form died with SIGABRT on the driver's box.  Cause, LOG.md round 5 item 1: HSA_STATUS_ERROR_ILLEGAL_INSTRUCTION inside PyTorch's
    strided-copy kernel at its FIRST launch — its code object is loaded lazily, here while the renderer's streams were busy and seven other processes did the same on
    the same GPU; the frame path now launches only kernels of the renderer's own, already resident module.)"""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", LUMEN_BENCH_ONE_GPU="1", LUMEN_BENCH_LOG_DIR=str(tmp_path))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    res = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n_ranks), "--workload", workload, "--steps", "2", "--warmup", "1"],
                         env=env, capture_output=True, text=True, timeout=1200)
    assert res.returncode == 0, res.stdout[-1500:] + "\n" + res.stderr[-1500:] + "\n" + _rank_logs(tmp_path)
    lines = [l for l in res.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, res.stdout[-2000:]
    j = json.loads(lines[0])
    assert j["n_gpus"] == n_ranks and j["rccl_world"] == n_ranks and len(j["devices"]) == n_ranks and "rehearsal" in j
    assert len(j["per_rank"]) == n_ranks and all(p["render_ms_per_step"] > 0 for p in j["per_rank"])
    assert j["value"] > 0 and j["scaling"] == "strong" and j["config"]["tiles"].split(" ")[0] in ("2x1", "1x2", "4x2", "2x4")
