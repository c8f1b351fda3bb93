"""One process per GPU, started by the script itself.

The reference starts its ranks from a shell wrapper -- ``tools/run-nus.sh:11-13`` runs
``python -m torch.distributed.launch --nproc_per_node=$GPUS ... tools/train.py --launcher pytorch`` and
``mmdet3d/apis/train.py:82-86`` wraps the model in DistributedDataParallel.  Here ``bench.py --gpus N`` and
``tools/train_step.py --gpus N`` do the same thing without a wrapper: when N > 1 and no rendezvous environment is
present (``WORLD_SIZE`` unset), the script re-executes itself under ``python -m torch.distributed.run`` with N ranks on
127.0.0.1 and a free port; under a launcher (torchrun, the driver's own ``torch.distributed.run`` line) it does nothing.
Fewer than N visible devices is an error, not a silent single-GPU run.
"""
import os
import socket
import subprocess
import sys


def free_port():
    s = socket.socket(socket.AF_INET, socket.SOCK_STREAM)
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def world_from_env():
    """(rank, world, local_rank) of this process as torch.distributed.run exports them (1-process defaults)."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")),
            int(os.environ.get("LOCAL_RANK", "0")))


def launch_command(script, argv, gpus, port):
    """The command line a self-launch executes (kept separate so a CPU test can look at it)."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={gpus}",
            "--master-addr", "127.0.0.1", "--master-port", str(port), script] + list(argv)


def self_launch(gpus, backend="nccl", script=None, argv=None):
    """Re-execute the running script with `gpus` ranks if it was started plainly.  Returns None when this process is
    already a rank (or gpus == 1); otherwise runs the ranks, and exits with their return code -- it never returns to
    a caller that would go on to measure one GPU and label it N.

    backend "nccl" (RCCL) needs `gpus` visible devices; "gloo" is the CPU rehearsal used by the tests."""
    if gpus <= 1 or "WORLD_SIZE" in os.environ:
        _, world, _ = world_from_env()
        if "WORLD_SIZE" in os.environ and world != gpus:
            raise SystemExit(f"--gpus {gpus} but the launcher started WORLD_SIZE={world} ranks")
        return None
    if backend == "nccl":
        import torch
        have = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if have < gpus:
            raise SystemExit(f"--gpus {gpus} asked for, but only {have} GPU(s) are visible: refusing to measure "
                             f"fewer devices than the line would claim")
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL across processes fails without it here
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // gpus)))
    cmd = launch_command(script or os.path.abspath(sys.argv[0]), sys.argv[1:] if argv is None else argv, gpus, free_port())
    rc = subprocess.call(cmd, env=env)
    raise SystemExit(rc)


def rccl_version():
    """'2.x.y' of the RCCL torch was built against (backend "nccl" IS RCCL on ROCm); '' without a GPU build."""
    try:
        import torch
        v = torch.cuda.nccl.version()
        return ".".join(str(x) for x in v) if isinstance(v, tuple) else str(v)
    except Exception:
        return ""
