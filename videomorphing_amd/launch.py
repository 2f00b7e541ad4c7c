"""Self-launch of an N-rank job from a plain `python script.py --gpus N` invocation.

One process per GPU is the execution model (torch.distributed over RCCL); a driver that
starts the script WITHOUT torch.distributed.run still has to get N ranks.  The parent
process here never touches the GPU (no torch import, no HIP call): it starts
`python -m torch.distributed.run --nproc-per-node N script args...` as a CHILD process,
relays its output and returns its exit code -- it does not exec (a process image that
initialised the GPU must not be replaced, and a parent that did not is free to wait).
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


def needs_launch(n_ranks, environ=None):
    """True when N > 1 ranks were asked for and this process is not already one of them."""
    environ = os.environ if environ is None else environ
    return n_ranks > 1 and "WORLD_SIZE" not in environ


def self_launch(script, argv, n_ranks, timeout=None):
    """Run `script argv` as n_ranks ranks on this node; stdout/stderr pass through.
    Returns the launcher's exit code."""
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC only on this pool (RCCL needs it)
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 1) // n_ranks)))
    env["MASTER_ADDR"] = "127.0.0.1"
    port = str(free_port())
    env["MASTER_PORT"] = port
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n_ranks),
           "--master-addr", "127.0.0.1", "--master-port", port, script] + list(argv)
    proc = subprocess.Popen(cmd, env=env)
    try:
        return proc.wait(timeout=timeout)
    except subprocess.TimeoutExpired:
        proc.kill()          # exactly the child we started
        proc.wait()
        return 124
