"""Self-launch of the one-process-per-GPU job: `python bench.py --gpus N` without a launcher starts its own N ranks.

The parent must not have touched the GPU when it does this (a process that has initialised HIP may neither fork
workers that use the card nor exec another program on this pool), so everything here is plain Python + subprocess:
the parent counts devices from sysfs (count_gpus(): KFD topology nodes with SIMDs; no HIP / torch call at all), starts
`python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ...` as a CHILD and exits
with its return code.  The ranks read RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* like under any other launcher.
The reference has no inference-time launcher (its only distributed code is accelerate in train/train_large.py:58,82-86).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys
from typing import List, Optional, Sequence


KFD_NODES = "/sys/class/kfd/kfd/topology/nodes"


def count_gpus(root: str = KFD_NODES, env: Optional[dict] = None) -> Optional[int]:
    """GPUs this process could use, WITHOUT touching HIP: KFD topology nodes whose `properties` file has simd_count > 0
    (CPU nodes have 0), capped by the length of HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when one is set.  None when the
    topology is not readable although /dev/kfd exists (then the caller skips its early check and the ranks fail by themselves
    on a missing device); 0 when there is no /dev/kfd either -- HIP cannot open a GPU without it.
    (torch.cuda.device_count() happens not to initialise the GPU on this image, but on a torch built without the amdsmi path
    it calls hipGetDeviceCount, and a parent that has initialised HIP must not start GPU children on this pool.)"""
    e = os.environ if env is None else env
    try:
        nodes = sorted(os.listdir(root))
    except OSError:
        return None if (root != KFD_NODES or os.path.exists("/dev/kfd")) else 0
    n = 0
    for node in nodes:
        try:
            with open(os.path.join(root, node, "properties")) as f:
                props = dict(line.split()[:2] for line in f if len(line.split()) >= 2)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) > 0:
            n += 1
    for var in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES"):
        v = e.get(var)
        if v is not None:
            n = min(n, len([x for x in v.split(",") if x.strip() != ""]))
    return n


def launched_by_a_launcher() -> bool:
    """True inside a rank started by torch.distributed.run (or any launcher that sets the torchrun contract)."""
    return "RANK" in os.environ and "WORLD_SIZE" in os.environ


def free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return int(s.getsockname()[1])


def rank_command(script: str, argv: Sequence[str], n: int, port: Optional[int] = None) -> List[str]:
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={int(n)}", "--master-addr", "127.0.0.1",
            "--master-port", str(port or free_port()), script, *argv]


def spawn_ranks(script: str, argv: Sequence[str], n: int, backend: str = "nccl", n_devices: Optional[int] = None,
                env: Optional[dict] = None) -> int:
    """Start n ranks of `script argv` and wait; returns the launcher's exit code.
    backend "nccl" (= RCCL) needs one GPU per rank: fewer visible devices is an error here, never a silent
    smaller job.  Any other backend (gloo) is a rehearsal in which ranks may share devices."""
    if n < 2:
        raise ValueError("spawn_ranks is for N >= 2")
    if backend == "nccl" and n_devices is not None and n_devices < n:
        print(f"error: --gpus {n} needs {n} visible GPUs for the RCCL job, found {n_devices} "
              f"(MGEA_DIST_BACKEND=gloo rehearses the {n}-rank flow on fewer devices)", file=sys.stderr)
        return 2
    e = dict(os.environ if env is None else env)
    e.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL between processes needs it on this driver
    e.setdefault("OMP_NUM_THREADS", "4")
    e["MGEA_DIST_BACKEND"] = backend
    return subprocess.call(rank_command(script, argv, n), env=e)
