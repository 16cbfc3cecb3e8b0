"""HIP's hardware-queue pool, decided from the ENVIRONMENT before the first HIP call of a process (no torch.cuda call here: on some installs
counting devices already starts the runtime, after which the variable is ignored).

HIP maps its streams onto a pool of hardware queues (4 by default).  A training step keeps four streams busy at once (student, its weight
gradients, the teacher one batch ahead, the gradient reducer) and RCCL adds its own: with four queues the teacher's stream lands on the
student's queue as soon as a process group exists, the two phases run one after the other again and a step takes 24.2 instead of 20.4 ms
(measured with a one-rank RCCL group).  Eight queues are right wherever a rank has a GPU to itself; two processes SHARING
one GPU with eight queues each oversubscribe the hardware queues, which the driver then time-slices (a two-rank rehearsal: 64 s -> > 200 s).

The rule (`choose`):
  * GPU_MAX_HW_QUEUES already set                      -> left alone
  * UNITE_RANKS_SHARE_GPU=1 (set by a launcher that KNOWS it starts more ranks than there are GPUs: bench.py's self-start on a one-GPU
    box, the two-rank rehearsals of the tests)         -> the default pool stays
  * the visible-device mask names ONE device           -> this rank owns it (the one-GPU-per-rank launchers: HIP_VISIBLE_DEVICES=<local rank>)  -> 8
  * ranks on this node (LOCAL_WORLD_SIZE) <= GPUs seen -> every rank has its own                                                           -> 8
    (GPUs seen: the mask's entries, else the GPU nodes under /sys/class/kfd; unknown counts as enough)
  * more ranks than GPUs                                -> they share devices: the default pool stays
"""
import logging
import os
from typing import Mapping, Optional, Tuple

_MASKS = ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "CUDA_VISIBLE_DEVICES")
_KFD = "/sys/class/kfd/kfd/topology/nodes"
_log = logging.getLogger("unite_amd")


def mask_entries(env: Mapping[str, str]) -> Optional[int]:
    """number of devices the first visible-device mask in `env` names (None: no mask)"""
    for k in _MASKS:
        v = env.get(k)
        if v is not None:
            return len([e for e in v.split(",") if e.strip() != ""])
    return None


def kfd_gpu_nodes(root: str = _KFD) -> Optional[int]:
    """GPU nodes of the KFD topology (nodes with simd_count > 0); None where the tree is absent (no amdgpu driver, a container without /sys)"""
    try:
        nodes = os.listdir(root)
    except OSError:
        return None
    n = 0
    for d in nodes:
        try:
            with open(os.path.join(root, d, "properties")) as f:
                for line in f:
                    if line.startswith("simd_count"):
                        n += int(line.split()[1]) > 0
                        break
        except (OSError, ValueError, IndexError):
            continue
    return n


def choose(env: Mapping[str, str], gpus_on_node: Optional[int]) -> Tuple[Optional[str], str]:
    """(value to set or None, reason) -- pure: `env` is the process environment, `gpus_on_node` the KFD count (None: unknown)"""
    if "GPU_MAX_HW_QUEUES" in env:
        return None, f"GPU_MAX_HW_QUEUES={env['GPU_MAX_HW_QUEUES']} set by the caller"
    if env.get("UNITE_RANKS_SHARE_GPU", "0") == "1":
        return None, "the launcher says ranks share a GPU (UNITE_RANKS_SHARE_GPU=1): HIP's default queue pool stays"
    try:
        ranks = max(1, int(env.get("LOCAL_WORLD_SIZE", "1")))
    except ValueError:
        ranks = 1
    masked = mask_entries(env)
    if masked == 1:      # one rank per process mask (SLURM --gpus-per-task=1, HIP_VISIBLE_DEVICES=$LOCAL_RANK); an inherited one-GPU mask with several
        # ranks looks the same from inside a rank -- the launcher that creates that situation must say so (UNITE_RANKS_SHARE_GPU above)
        return "8", "the visible-device mask names one device: this rank owns it"
    seen = masked if masked is not None else gpus_on_node
    if seen is None or seen == 0:
        return "8", f"{ranks} rank(s) on the node, GPU count unknown: assuming one GPU per rank"
    if ranks <= seen:
        return "8", f"{ranks} rank(s) on the node, {seen} GPU(s) visible: one GPU per rank"
    return None, f"{ranks} ranks share {seen} GPU(s): HIP's default queue pool stays"


_applied = None


def apply() -> str:
    """apply the rule to os.environ once per process; returns the one-line reason (also logged at INFO on the `unite_amd` logger)"""
    global _applied
    if _applied is None:
        value, reason = choose(os.environ, kfd_gpu_nodes())
        if value is not None:
            os.environ["GPU_MAX_HW_QUEUES"] = value
            reason = f"GPU_MAX_HW_QUEUES={value}: " + reason
        _applied = reason
        _log.info("hardware queues: %s", reason)
        if os.environ.get("UNITE_VERBOSE"):
            import sys
            print(f"[unite_amd] hardware queues: {reason}", file=sys.stderr, flush=True)
    return _applied
