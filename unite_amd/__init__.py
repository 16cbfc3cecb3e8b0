"""unite_amd -- MI355X-native (gfx950) implementation of UNITE's data-parallel training hot path.

Drop-in surface (same names / arguments as reddyav1/unite):
    unite_amd.create_model("adaptation_umt_base_patch16_224", ...)      run_stage1.py:275
    unite_amd.clip.clip_b16(...)                                         run_stage1.py:782
    unite_amd.optim_factory.create_optimizer / LayerDecayValueAssigner   run_stage1.py:827
    unite_amd.utils.NativeScalerWithGradNormCount / cosine_scheduler     run_stage1.py:830-840
    unite_amd.engine_stage1.train_one_epoch                              run_stage1.py:294-505
All device arithmetic is in unite_amd/lib/libunite_hip.so (C ABI: include/unite_hip.h).
"""
import os as _os

# HIP maps its streams onto a pool of hardware queues (4 by default).  A training step here keeps four streams busy at once (student, its
# weight gradients, the teacher one batch ahead, the gradient reducer) and RCCL adds its own: with four queues the teacher's stream lands
# on the student's queue as soon as a process group exists, the two phases run one after the other again and a step takes 24.2 instead of
# 20.4 ms (measured with a one-rank RCCL group).  Has to be in the environment before the first HIP call of the process.
# Only where every rank of the node has a GPU of its own: two processes sharing one GPU (the gloo rehearsals of the tests) with eight queues
# each oversubscribe the hardware queues, which the driver then time-slices -- a two-rank rehearsal went from 64 s to > 200 s.


def _own_gpu_per_rank() -> bool:
    import torch
    return int(_os.environ.get("LOCAL_WORLD_SIZE", "1")) <= max(torch.cuda.device_count(), 1)      # (counting devices does not start HIP)


if _own_gpu_per_rank():
    _os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from .registry import create_model, register_model, list_models  # noqa: F401,E402
from . import modeling_adaptation  # noqa: F401,E402  (registers the student factories)
from . import modeling_finetune  # noqa: F401,E402  (registers the stage-2 classifier factories)
from . import clip  # noqa: F401,E402

__version__ = "0.1.0"
