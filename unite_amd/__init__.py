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

# HIP's hardware-queue pool is sized before the first HIP call of the process, from the environment alone (hwqueues.py states the rule and why)
from .hwqueues import apply as _apply_hw_queues  # noqa: E402

_apply_hw_queues()

from .registry import create_model, register_model, list_models  # noqa: F401,E402
from . import modeling_adaptation  # noqa: F401,E402  (registers the student factories)
from . import modeling_finetune  # noqa: F401,E402  (registers the stage-2 classifier factories)
from . import clip  # noqa: F401,E402

__version__ = "0.1.0"
