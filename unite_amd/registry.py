"""Model registry: the two timm entry points the reference drivers use (``@register_model`` at
modeling_adaptation.py:337,359 / modeling_finetune.py:386-419 and ``create_model`` at run_stage1.py:275)."""
from typing import Callable, Dict

_MODELS: Dict[str, Callable] = {}


def register_model(fn: Callable) -> Callable:
    _MODELS[fn.__name__] = fn
    return fn


def create_model(model_name: str, pretrained: bool = False, **kwargs):
    if model_name not in _MODELS:
        raise RuntimeError(f"Unknown model ({model_name}); registered: {sorted(_MODELS)}")
    # timm 0.4.12's create_model (the reference's pinned version) drops these three keyword arguments when they are None before it calls
    # the entry point: that is how run_stage2.py:338 can pass drop_block_rate=None to a VisionTransformer that has no such parameter
    for k in ("drop_block_rate", "drop_connect_rate", "drop_path_rate"):
        if k in kwargs and kwargs[k] is None:
            kwargs.pop(k)
    return _MODELS[model_name](pretrained=pretrained, **kwargs)


def list_models():
    return sorted(_MODELS)
