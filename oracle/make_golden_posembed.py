#!/usr/bin/env python
"""Stage hand-off fixture: the REFERENCE's own ``load_from_ckpt`` (run_stage2.py:349-438) -- model-key selection, the
Kinetics-710 head slicing, 'backbone.' / 'encoder.' prefix stripping and the position-table interpolation (linear in time from the 8
pre-training frames, then bicubic in space, class tokens kept) -- executed from the file's syntax tree (the script itself cannot be
imported: wandb, decord, timm.create_model, missing src/knn.py) on small checkpoints written by this script
-> tests/golden/stage2_ckpt.npz.  The function's globals are the real ``torch`` / ``json`` / ``OrderedDict`` and a ``utils`` whose
``load_state_dict`` records what it is given.  TEST INFRASTRUCTURE; runs only where /root/reference exists."""
import ast
import json
import os
import sys
import tempfile
import types
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
from oracle import make_golden as G  # noqa: E402


def extract(name):
    path = os.path.join(G.REF, "run_stage2.py")
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == name]
    assert len(body) == 1
    return compile(ast.Module(body=body, type_ignores=[]), path, "exec")


def main():
    got = {}
    ns = {"torch": torch, "json": json, "OrderedDict": OrderedDict,
          "utils": types.SimpleNamespace(load_state_dict=lambda model, sd, prefix='': got.update(sd=sd, prefix=prefix))}
    exec(extract("load_from_ckpt"), ns)
    g = torch.Generator().manual_seed(17)
    D = 16
    out = {}
    # case a: 8 frames x 3 x 3 -> 16 frames x 2 x 2, no class token, 'encoder.' keys under 'model', a 710-way head cut to 400 classes
    # case b: 8 frames x 2 x 2 -> 8 frames x 3 x 3 with one class token (space only), 'backbone.' keys under 'module', head deleted
    # case c: 8 -> 4 frames, same grid (time only), a bare state_dict
    cases = {
        "a": dict(t_new=16, s_old=3, s_new=2, extra=0, key="model", prefix="encoder.", head=710, nb=400, delete_head=False),
        "b": dict(t_new=8, s_old=2, s_new=3, extra=1, key="module", prefix="backbone.", head=8, nb=8, delete_head=True),
        "c": dict(t_new=4, s_old=2, s_new=2, extra=0, key=None, prefix="", head=5, nb=5, delete_head=False),
    }
    with tempfile.TemporaryDirectory() as tmp:
        for tag, c in cases.items():
            n_old = 8 * c["s_old"] ** 2 + c["extra"]
            sd = OrderedDict()
            sd[c["prefix"] + "pos_embed"] = torch.randn(1, n_old, D, generator=g)
            sd[c["prefix"] + "blocks.0.norm1.weight"] = torch.randn(D, generator=g)
            sd["head.weight"] = torch.randn(c["head"], D, generator=g)
            sd["head.bias"] = torch.randn(c["head"], generator=g)
            path = os.path.join(tmp, f"{tag}.pth")
            torch.save({c["key"]: sd} if c["key"] else sd, path)
            n_new = c["t_new"] * c["s_new"] ** 2
            model = types.SimpleNamespace(patch_embed=types.SimpleNamespace(num_patches=n_new, tubelet_size=1),
                                          pos_embed=torch.zeros(1, n_new + c["extra"], D))
            args = types.SimpleNamespace(finetune=path, model_key="model|module", delete_head=c["delete_head"], nb_classes=c["nb"],
                                         num_frames=c["t_new"], model_prefix="")
            got.clear()
            ns["load_from_ckpt"](args, model)
            for k, v in sd.items():
                out[f"{tag}.in.{k}"] = v
            out[f"{tag}.in.cfg"] = np.array(json.dumps(c))
            out[f"{tag}.out.keys"] = np.array(list(got["sd"].keys()))
            for k, v in got["sd"].items():
                out[f"{tag}.out.{k}"] = v
            print(tag, {k: tuple(v.shape) for k, v in got["sd"].items()})
    np.savez_compressed(os.path.join(G.OUT, "stage2_ckpt.npz"), **G._np(out))


if __name__ == "__main__":
    main()
