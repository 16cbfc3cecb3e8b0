#!/usr/bin/env python
"""Generate tests/golden/stage3_step.npz by RUNNING THE REFERENCE'S OWN stage-3 ``train_one_epoch`` on the CPU.

TEST INFRASTRUCTURE.  Runs only in the build container (needs /root/reference); only inputs / outputs are committed.

``run_stage3.py`` cannot be imported as a module (wandb, decord / cv2 through src.datasets, timm.create_model, the missing
src/knn.py: SURVEY.md 8c).  Its training step can still be executed: the two function definitions ``pool_outputs``
(run_stage3.py:333-338) and ``train_one_epoch`` (:340-710) are taken out of the file's syntax tree with ``ast`` and compiled, unchanged,
into a namespace that holds exactly the globals those two functions use:
  * ``torch`` -- the real torch behind a thin proxy whose ``cuda.synchronize`` is a no-op and whose ``cuda.amp.autocast`` is a null
    context (the reference on a CPU-only box: autocast does not touch CPU tensors, SURVEY A-17);
  * ``nn``, ``F``, ``einops``, ``math``, ``sys``, ``time``, ``np``, ``Iterable`` -- the real modules;
  * ``utils`` -- the reference's own src/utils.py (imported as in oracle/make_golden.py: stand-ins for timm / tensorboardX /
    torch._six / the OpenAI clip package), i.e. its MetricLogger, SmoothedValue, get_greedy_masks and clip_infer run as written;
    ``utils.setup_clip`` (OpenAI ``clip.load`` + tokenizer: network / weights unavailable offline) is replaced by a function that returns
    an object whose ``encode_image`` yields INJECTED per-frame image features, and injected class text features -- the similarity
    arithmetic itself (normalise, softmax(100 cos), mean over frames) is then the reference's own ``clip_infer``;
  * ``wandb`` -- an object whose ``log`` records the dictionary the step reports (select ratio, error rates).
The model is the reference's own AdaptationVisionTransformer (tiny configuration, seeded weights) behind a one-attribute ``.module``
holder (stands for DistributedDataParallel), the mask teacher the reference's clip.VisionTransformer, the classifier an nn.Linear;
``loss_scaler`` is a recorder that runs ``loss.backward()`` and the reference's ``utils.get_grad_norm_`` (utils.py:631-643).

Usage:  python oracle/make_golden_stage3.py   (writes tests/golden/stage3_step.npz)
"""
from __future__ import annotations

import ast
import contextlib
import math
import os
import sys
import time
import types
from functools import partial
from typing import Iterable

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))

from oracle import make_golden as MG  # noqa: E402
from oracle.filler import fill_state_dict, make_videos  # noqa: E402

NOISE = float(os.environ.get("NOISE", "0.26"))          # spread of the injected CLIP features around their common component
# Seed and classifier scale are chosen (a search over seeds 1-40 at scales 0.3 / 0.4: the MARGINS line printed at the end, DRY=1 for the line alone) so that
# every one of the seven selection rules picks a PROPER subset of the four target clips, the confidences fall on both sides of the 0.5 / 0.6
# thresholds, and no decision sits close enough to its threshold for a bf16-operand logit error (measured <= 0.035) to flip it: the gaps between
# the two largest logits are >= 0.20 (full clips) / 0.22 (masked committee), the confidences >= 0.025 from 0.5 and >= 0.05 from 0.6
SEED = int(os.environ.get('SEED', '15'))
CLS_SCALE = float(os.environ.get('CLS_SCALE', '0.3'))
GRAD_KEYS = ['encoder.patch_embed.proj.weight', 'encoder.blocks.0.attn.qkv.weight', 'encoder.blocks.0.attn.q_bias', 'encoder.blocks.1.mlp.fc1.weight',
             'encoder.blocks.2.mlp.fc2.bias', 'encoder.blocks.2.norm2.weight', 'encoder.norm.weight', 'encoder.norm.bias']
CLIP_THRESHOLD = 0.6       # args.clip_threshold of clip_matchORconf (run_stage3.py:560); `conf` / `clip_only` use the hard-wired 0.5 (:522)
STRATEGIES = ["clip_matchORconf", "conf", "cons", "consORconf", "consANDconf", "clip_only", "oracle"]
B_S, B_T, T, C_CLS, MASK_RATIO = 4, 4, 2, 5, 0.5       # (B_S + B_T) must divide B_T * T: run_stage3.py:495 rearranges with the TOTAL batch


class _TorchProxy:
    """the real torch, except torch.cuda.synchronize() / torch.cuda.amp.autocast() on a box without a GPU"""

    class _Cuda:
        amp = types.SimpleNamespace(autocast=contextlib.nullcontext)

        @staticmethod
        def synchronize():
            return None

        # MetricLogger.log_every's no-GPU branch formats its line without the 'total_eta' field the template demands (src/utils.py:354-358:
        # KeyError on a CPU-only box), so the logger is shown a "GPU" that has allocated nothing
        @staticmethod
        def is_available():
            return True

        @staticmethod
        def max_memory_allocated():
            return 0

        def __getattr__(self, name):
            return getattr(torch.cuda, name)

    cuda = _Cuda()

    def __getattr__(self, name):
        return getattr(torch, name)


def extract(names):
    """the named top-level function definitions of run_stage3.py, compiled unchanged"""
    path = os.path.join(MG.REF, "run_stage3.py")
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names)
    return compile(ast.Module(body=body, type_ignores=[]), path, "exec")


class Holder(nn.Module):
    """``model.module`` as under DistributedDataParallel (run_stage3.py:468 reads model.module.encoder.patch_embed.num_patches)"""

    def __init__(self, module):
        super().__init__()
        self.module = module

    def forward(self, *a, **k):
        return self.module(*a, **k)


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    MG._install_standins()
    import einops
    clip_ref = MG._load("src.models.clip", "src/models/clip.py")
    MG._load("src.models.modeling_finetune", "src/models/modeling_finetune.py")
    ad_ref = MG._load("src.models.modeling_adaptation", "src/models/modeling_adaptation.py")
    utils_ref = MG._load("src.utils", "src/utils.py")

    skw = dict(img_size=32, patch_size=16, encoder_embed_dim=128, encoder_depth=3, encoder_num_heads=2, encoder_num_classes=0,
               mlp_ratio=4, qkv_bias=True, norm_layer=partial(nn.LayerNorm, eps=1e-6), num_frames=T, tubelet_size=1,
               clip_decoder_embed_dim=128, clip_output_dim=64, clip_return_layers=[1, 2])
    tkw = dict(input_resolution=32, patch_size=16, width=128, layers=3, heads=2, output_dim=64, return_attn=True, clip_return_layers=[1, 2])
    g = torch.Generator().manual_seed(SEED)
    videos_s, videos_t, videos_t_aug = (make_videos(n, T, 32, 32, seed=sd) for n, sd in ((B_S, 81), (B_T, 82), (B_T, 83)))
    labels_s = torch.randint(0, C_CLS, (B_S,), generator=g)
    labels_t = torch.randint(0, C_CLS, (B_T,), generator=g)
    cls_w = torch.randn(C_CLS, 128, generator=g) * CLS_SCALE
    cls_b = torch.randn(C_CLS, generator=g) * 0.1
    # stand for OpenAI CLIP's encode_image / encode_text outputs (un-normalised).  A common component keeps the cosines close together, so
    # that softmax(100 cos) is not saturated and the frame-averaged probabilities fall on both sides of the 0.5 thresholds
    base = torch.randn(16, generator=g)
    img_feats = base + NOISE * torch.randn(B_T * T, 16, generator=g)
    text_feats = base + NOISE * torch.randn(C_CLS, 16, generator=g)

    ns = {"torch": _TorchProxy(), "nn": nn, "F": F, "einops": einops, "math": math, "sys": sys, "time": time, "np": np,
          "Iterable": Iterable, "utils": utils_ref}
    logged = []
    ns["wandb"] = types.SimpleNamespace(log=lambda d: logged.append(d))
    exec(extract(["pool_outputs", "train_one_epoch"]), ns)
    utils_ref.torch = ns["torch"]          # the same proxy inside src/utils.py (log_every, clip_infer's autocast)

    fake_clip = types.SimpleNamespace(encode_image=lambda images: img_feats.clone())
    utils_ref.setup_clip = lambda args, device: (fake_clip, text_feats.clone())

    out = {"in.seed_student": 3, "in.seed_teacher": 1, "in.videos_s": videos_s, "in.videos_t": videos_t, "in.videos_t_aug": videos_t_aug,
           "in.labels_s": labels_s, "in.labels_t": labels_t, "in.cls_w": cls_w, "in.cls_b": cls_b, "in.img_feats": img_feats,
           "in.text_feats": text_feats, "in.mask_ratio": MASK_RATIO, "in.clip_threshold": CLIP_THRESHOLD, "in.strategies": np.array(STRATEGIES)}
    for strat in STRATEGIES:
        student = ad_ref.AdaptationVisionTransformer(**skw)
        student.load_state_dict(fill_state_dict(MG._shapes(student), seed=3))
        teacher = clip_ref.VisionTransformer(**tkw).eval()
        teacher.load_state_dict(fill_state_dict(MG._shapes(teacher), seed=1))
        cls = nn.Linear(128, C_CLS)
        with torch.no_grad():
            cls.weight.copy_(cls_w)
            cls.bias.copy_(cls_b)
        rec = {"logits": [], "masks": None, "sims": None}
        cls.register_forward_hook(lambda m, i, o: rec["logits"].append(o.detach().clone()))
        greedy, infer = utils_ref.get_greedy_masks, utils_ref.clip_infer

        def greedy_rec(*a, **k):
            rec["masks"] = greedy(*a, **k)
            return rec["masks"]

        def infer_rec(*a, **k):
            rec["sims"] = infer(*a, **k)
            return rec["sims"]
        utils_ref.get_greedy_masks, utils_ref.clip_infer = greedy_rec, infer_rec

        class Scaler:
            def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False):
                loss.backward(create_graph=create_graph)
                self.norm = utils_ref.get_grad_norm_(parameters)
                return self.norm

            def state_dict(self):
                return {"scale": 1.0}
        scaler = Scaler()
        args = types.SimpleNamespace(class_loss_src_ratio=1.0, log_freq=1, epochs=1, selection_strategy=strat, return_aug_for_val=True,
                                     masking_type="clip_attention", use_cls_token=False, clip_threshold=CLIP_THRESHOLD, train_masked=True,
                                     conf_weighted_loss=True, class_loss_tgt_ratio=1.0, full_oracle=False, class_loss_src_ratio_pl=1.0,
                                     nb_classes=C_CLS)
        optimizer = types.SimpleNamespace(param_groups=[], zero_grad=lambda: None)
        logged.clear()
        model = Holder(student)
        local = {}

        def at_return(frame, event, arg):
            # the step's own local variables when train_one_epoch returns (one batch per run, so they are this batch's): the per-clip selection
            # mask, the student's predictions = pseudo-labels, their soft-max confidences and the labels the target loss was taken against
            if event == "return" and frame.f_code.co_name == "train_one_epoch":
                for name in ("sel_mask", "preds_full_t", "msp_t", "ce_target", "sel_mask_cons", "sel_mask_conf"):
                    if name in frame.f_locals and torch.is_tensor(frame.f_locals[name]):
                        local[name] = frame.f_locals[name].detach().clone()
        try:
            sys.setprofile(at_return)
            stats = ns["train_one_epoch"](
                model, [(videos_s, labels_s)], [(videos_t, videos_t_aug, labels_t)], optimizer, torch.device("cpu"), 0, scaler,
                max_norm=None, log_writer=None, lr_scheduler=None, start_steps=0, lr_schedule_values=None, wd_schedule_values=None,
                src_classifier=cls, teacher_model=teacher, clip_input_resolution=32, mask_type="attention", mask_ratio=MASK_RATIO,
                use_wandb=True, args=args)
        finally:
            sys.setprofile(None)
            utils_ref.get_greedy_masks, utils_ref.clip_infer = greedy, infer
        w = logged[0]
        assert local["sel_mask"].shape == (B_T,) and local["preds_full_t"].shape == (B_T,)
        assert abs(float(local["sel_mask"].float().mean()) - float(w["train/select_ratio"])) < 1e-9
        pre = f"{strat}."
        out.update({pre + "loss": stats["loss"], pre + "loss_s": stats["loss_class"], pre + "loss_t": stats["loss_class_t"],
                    pre + "grad_norm": float(scaler.norm), pre + "select_ratio": float(w["train/select_ratio"]),
                    pre + "logits_s": rec["logits"][0], pre + "logits_full_t": rec["logits"][1],
                    pre + "logits_masked": rec["logits"][2].view(2, B_T, C_CLS), pre + "masks": rec["masks"],
                    # per clip, from the reference's own locals: selected or not, the pseudo-label (the student's prediction on the full clip,
                    # run_stage3.py:489,599-602), its confidence; and the labels of the SELECTED clips the target loss used, in clip order
                    pre + "sel_mask": local["sel_mask"].to(torch.uint8), pre + "pseudo_labels": local["preds_full_t"].long(),
                    pre + "msp_t": local["msp_t"].float(),
                    pre + "ce_target": local.get("ce_target", torch.zeros(0, dtype=torch.long)).long()})
        if rec["sims"] is not None:
            out[pre + "similarities"] = rec["sims"]
        for k, p in student.named_parameters():
            if p.grad is not None:
                out[pre + "gnorm." + k] = p.grad.norm()
                if k in GRAD_KEYS and strat in ("clip_matchORconf", "cons"):
                    out[pre + "g." + k] = p.grad
        unused = sorted(k for k, p in student.named_parameters() if p.grad is None)
        assert all(k.startswith("clip_decoder.") for k in unused), unused
        if strat == STRATEGIES[0]:
            pf = rec["logits"][1].softmax(-1)
            print("msp_t", pf.max(-1).values.tolist(), "preds", pf.argmax(-1).tolist(), "clip msp/preds", rec["sims"].max(-1).values.tolist(),
                  rec["sims"].argmax(-1).tolist(), "masked preds", rec["logits"][2].view(2, B_T, C_CLS).argmax(-1).tolist(), "labels_t", labels_t.tolist())
        print(f"{strat:18s} loss {stats['loss']:.6f} = src {stats['loss_class']:.6f} + tgt {stats['loss_class_t']:.6f}  "
              f"select_ratio {w['train/select_ratio']:.2f}  grad_norm {float(scaler.norm):.5f}")
    # how far every decision of the step is from flipping (the HIP step computes the same logits from bf16 operands: measured error <= 0.035):
    # gaps between the two largest logits of the full and the masked passes, distance of the confidences from the thresholds
    lf, lm = out["cons.logits_full_t"], out["cons.logits_masked"]
    top2 = lambda x: (x.topk(2, dim=-1).values[..., 0] - x.topk(2, dim=-1).values[..., 1]).min().item()
    msp = lf.softmax(-1).max(-1).values
    sims = out["clip_only.similarities"].max(-1).values
    margins = {"full_top2_gap": top2(lf), "masked_top2_gap": top2(lm), "msp_to_0.5": (msp - 0.5).abs().min().item(),
               "msp_to_clip_threshold": (msp - CLIP_THRESHOLD).abs().min().item(), "clip_msp_to_0.5": (sims - 0.5).abs().min().item(),
               "clip_msp_to_clip_threshold": (sims - CLIP_THRESHOLD).abs().min().item()}
    ratios = [float(out[s + ".select_ratio"]) for s in STRATEGIES]
    print("MARGINS seed", SEED, "scale", CLS_SCALE, " ".join(f"{k}={v:.3f}" for k, v in margins.items()), "ratios", ratios)
    if os.environ.get("DRY", "0") == "1":
        return
    path = os.path.join(MG.OUT, "stage3_step.npz")
    np.savez_compressed(path, **MG._np(out))
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
