"""Evaluation path (engine_for_finetuning.py:174-352, run_stage3.py:714-787).  Host-side pieces on CPU; the model passes on the GPU
against the oracle's logits."""
import os
from functools import partial
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from unite_amd import engine_for_finetuning as E


def test_accuracy_ece_merge_host_logic(tmp_path):
    out = torch.tensor([[3., 1., 0., 0., 0., 0.], [0., 5., 1., 0., 0., 0.], [0., 0., 1., 2., 0., 0.], [1., .5, .4, .3, .2, 9.]])
    tgt = torch.tensor([0, 2, 3, 1])
    a1, a5 = E.accuracy(out, tgt, topk=(1, 5))
    assert a1.item() == 50.0 and a5.item() == 100.0          # sample 1: label 2 is rank 2; sample 3: label 1 is rank 3
    # ECE by hand: two bins
    sm = torch.tensor([[0.9, 0.1], [0.9, 0.1], [0.6, 0.4], [0.6, 0.4]])
    lab = torch.tensor([0, 1, 0, 0])
    assert abs(E.compute_ece(sm, lab) - (0.5 * abs(0.5 - 0.9) + 0.5 * abs(1.0 - 0.6))) < 1e-6
    # merge: two ranks, duplicate view dropped, mean of per-view soft-max
    with open(tmp_path / "0.txt", "w") as f:
        f.write("50.0, 100.0\n")
        f.write("vidA [2.0, 0.0, 0.0] 0 0 0\n")
        f.write("vidA [0.0, 3.0, 0.0] 0 1 0\n")
        f.write("vidB [0.0, 0.0, 1.0] 1 0 0\n")
    with open(tmp_path / "1.txt", "w") as f:
        f.write("0.0, 0.0\n")
        f.write("vidA [0.0, 3.0, 0.0] 0 1 0\n")           # same (chunk, split): ignored
        f.write("vidB [0.0, 4.0, 0.0] 1 1 0\n")
    top1, top5 = E.merge(str(tmp_path), 2)
    # vidA: mean(softmax([2,0,0]), softmax([0,3,0])) -> class 1 wins (0.906+0.042)/2=0.474 vs (0.787+0.042)/2=0.414: wrong (label 0)
    # vidB: mean(softmax([0,0,1]), softmax([0,4,0])) -> class 1 (label 1): right
    assert top1 == 50.0 and top5 == 100.0


@pytest.mark.gpu
def test_validation_and_final_test_on_gpu(tmp_path):
    from oracle import umt_oracle as O
    from oracle.filler import fill_state_dict, make_videos
    from tests.shapes import TINY_V, vit_shapes
    from unite_amd.modeling_finetune import VisionTransformer
    m = VisionTransformer(img_size=32, patch_size=16, embed_dim=128, depth=2, num_heads=2, mlp_ratio=4, qkv_bias=True,
                          norm_layer=partial(torch.nn.LayerNorm, eps=1e-6), num_classes=5, all_frames=4, tubelet_size=1,
                          use_mean_pooling=True, init_scale=1.0)
    sd = fill_state_dict(vit_shapes(TINY_V), 17)
    sd["head.weight"] = sd["head.weight"] * 40            # spread the logits so that the predictions are not all ties
    m.load_state_dict(sd)
    m = m.to("cuda")
    vids = [make_videos(3, 4, 32, 32, seed=50 + i) for i in range(2)]
    labels = [torch.tensor([0, 3, 1]), torch.tensor([4, 4, 2])]
    ref_logits = torch.cat([O.vit_classifier_forward(sd, v, TINY_V) for v in vids])
    ref_loss = np.mean([torch.nn.functional.cross_entropy(ref_logits[3 * i:3 * i + 3], labels[i]).item() for i in range(2)])
    ref_a1 = (ref_logits.argmax(1) == torch.cat(labels)).float().mean().item() * 100
    stats, ece = E.validation_one_epoch(list(zip(vids, labels)), m, torch.device("cuda"), save_preds_path=str(tmp_path / "p"))
    assert abs(stats["loss"] - ref_loss) <= 2e-2 * max(1.0, abs(ref_loss))
    assert abs(stats["acc1"] - ref_a1) < 1e-4 and 0.0 <= ece <= 1.0
    assert np.array_equal(np.load(tmp_path / "p" / "preds.npy"), ref_logits.argmax(1).numpy())
    loader = [(v, l, [f"clip{i}_{j}" for j in range(3)], torch.tensor([0, 0, 0]), torch.tensor([i, i, i])) for i, (v, l) in enumerate(zip(vids, labels))]
    stats2, _ = E.final_test(loader, m, torch.device("cuda"), str(tmp_path / "0.txt"))
    assert abs(stats2["acc1"] - ref_a1) < 1e-4
    top1, _ = E.merge(str(tmp_path), 1)
    assert abs(top1 - ref_a1) < 1e-4                      # one view per clip: merge reduces to the per-clip accuracy


@pytest.mark.gpu
def test_stage3_validation_on_gpu():
    import tests.test_stage3_gpu as T
    from oracle import umt_oracle as O
    from unite_amd.engine_stage3 import validation_one_epoch
    s, t, cls, ssd, tsd, d = T._setup(seed=1)
    vids, lab = d["videos_t"], torch.tensor([2, 3, 0, 2])
    x, _ = O.student_forward(ssd, vids, torch.zeros(4, 32, dtype=torch.bool), T.S3_S, clip_only=False)
    ref = torch.nn.functional.linear(x.mean(1), cls.weight.detach().cpu(), cls.bias.detach().cpu())
    args = SimpleNamespace(return_aug_for_val=True, use_cls_token=False)
    stats = validation_one_epoch([(vids, vids, lab)], s, cls, torch.device("cuda"), args=args)
    assert abs(stats["loss"] - torch.nn.functional.cross_entropy(ref, lab).item()) <= 2.5e-2
    assert abs(stats["acc1"] - (ref.argmax(1) == lab).float().mean().item() * 100) < 1e-4


@pytest.mark.gpu
def test_stage3_final_test_on_gpu(tmp_path):
    """run_stage3.py:927-989 + merge: the per-view lines carry encoder -> mean pool -> source classifier logits (checked against the oracle's,
    line by line), two views per clip merge into the per-clip accuracy of the mean soft-max."""
    import tests.test_stage3_gpu as T
    from oracle import umt_oracle as O
    from unite_amd.engine_stage3 import final_test
    s, t, cls, ssd, tsd, d = T._setup(seed=1)
    views = [d["videos_t"], d["videos_s"]]                       # two "views" (chunk 0 / 1) of four clips
    lab = torch.tensor([2, 3, 0, 2])
    refs = []
    for v in views:
        x, _ = O.student_forward(ssd, v, torch.zeros(4, 32, dtype=torch.bool), T.S3_S, clip_only=False)
        refs.append(torch.nn.functional.linear(x.mean(1), cls.weight.detach().cpu(), cls.bias.detach().cpu()))
    loader = [(v, lab, [f"clip{j}" for j in range(4)], torch.tensor([c] * 4), torch.tensor([0] * 4)) for c, v in enumerate(views)]
    args = SimpleNamespace(use_cls_token=False)
    stats = final_test(loader, s, cls, torch.device("cuda"), str(tmp_path / "0.txt"), args)
    ref_a1 = (torch.cat(refs).argmax(1) == torch.cat([lab, lab])).float().mean().item() * 100
    assert abs(stats["acc1"] - ref_a1) < 1e-4 and stats["loss"] > 0
    lines = open(tmp_path / "0.txt").read().splitlines()
    assert len(lines) == 1 + 8
    for c in range(2):
        for j in range(4):
            name, _, rest = lines[1 + 4 * c + j].partition(" [")
            vec, _, tail = rest.partition("]")
            got = np.array([float(x) for x in vec.split(",")])
            assert name == f"clip{j}" and tail.split() == [str(int(lab[j])), str(c), "0"]
            np.testing.assert_allclose(got, refs[c][j].numpy(), atol=4e-2, rtol=1e-2)
    top1, top5 = E.merge(str(tmp_path), 1)
    mean_sm = (torch.softmax(refs[0].double(), 1) + torch.softmax(refs[1].double(), 1)) / 2
    assert abs(top1 - (mean_sm.argmax(1) == lab).double().mean().item() * 100) < 1e-9 and top5 >= top1
