#!/usr/bin/env python
"""Frame-sampling fixture: the REFERENCE's own index arithmetic -- VideoClsDataset_sparse._get_seq_frames
(src/datasets/kinetics_sparse.py:283-312) and VideoMAE._sample_train_indices / _get_frame_id_list (src/datasets/mae.py:253-287) -- on seeded
``random`` / ``numpy.random`` streams -> tests/golden/sampling.json.  The two modules import decord / cv2 / torchvision at the top, which this
image does not have, so they are not imported: the three METHOD definitions are taken out of the files' syntax trees and compiled,
unchanged, with the real ``numpy`` and ``random`` as their globals and a plain namespace as ``self``.  TEST INFRASTRUCTURE."""
import ast
import json
import os
import random
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REF = os.environ.get("UNITE_REFERENCE", "/root/reference")
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")


def methods(relpath, cls, names):
    path = os.path.join(REF, relpath)
    tree = ast.parse(open(path).read(), filename=path)
    body = [n for c in tree.body if isinstance(c, ast.ClassDef) and c.name == cls for n in c.body
            if isinstance(n, ast.FunctionDef) and n.name in names]
    assert sorted(n.name for n in body) == sorted(names), (cls, names)
    for n in body:
        n.decorator_list = []              # @staticmethod: compiled here as a plain function
    ns = {"np": np, "random": random}
    exec(compile(ast.Module(body=body, type_ignores=[]), path, "exec"), ns)
    return [ns[n] for n in names]


def main():
    (get_seq,) = methods("src/datasets/kinetics_sparse.py", "VideoClsDataset_sparse", ["_get_seq_frames"])
    sample, frame_ids = methods("src/datasets/mae.py", "VideoMAE", ["_sample_train_indices", "_get_frame_id_list"])
    out = {"seq_frames": [], "train_indices": []}
    for seed, (video_size, num_frames, clip_idx, skip, mode, tns) in enumerate([
            (300, 8, -1, 0, "train", 1), (37, 16, -1, 0, "train", 1), (9, 8, -1, 0, "train", 1), (1, 4, -1, 0, "train", 1),
            (300, 8, 0, 0, "validation", 1), (61, 16, 0, 0, "validation", 1),
            (300, 8, 0, 0, "test", 4), (300, 8, 3, 0, "test", 4), (50, 16, 2, 0, "test", 5),
            (300, 8, -1, 4, "train", 1), (20, 8, -1, 4, "train", 1), (300, 16, 0, 2, "test", 4)]):
        random.seed(1000 + seed)
        self = types.SimpleNamespace(mode=mode, test_num_segment=tns)
        seq = get_seq(self, video_size, num_frames, clip_idx=clip_idx, skip_frames=skip)
        out["seq_frames"].append(dict(seed=1000 + seed, video_size=video_size, num_frames=num_frames, clip_idx=clip_idx, skip_frames=skip,
                                      mode=mode, test_num_segment=tns, out=[int(v) for v in seq]))
    for seed, (n, segs, skip_length, new_step, jitter) in enumerate([(300, 8, 1, 1, False), (300, 8, 4, 2, True), (10, 8, 1, 1, False),
                                                                     (6, 8, 1, 1, False), (40, 4, 8, 2, True), (5, 2, 8, 4, False)]):
        np.random.seed(2000 + seed)
        self = types.SimpleNamespace(skip_length=skip_length, num_segments=segs, temporal_jitter=jitter, new_step=new_step)
        idx, skip_offsets = sample(self, n)
        ids = frame_ids(self, n, idx, skip_offsets)
        out["train_indices"].append(dict(seed=2000 + seed, num_frames=n, num_segments=segs, skip_length=skip_length, new_step=new_step,
                                         temporal_jitter=jitter, indices=[float(v) for v in idx], skip_offsets=[int(v) for v in skip_offsets],
                                         frame_ids=[int(v) for v in ids]))
    # GroupMultiScaleCrop's crop-box sampler (src/datasets/transforms.py:154-205): scales [1, .875, .75, .66], 13 fixed offsets
    crop_size, fix_offset, fill = methods("src/datasets/transforms.py", "GroupMultiScaleCrop", ["_sample_crop_size", "_sample_fix_offset", "fill_fix_offset"])
    out["crop_boxes"] = []
    for seed, (im_w, im_h, size, fix_crop, more) in enumerate([(340, 256, 224, True, True), (320, 240, 224, True, True), (456, 256, 224, True, False),
                                                                (256, 340, 224, False, True), (224, 224, 224, True, True), (398, 224, 112, True, True)]):
        self = types.SimpleNamespace(scales=[1, .875, .75, .66], max_distort=1, fix_crop=fix_crop, more_fix_crop=more, input_size=[size, size])
        self._sample_fix_offset = lambda *a, self=self: fix_offset(self, *a)
        self.fill_fix_offset = fill
        random.seed(3000 + seed)
        draws = [[int(v) for v in crop_size(self, (im_w, im_h))] for _ in range(6)]
        out["crop_boxes"].append(dict(seed=3000 + seed, im_w=im_w, im_h=im_h, input_size=size, fix_crop=fix_crop, more_fix_crop=more, draws=draws))
    # the loader-side mask generators (src/datasets/masking_generator.py: numpy only, so the module itself is loaded from its file)
    import importlib.util
    spec = importlib.util.spec_from_file_location("ref_masking_generator", os.path.join(REF, "src/datasets/masking_generator.py"))
    mg = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mg)
    out["masks"] = []
    for seed, (kind, size, ratio) in enumerate([("tube", (8, 14, 14), 0.8), ("tube", (2, 2, 2), 0.5), ("tube", (4, 6, 6), 0.75),
                                                 ("random", (8, 14, 14), 0.8), ("random", (2, 3, 3), 0.5)]):
        gen = (mg.TubeMaskingGenerator if kind == "tube" else mg.RandomMaskingGenerator)(size, ratio)
        np.random.seed(4000 + seed)
        draws = ["".join("1" if v else "0" for v in gen()) for _ in range(3)]         # one character per token, 1 = masked
        out["masks"].append(dict(seed=4000 + seed, kind=kind, input_size=list(size), mask_ratio=ratio, draws=draws))
    with open(os.path.join(OUT, "sampling.json"), "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out)[:600])


if __name__ == "__main__":
    main()
