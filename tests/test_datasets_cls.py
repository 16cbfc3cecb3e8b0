"""Stage-2 / stage-3 input path (src/datasets/kinetics_sparse.py ``VideoClsDataset_sparse``, build.py ``build_dataset``) against
tests/golden/dataset_cls.npz -- outputs of the reference's OWN dataset class and RandAugment module, executed by oracle/make_golden_dataset_cls.py
under stand-ins for decord / torchvision / cv2 that take no part in the pinned arithmetic (the fixture's clips need no cv2 resize).
CPU: the RandAugment restatement byte for byte, every worker-side draw (frames, crop box, flip, label, name, view order).
GPU: the device half -- training clips to 5e-6 (f32 bilinear interpolation), validation / test views bit for bit, the OpenCV-style resize
bit for bit against oracle/cv2_resize.py (parity unpinned: cv2 is not in the image)."""
import os
import random
import types

import numpy as np
import pytest
import torch

from oracle.make_golden_dataset_cls import CROP, H, SHORT, T, W, video

POL_ARGS = dict(aa="rand-m7-n4-mstd0.5-inc1", train_interpolation="bicubic", reprob=0.25, remode="pixel", recount=1, data_set="Kinetics_sparse",
                num_sample=1, train_fraction=1.0, return_aug_for_val=False)
TRAIN_CASES = [("train", {}), ("train_erase", dict(reprob=1.0)), ("train_noerase", dict(reprob=0.0)), ("train_ssv2", dict(data_set="SSV2"))]


@pytest.fixture(scope="module")
def z(golden_dir):
    return np.load(os.path.join(golden_dir, "dataset_cls.npz"))


@pytest.fixture(scope="module")
def lists(z, tmp_path_factory):
    """the fixture's five clips (regenerated from their seeds) as .npy files + the two annotation lists"""
    root = tmp_path_factory.mktemp("cls_videos")
    names, labels = [], z["ds.labels"].tolist()
    for v in range(5):
        portrait = v == 3
        np.save(root / f"vid{v}.npy", video(20 + v, int(z["ds.video_frames"][v]), W if portrait else H, H if portrait else W))
        names.append(f"vid{v}.npy")
    (root / "list.txt").write_text("".join(f"{n} {l}\n" for n, l in zip(names, labels)))
    (root / "list_test.txt").write_text("".join(f"{names[i]} {labels[i]}\n" for i in z["ds.test_ids"].tolist()))
    return root


def seed_all(s):
    random.seed(s)
    np.random.seed(s)
    torch.manual_seed(s)


def dataset(lists, mode, **over):
    from unite_amd.datasets_cls import VideoClsDataset_sparse
    args = types.SimpleNamespace(**{**POL_ARGS, **over})
    ann = lists / ("list_test.txt" if mode == "test" else "list.txt")
    return VideoClsDataset_sparse(str(ann), prefix=str(lists), split=" ", mode=mode, clip_len=T, frame_sample_rate=0, crop_size=CROP,
                                  short_side_size=SHORT, test_num_segment=2, test_num_crop=3, args=args)


def test_rand_augment_policies_byte_for_byte(z):
    """unite_amd/rand_augment.py against the reference's rand_augment.py: four policies x two interpolations x six seeds on a three-frame clip"""
    from PIL import Image
    from unite_amd.rand_augment import create_random_augment, parse_policy
    frames = z["ra.frames"]
    changed = 0
    for pi, pol in enumerate(str(p) for p in z["ra.policies"]):
        for interp in ("bicubic", "bilinear"):
            for sd in range(int(z["ra.n_seeds"])):
                random.seed(100 * pi + sd)
                np.random.seed(100 * pi + sd)
                out = create_random_augment((CROP, CROP), pol, interp)([Image.fromarray(f) for f in frames])
                got, ref = np.stack([np.asarray(i) for i in out]), z[f"ra.{pi}.{interp}.{sd}"]
                assert np.array_equal(got, ref), (pol, interp, sd)
                changed += not np.array_equal(ref, frames)
    assert changed >= 30                                           # the policies do something in most cases
    assert parse_policy("rand-m7-n4-mstd0.5-inc1") == (7, 4, None, 0.5, True) and parse_policy("rand-m5-n2-w0") == (5, 2, 0, None, False)
    with pytest.raises(NotImplementedError):
        create_random_augment((CROP, CROP), "augmix-m3", "bilinear")


def test_training_and_validation_draws_equal_the_reference(z, lists):
    """every random decision of a sample, in the reference's order: frame numbers, RandAugment, crop box, flip, erasing -- and the finished clip,
    with torch standing in for the device half on the CPU (crop -> F.interpolate -> flip -> erase): the draws and their semantics are right
    independently of the kernels"""
    for tag, over in TRAIN_CASES:
        ds = dataset(lists, "train", **over)
        for k in range(int(z[f"ds.{tag}.n"])):
            seed_all(300 + k)
            raw, pre = ds[k % 5], f"ds.{tag}.{k}."
            assert np.array_equal(raw["aug_frames"], z[pre + "aug_frames"]), pre
            assert list(raw["crop"]) == z[pre + "crop"].tolist() and raw["flip"] == bool(z[pre + "flipped"]), pre
            assert raw["label"] == int(z[pre + "label"]) and raw["index"] == int(z[pre + "index"])
            if tag == "train_noerase":
                assert raw["erase"] == []
            if tag == "train_erase":
                assert len(raw["erase"]) == 1 and raw["erase"][0][5].shape[:2] == (T, 3)
            fr = torch.from_numpy(raw["aug_frames"]).float().div(255)
            fr = ((fr - torch.tensor([0.485, 0.456, 0.406])) / torch.tensor([0.229, 0.224, 0.225])).permute(3, 0, 1, 2)
            i, j, h, w = raw["crop"]
            out = torch.nn.functional.interpolate(fr[:, :, i:i + h, j:j + w], size=(CROP, CROP), mode="bilinear", align_corners=False)
            out = out.flip(-1) if raw["flip"] else out
            for start, top, left, hh, ww, fill in raw["erase"]:
                out[:, start:, top:top + hh, left:left + ww] = fill.permute(1, 0, 2, 3)
            torch.testing.assert_close(out, torch.from_numpy(z[pre + "out"]), atol=1e-6, rtol=0)
    ds = dataset(lists, "validation", return_aug_for_val=True)
    for idx in range(5):
        seed_all(400 + idx)
        raw, pre = ds[idx], f"ds.val_aug.{idx}."
        assert raw["name"] == str(z[pre + "name"]) == f"vid{idx}" and raw["label"] == int(z[pre + "label"])
        assert np.array_equal(raw["aug_frames"], z[pre + "aug_frames"]) and list(raw["crop"]) == z[pre + "crop"].tolist() and raw["erase"] == []
    ds = dataset(lists, "test")
    assert len(ds) == int(z["ds.test.n"]) == 2 * 3 * 3
    for idx in range(len(ds)):
        raw, pre = ds[idx], f"ds.test.{idx}."
        assert (raw["name"], raw["label"], raw["chunk_nb"], raw["split_nb"]) == (str(z[pre + "name"]), int(z[pre + "label"]), int(z[pre + "chunk"]),
                                                                                 int(z[pre + "split"]))


def test_build_dataset_modes_and_fraction(lists):
    from unite_amd.datasets_cls import build_dataset
    a = types.SimpleNamespace(**POL_ARGS, ann_file_train=str(lists / "list.txt"), ann_file_val=str(lists / "list.txt"),
                              ann_file_test=str(lists / "list_test.txt"), prefix=str(lists), split=" ", num_frames=T, sampling_rate=0,
                              test_num_segment=2, test_num_crop=3, input_size=CROP, short_side_size=SHORT, nb_classes=5)
    tr, nb = build_dataset(True, False, a)
    va, _ = build_dataset(False, False, a)
    te, _ = build_dataset(False, True, a)
    assert (tr.mode, va.mode, te.mode, nb) == ("train", "validation", "test", 5) and (len(tr), len(va), len(te)) == (5, 5, 18)
    a.train_fraction = 0.6
    random.seed(3)
    expect = random.sample(range(5), 3)
    random.seed(3)
    part, _ = build_dataset(True, False, a)
    assert part.dataset_samples == [f"vid{i}.npy" for i in expect] and len(build_dataset(False, False, a)[0]) == 5
    a.data_set = "UCF101"
    with pytest.raises(NotImplementedError):
        build_dataset(True, False, a)


def test_cv2_style_resize_oracle_properties():
    """oracle/cv2_resize.py (OpenCV's published 8-bit linear resize; parity unpinned): identities it must satisfy"""
    from oracle.cv2_resize import resize_linear_u8, resize_sizes
    rng = np.random.RandomState(0)
    img = rng.randint(0, 256, size=(2, 30, 44, 3), dtype=np.uint8)
    assert np.array_equal(resize_linear_u8(img, 30, 44), img)                                   # same size: a copy
    flat = np.full((1, 17, 23, 3), 77, np.uint8)
    assert np.array_equal(resize_linear_u8(flat, 40, 31), np.full((1, 40, 31, 3), 77, np.uint8))        # weights sum to 2048 everywhere
    up = resize_linear_u8(img, 60, 88)                                                          # exact 2x: output (21, 41) sits a quarter past source (10, 20)
    px = img[0].astype(np.int64)
    h0, h1 = px[10, 20] * 1536 + px[10, 21] * 512, px[11, 20] * 1536 + px[11, 21] * 512          # horizontal pass, weights 0.75 / 0.25 in 11 bits
    want = (((1536 * (h0 >> 4)) >> 16) + ((512 * (h1 >> 4)) >> 16) + 2) >> 2                     # vertical pass in OpenCV's two-shift form
    assert up.shape == (2, 60, 88, 3) and np.array_equal(up[0, 21, 41], want.astype(np.uint8))
    ramp = np.tile(np.arange(44, dtype=np.uint8)[None, None, :, None] * 5, (1, 30, 1, 3))
    dn = resize_linear_u8(ramp, 15, 22)                                                         # a horizontal ramp stays monotone and constant down columns
    assert (np.diff(dn[0, 0, :, 0].astype(int)) >= 0).all() and (dn[0] == dn[0, :1]).all()
    assert resize_sizes(240, 320, 224) == (224, 298) and resize_sizes(320, 240, 224) == (298, 224) and resize_sizes(256, 256, 224) == (224, 224)


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_device_batches_equal_the_reference_dataset(z, lists):
    from unite_amd.datasets import DeviceLoader
    dev = torch.device("cuda:0")
    for tag, over in TRAIN_CASES:                                                               # training clips, one sample per batch and seed
        ds = dataset(lists, "train", **over)
        for k in range(int(z[f"ds.{tag}.n"])):
            seed_all(300 + k)
            videos, labels, idx, extra = ds.transform.batch([ds[k % 5]], dev)
            pre = f"ds.{tag}.{k}."
            assert tuple(videos.shape) == (1, 3, T, CROP, CROP) and labels.tolist() == [int(z[pre + "label"])] and idx.tolist() == [k % 5] and extra == {}
            torch.testing.assert_close(videos[0].cpu(), torch.from_numpy(z[pre + "out"]), atol=5e-6, rtol=0)
    ds = dataset(lists, "validation")                                                           # validation: bit for bit (no resize needed here)
    loader = DeviceLoader(ds, 5, dev, sampler=None, num_workers=0, drop_last=False)
    (videos, labels, names), = list(loader)
    assert names == [f"vid{i}" for i in range(5)] and labels.tolist() == z["ds.labels"].tolist()
    for i in range(5):
        assert torch.equal(videos[i].cpu(), torch.from_numpy(z[f"ds.val.{i}.vids"])), i
    ds = dataset(lists, "validation", return_aug_for_val=True)
    for i in range(5):
        seed_all(400 + i)
        vids, aug, labels, names = ds.transform.batch([ds[i]], dev)
        assert torch.equal(vids[0].cpu(), torch.from_numpy(z[f"ds.val_aug.{i}.vids"]))
        torch.testing.assert_close(aug[0].cpu(), torch.from_numpy(z[f"ds.val_aug.{i}.vids_aug"]), atol=5e-6, rtol=0)
    ds = dataset(lists, "test")                                                                 # test views: bit for bit, three crops x two chunks
    for i in range(len(ds)):
        out, labels, names, chunk, split = ds.transform.batch([ds[i]], dev)
        pre = f"ds.test.{i}."
        assert torch.equal(out[0].cpu(), torch.from_numpy(z[pre + "out"])), i
        assert (chunk.tolist(), split.tolist()) == ([int(z[pre + "chunk"])], [int(z[pre + "split"])])


@pytest.mark.gpu
@pytest.mark.parametrize("shape", [(3, 240, 320, 224, 298), (2, 320, 240, 298, 224), (2, 256, 340, 224, 297), (1, 100, 60, 373, 224), (2, 37, 53, 37, 80)])
def test_resize_u8_linear_bit_exact_vs_published_algorithm(shape):
    """unite_resize_u8_linear against oracle/cv2_resize.py (down- and up-scaling, both orientations).  Parity unpinned: see the oracle's header"""
    from oracle.cv2_resize import resize_linear_u8
    from unite_amd import ops
    Tn, Hn, Wn, OH, OW = shape
    rng = np.random.RandomState(Hn + Wn)
    fr = rng.randint(0, 256, size=(Tn, Hn, Wn, 3), dtype=np.uint8)
    out = ops.resize_u8_linear(torch.from_numpy(fr).cuda(), torch.empty(Tn, OH, OW, 3, dtype=torch.uint8, device="cuda"))
    assert np.array_equal(out.cpu().numpy(), resize_linear_u8(fr, OH, OW))


@pytest.mark.gpu
def test_validation_view_with_a_resize(lists):
    """a clip whose short side is NOT the requested one goes through resize -> centre crop -> normalise: equal to the oracle's resize followed
    by the reference's arithmetic for the rest (crop offsets, / 255, normalise)"""
    from oracle.cv2_resize import resize_linear_u8, resize_sizes
    from unite_amd.datasets_cls import DeviceClsTransform, MEAN, STD
    rng = np.random.RandomState(5)
    fr = rng.randint(0, 256, size=(4, 60, 90, 3), dtype=np.uint8)
    tf = DeviceClsTransform("validation", crop_size=32, short_side_size=40)
    got = tf._val_clip(fr, torch.device("cuda:0")).cpu()
    oh, ow = resize_sizes(60, 90, 40)
    rs = resize_linear_u8(fr, oh, ow)
    y1, x1 = int(round((oh - 32) / 2.)), int(round((ow - 32) / 2.))
    ref = torch.from_numpy(rs[:, y1:y1 + 32, x1:x1 + 32]).float().div(255)
    ref = ((ref - torch.tensor(MEAN)) / torch.tensor(STD)).permute(3, 0, 1, 2)
    assert torch.equal(got, ref)
