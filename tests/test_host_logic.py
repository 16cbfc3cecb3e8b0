"""CPU-only: host-side mirrors of the reference API (state_dict contracts, parameter grouping, schedules, masks,
flat-buffer layout) against the reference-generated golden vectors."""
import os
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import umt_oracle as O
from tests.shapes import TINY_S, TINY_T, student_shapes, teacher_shapes, vit_shapes


def _load(golden_dir, name):
    z = np.load(os.path.join(golden_dir, name))
    return {k: z[k] for k in z.files}


def test_full_size_state_dict_contract():
    """184 tensors / 88,005,888 parameters (student), OpenAI visual.* layout (teacher)"""
    import unite_amd
    s = unite_amd.create_model("adaptation_umt_base_patch16_224", pretrained=False, drop_path_rate=0.1, drop_block_rate=None,
                               num_frames=8, tubelet_size=1, clip_decoder_embed_dim=768, clip_output_dim=512,
                               clip_return_layers=[6, 7, 8, 9, 10, 11], use_cls_token=False)
    sd = s.state_dict()
    assert [(k, tuple(v.shape)) for k, v in sd.items()] == student_shapes(O.StudentCfg())
    assert len(sd) == 184 and sum(v.numel() for v in sd.values()) == 88005888
    assert s.encoder.patch_embed.num_patches == 1568 and s.encoder.patch_embed.patch_size == (16, 16)
    assert [round(b.drop_path_rate, 6) for b in s.encoder.blocks] == [round(x.item(), 6) for x in torch.linspace(0, 0.1, 12)]
    assert s.no_weight_decay() == {'pos_embed', 'cls_token', 'mask_token', 'clip_mask_token', 'clip_pos_embed'}
    t = unite_amd.clip.clip_b16(pretrained=False, return_attn=True)
    assert sorted((k, tuple(v.shape)) for k, v in t.state_dict().items()) == sorted(teacher_shapes(O.TeacherCfg()))
    assert not t.training
    torch.testing.assert_close(s.encoder.pos_embed, O.sinusoid_table(1568, 768))
    with pytest.raises(RuntimeError):
        unite_amd.create_model("no_such_model")


def test_parameter_groups_match_reference(golden_dir):
    from unite_amd.optim_factory import get_parameter_groups, LayerDecayValueAssigner, get_num_layer_for_vit
    from tests.test_model_gpu import build_tiny
    z = _load(golden_dir, "student_tiny.npz")
    s, _ = build_tiny()
    groups, names = get_parameter_groups(s, 0.05, s.no_weight_decay(), with_names=True)
    by = {("decay" if g["weight_decay"] > 0 else "no_decay"): g["params"] for g in names}
    assert by["decay"] == list(z["groups.decay"]) and by["no_decay"] == list(z["groups.no_decay"])
    # layer decay is inert on encoder.* names (SURVEY A-7) and live on stage-2 names
    u = _load(golden_dir, "utils.npz")
    nl = 2
    assigner = LayerDecayValueAssigner([0.65 ** (nl + 1 - i) for i in range(nl + 2)])
    for name, sc in zip(u["layer.names"], u["layer.scale"]):
        assert assigner.get_scale(assigner.get_layer_id(str(name))) == pytest.approx(float(sc))
    assert get_num_layer_for_vit("encoder.blocks.3.attn.qkv.weight", 14) == 13


def test_schedules_and_greedy_masks(golden_dir):
    from unite_amd.utils import cosine_scheduler, get_greedy_masks
    u = _load(golden_dir, "utils.npz")
    np.testing.assert_allclose(cosine_scheduler(1.5e-4, 1e-5, 4, 5, warmup_epochs=1), u["cos.a"], rtol=1e-12)
    np.testing.assert_allclose(cosine_scheduler(1.5e-4, 1e-5, 3, 7, warmup_epochs=1, warmup_steps=4, start_warmup_value=1e-6),
                               u["cos.b"], rtol=1e-12)
    attn = torch.from_numpy(u["greedy.attn"])
    assert np.array_equal(get_greedy_masks(attn, 0.75, 2).numpy(), u["greedy.k2_r075"])
    assert np.array_equal(get_greedy_masks(attn, 0.8, 3).numpy(), u["greedy.k3_r08"])


def test_flat_layout_packs_qkv_bias():
    """offset arithmetic of the flat store (no device needed): q_bias, zero gap, v_bias contiguous; 1024-element granules"""
    from unite_amd.flat_params import FlatParams, CHUNK
    from tests.test_model_gpu import build_tiny
    s, _ = build_tiny()
    fp = FlatParams.__new__(FlatParams)
    named = list(s.named_parameters())
    # replicate the offset pass only
    fp.names = [n for n, _ in named]
    off, offsets, packed = 0, {}, {}
    i = 0
    while i < len(named):
        n, p = named[i]
        if n.endswith("attn.q_bias"):
            d = p.numel()
            offsets[n] = off
            offsets[named[i + 1][0]] = off + 2 * d
            packed[n[:-6]] = (off, 3 * d)
            off += (3 * d + CHUNK - 1) // CHUNK * CHUNK
            i += 2
            continue
        offsets[n] = off
        off += (p.numel() + CHUNK - 1) // CHUNK * CHUNK
        i += 1
    assert all(o % CHUNK == 0 or n.endswith("v_bias") for n, o in offsets.items())
    for pre, (o, k) in packed.items():
        assert offsets[pre + "v_bias"] - offsets[pre + "q_bias"] == 2 * (k // 3)


def test_metric_logger_and_scaler_surface():
    from unite_amd.utils import MetricLogger, SmoothedValue, NativeScalerWithGradNormCount
    m = MetricLogger(delimiter="  ")
    m.add_meter('lr', SmoothedValue(window_size=1, fmt='{value:.6f}'))
    m.update(loss=2.0, lr=1e-4, weight_decay=None)
    m.update(loss=1.0)
    assert m.meters["loss"].global_avg == 1.5 and "weight_decay" not in m.meters
    assert NativeScalerWithGradNormCount().state_dict() == {"scale": 1.0}
    assert list(m.log_every(range(3), 0)) == [0, 1, 2]


def test_distributed_sampler_matches_reference_golden(golden_dir):
    """unite_amd.data.DistributedSampler (shuffle / seed + epoch / drop_last / repetitions / wrap-around padding) against the index
    streams of the reference's sampler for every rank (tests/golden/sampler.json, written by oracle/make_golden_sampler.py from
    src/datasets/distributed.py:81-163); plus the properties data-parallel training relies on."""
    import json
    from unite_amd.data import DistributedSampler
    cases = json.load(open(os.path.join(golden_dir, "sampler.json")))["cases"]
    assert len(cases) >= 10
    for c in cases:
        ds = list(range(c["n"]))
        seen = []
        for r in range(c["num_replicas"]):
            s = DistributedSampler(ds, num_replicas=c["num_replicas"], rank=r, shuffle=c["shuffle"], seed=c["seed"],
                                   drop_last=c["drop_last"], repetitions=c["repetitions"])
            s.set_epoch(c["epoch"])
            idx = list(iter(s))
            assert idx == c["indices"][r], (c["n"], c["num_replicas"], r)
            assert len(s) == len(idx)
            assert list(iter(s)) == idx                       # same epoch -> same stream
            seen += idx
        # every rank draws the same number of samples; together they cover the epoch (each sample `repetitions` times when the
        # world size divides the epoch, at least floor(.) times with drop_last, at least that often with padding)
        total = c["n"] * c["repetitions"]
        counts = np.bincount(np.array(seen), minlength=c["n"])
        if total % c["num_replicas"] == 0:
            assert (counts == c["repetitions"]).all()
        elif c["drop_last"]:
            assert counts.sum() == total - total % c["num_replicas"] and counts.max() <= c["repetitions"]
        else:
            assert (counts >= c["repetitions"]).all()
    with pytest.raises(ValueError):
        DistributedSampler([0, 1], num_replicas=2, rank=2)


def test_sparse_frame_sampling_properties():
    """unite_amd.data.sample_train_indices / frame_id_list (reference src/datasets/mae.py:253-287; parity unpinned -- the reference
    module cannot be imported without decord / cv2): one frame per segment, inside its segment, inside the video, reproducible from
    the generator state, degenerate videos handled as the reference does (zeros)."""
    from unite_amd.data import frame_id_list, sample_train_indices
    rs = np.random.RandomState(0)
    for num_frames, segs in [(300, 8), (64, 16), (17, 16), (16, 16), (9, 8)]:
        off, skip = sample_train_indices(num_frames, segs, rng=rs)
        assert len(off) == segs and len(skip) == 1 and (skip == 0).all()
        seg_len = num_frames // segs
        ids = frame_id_list(num_frames, off, skip)
        assert len(ids) == segs and all(0 <= i < num_frames for i in ids)
        if seg_len > 0:
            for k, i in enumerate(ids):
                assert k * seg_len <= i < (k + 1) * seg_len
        assert ids == sorted(ids)
    off, _ = sample_train_indices(5, 8, rng=rs)                  # shorter than the segment count: every frame id is 0
    assert frame_id_list(5, off, [0]) == [0] * 8
    a = sample_train_indices(300, 8, skip_length=4, new_step=2, temporal_jitter=True, rng=np.random.RandomState(3))
    b = sample_train_indices(300, 8, skip_length=4, new_step=2, temporal_jitter=True, rng=np.random.RandomState(3))
    assert (a[0] == b[0]).all() and (a[1] == b[1]).all() and len(a[1]) == 2 and set(a[1]) <= {0, 1}
    ids = frame_id_list(300, a[0], a[1], skip_length=4, new_step=2)
    assert len(ids) == 16 and all(0 <= i < 300 for i in ids)


def test_driver_side_utils_surface():
    """every ``utils.*`` name the reference drivers touch (run_stage1/2/3.py) exists in unite_amd.utils with the reference's behaviour
    for the pure-host ones (utils.py:84-87, 854-909)"""
    from unite_amd import utils
    for name in ["is_main_process", "get_world_size", "get_rank", "cosine_scheduler", "seed_worker", "load_state_dict", "count_parameters",
                 "auto_load_model", "save_latest_model", "SmoothedValue", "MetricLogger", "save_model", "init_distributed_mode",
                 "TensorboardLogger", "step_scheduler", "save_on_master", "experiment_exists", "confirm_exp_overwrite", "clip_infer",
                 "setup_clip", "get_greedy_masks", "create_ds_config", "multiple_samples_collate", "multiple_pretrain_samples_collate",
                 "str2bool", "get_class_names", "NativeScalerWithGradNormCount", "get_grad_norm_"]:
        assert hasattr(utils, name), name
    assert utils.str2bool("True") and utils.str2bool(True) and not utils.str2bool("no")
    assert [len(utils.get_class_names(SimpleNamespace(nb_classes=n))) for n in (8, 12, 23)] == [8, 12, 23]
    batch = [([torch.zeros(2), torch.ones(2)], [1, 2], [7, 7], {"a": 1}), ([torch.zeros(2), torch.ones(2)], [3, 4], [8, 8], {"a": 2})]
    x, y, idx, extra = utils.multiple_samples_collate(batch)
    assert x.shape == (4, 2) and y.tolist() == [1, 2, 3, 4] and idx.tolist() == [7, 7, 8, 8] and extra["a"].tolist() == [1, 2]
    assert isinstance(utils.multiple_samples_collate(batch, fold=True)[0], list)
    assert utils.count_parameters(torch.nn.Linear(3, 4)) == 16
    with pytest.raises(NotImplementedError):
        utils.setup_clip(None, None)


def test_frame_sampling_vs_reference_golden(golden_dir):
    """data.get_seq_frames / sample_train_indices / frame_id_list against the reference's own methods (kinetics_sparse.py:283-312,
    mae.py:253-287 compiled from the files' syntax trees by oracle/make_golden_sampling.py) on the same seeded ``random`` / ``numpy.random``
    streams: training, validation, multi-view test and skip-strategy clips, videos shorter than the clip, jitter on and off."""
    import json
    import os
    import random
    import numpy as np
    from unite_amd import data
    z = json.load(open(os.path.join(golden_dir, "sampling.json")))
    assert len(z["seq_frames"]) >= 12 and len(z["train_indices"]) >= 6
    for c in z["seq_frames"]:
        random.seed(c["seed"])
        got = data.get_seq_frames(c["video_size"], c["num_frames"], clip_idx=c["clip_idx"], skip_frames=c["skip_frames"], mode=c["mode"],
                                  test_num_segment=c["test_num_segment"])
        assert [int(v) for v in got] == c["out"], c
    for c in z["train_indices"]:
        np.random.seed(c["seed"])
        idx, skip = data.sample_train_indices(c["num_frames"], c["num_segments"], c["skip_length"], c["new_step"], c["temporal_jitter"])
        assert [float(v) for v in idx] == c["indices"] and [int(v) for v in skip] == c["skip_offsets"], c
        assert [int(v) for v in data.frame_id_list(c["num_frames"], idx, skip, c["skip_length"], c["new_step"])] == c["frame_ids"], c


def test_crop_box_sampler_vs_reference_golden(golden_dir):
    """data.MultiScaleCrop against GroupMultiScaleCrop._sample_crop_size (src/datasets/transforms.py:154-205, compiled from the file's syntax
    tree by oracle/make_golden_sampling.py) on the same seeded ``random`` stream; the reference returns (w, h, x0, y0)."""
    import json
    import os
    import random
    from unite_amd import data
    z = json.load(open(os.path.join(golden_dir, "sampling.json")))
    assert len(z["crop_boxes"]) >= 6
    for c in z["crop_boxes"]:
        crop = data.MultiScaleCrop(c["input_size"], fix_crop=c["fix_crop"], more_fix_crop=c["more_fix_crop"])
        random.seed(c["seed"])
        for w, h, x0, y0 in c["draws"]:
            assert crop(c["im_w"], c["im_h"]) == (x0, y0, w, h), c


def test_pil_resize_oracle_vs_pillow():
    """oracle/pil_resize.py (the checker of unite_crop_resize_u8) against Pillow itself: crop + Image.BILINEAR resize of random uint8
    images, up- and down-scaling, bit for bit.  (Pillow is the reference's own resampler, transforms.py:136-152.)"""
    import numpy as np
    PIL_Image = pytest.importorskip("PIL.Image")
    from oracle.pil_resize import crop_resize_bilinear
    rng = np.random.RandomState(1)
    for _ in range(12):
        H, W = int(rng.randint(30, 300)), int(rng.randint(30, 300))
        img = rng.randint(0, 256, (H, W, 3), dtype=np.uint8)
        w, h = int(rng.randint(8, W + 1)), int(rng.randint(8, H + 1))
        x0, y0 = int(rng.randint(0, W - w + 1)), int(rng.randint(0, H - h + 1))
        OW, OH = int(rng.choice([224, 112, 37])), int(rng.choice([224, 112, 37]))
        ref = np.asarray(PIL_Image.fromarray(img).crop((x0, y0, x0 + w, y0 + h)).resize((OW, OH), PIL_Image.BILINEAR))
        assert np.array_equal(ref, crop_resize_bilinear(img, (x0, y0, w, h), (OH, OW)))


def test_loader_mask_generators_vs_reference_golden(golden_dir):
    """TubeMaskingGenerator / RandomMaskingGenerator (src/datasets/masking_generator.py, loaded from the reference file by
    oracle/make_golden_sampling.py): the same masks from the same numpy seed, draw after draw."""
    import json
    import numpy as np
    from unite_amd.datasets import TubeMaskingGenerator, RandomMaskingGenerator
    cases = json.load(open(os.path.join(golden_dir, "sampling.json")))["masks"]
    assert {c["kind"] for c in cases} == {"tube", "random"}
    for c in cases:
        gen = (TubeMaskingGenerator if c["kind"] == "tube" else RandomMaskingGenerator)(tuple(c["input_size"]), c["mask_ratio"])
        np.random.seed(c["seed"])
        for want in c["draws"]:
            got = gen()
            assert "".join("1" if v else "0" for v in got) == want
            if c["kind"] == "tube":
                T, per = c["input_size"][0], c["input_size"][1] * c["input_size"][2]
                assert (got.reshape(T, per) == got[:per]).all() and got[:per].sum() == int(c["mask_ratio"] * per)


def test_video_mae_dataset_draws_and_reads(tmp_path):
    """unite_amd.datasets.VideoMAE over .npy videos: annotation parsing (mae.py:229-251), the frame numbers of a sample equal
    frame_id_list(sample_train_indices(...)) on the same numpy stream (pinned to the reference by sampling.json), the crop box and flip come
    from the `random` stream in the reference's order (crop first, then one random.random() whether or not flipping is on), an unreadable
    clip is replaced by another one, and the sample carries the raw frames of exactly those numbers."""
    import random
    import types
    import numpy as np
    from unite_amd import data as D
    from unite_amd.datasets import VideoMAE, DeviceAugmentationForVideoMAE, read_annotations
    rng = np.random.RandomState(0)
    sizes = [(40, 48, 64), (25, 36, 50), (9, 40, 40)]
    lines = []
    for i, (F, H, W) in enumerate(sizes):
        np.save(tmp_path / f"v{i}.npy", rng.randint(0, 256, size=(F, H, W, 3), dtype=np.uint8))
        lines.append(f"v{i}.npy {i % 2}")
    lines.append("missing.npy 1")
    (tmp_path / "train.txt").write_text("\n".join(lines) + "\n")
    assert read_annotations(str(tmp_path / "train.txt")) == [("v0.npy", 0), ("v1.npy", 1), ("v2.npy", 0), ("missing.npy", 1)]
    with pytest.raises(RuntimeError):
        read_annotations(str(tmp_path / "nope.txt"))
    args = types.SimpleNamespace(input_size=32, mask_type="tube", mask_ratio=0.75, window_size=(8, 2, 2), color_jitter=0.0, flip=True)
    ds = VideoMAE(None, str(tmp_path / "train.txt"), prefix=str(tmp_path), num_segments=8, new_length=8, new_step=1,
                  transform=DeviceAugmentationForVideoMAE(args), video_loader=True, use_decord=True)
    assert len(ds) == 4 and ds.new_length == 8 and ds.skip_length == 1
    for index, (F, H, W) in enumerate(sizes):
        np.random.seed(50 + index)
        random.seed(60 + index)
        frames, box, flip, mask, target = ds[index]
        np.random.seed(50 + index)
        random.seed(60 + index)
        idx, skip = D.sample_train_indices(F, 8, 1, 1, False)
        ids = D.frame_id_list(F, idx, skip, 1, 1)
        want = np.load(tmp_path / f"v{index}.npy")[ids]
        assert frames.dtype == torch.uint8 and tuple(frames.shape) == (8, H, W, 3) and np.array_equal(frames.numpy(), want)
        assert box == D.MultiScaleCrop(32)(W, H) and flip == (random.random() < 0.5)
        assert mask.shape == (32,) and mask.sum() == 8 * 3 and target == index % 2
        x0, y0, w, h = box
        assert 0 <= x0 and x0 + w <= W and 0 <= y0 and y0 + h <= H
    random.seed(3)
    frames, _, _, _, target = ds[3]                       # 'missing.npy' cannot be read: another clip takes its place (mae.py:206-209)
    assert tuple(frames.shape)[0] == 8 and target in (0, 1)


def test_hardware_queue_rule_is_decided_from_the_environment():
    """unite_amd/hwqueues.py: eight HIP hardware queues wherever a rank has a GPU to itself, the default pool where ranks share one; decided
    from the environment and the KFD topology alone (round-3 advisor finding: the old guard called torch.cuda.device_count() at import and
    misread one-device-per-rank masks)."""
    from unite_amd.hwqueues import choose, mask_entries
    assert choose({}, 8)[0] == "8" and choose({}, None)[0] == "8" and choose({}, 1)[0] == "8"
    assert choose({"LOCAL_WORLD_SIZE": "8"}, 8)[0] == "8"                                     # the case the setting exists for
    v, why = choose({"LOCAL_WORLD_SIZE": "2"}, 1)                                             # two ranks rehearsing on one GPU
    assert v is None and "share" in why
    assert choose({"LOCAL_WORLD_SIZE": "8", "HIP_VISIBLE_DEVICES": "3"}, 8)[0] == "8"         # a launcher that masks one GPU per rank
    assert choose({"LOCAL_WORLD_SIZE": "8", "ROCR_VISIBLE_DEVICES": "0,1,2,3"}, 8)[0] is None  # eight ranks on four visible devices
    assert choose({"LOCAL_WORLD_SIZE": "4", "CUDA_VISIBLE_DEVICES": "0,1,2,3"}, 8)[0] == "8"
    v, why = choose({"GPU_MAX_HW_QUEUES": "4", "LOCAL_WORLD_SIZE": "8"}, 8)
    assert v is None and "set by the caller" in why
    v, why = choose({"LOCAL_WORLD_SIZE": "2", "HIP_VISIBLE_DEVICES": "5", "UNITE_RANKS_SHARE_GPU": "1"}, 8)      # bench.py's self-start on a one-GPU box
    assert v is None and "share" in why
    assert choose({"LOCAL_WORLD_SIZE": "junk"}, 2)[0] == "8"
    assert mask_entries({"HIP_VISIBLE_DEVICES": "0, 2,"}) == 2 and mask_entries({}) is None


def test_bench_self_start_decision(monkeypatch):
    """bench.py starts ranks itself only for --gpus N > 1 outside a launcher (RANK unset); everything else falls through to the normal path"""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "bench.py"))
    b = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(b)
    monkeypatch.delenv("RANK", raising=False)
    assert b.self_start([]) is None and b.self_start(["--gpus", "1", "--steps", "3"]) is None and b.self_start(["--gpus=1"]) is None
    monkeypatch.setenv("RANK", "0")
    assert b.self_start(["--gpus", "8"]) is None                        # already under torch.distributed.run
    started = {}

    class FakeChild:
        stdout = iter(["RCCL banner\n", '{"metric": "m", "value": 1.0}\n'])

        def wait(self):
            return 0
    monkeypatch.delenv("RANK", raising=False)
    import subprocess
    monkeypatch.setattr(subprocess, "Popen", lambda cmd, **kw: started.setdefault("cmd", cmd) and FakeChild())
    assert b.self_start(["--gpus", "4", "--steps", "2"]) == 0
    cmd = started["cmd"]
    assert cmd[1:4] == ["-m", "torch.distributed.run", "--nnodes=1"] and "--nproc-per-node=4" in cmd and "127.0.0.1" in cmd
    assert cmd[-4:] == ["--gpus", "4", "--steps", "2"] and cmd[-5].endswith("bench.py")


def test_optimizer_frozen_names_are_exact_and_an_aborted_bucket_step_ends_the_run():
    """FusedAdamW host logic without a device (round-3 advisor findings): (1) a parameter frozen after the optimizer was built is matched by its
    exact name -- freezing `head.weight` must not freeze `head.weight_g`, while set_unused() prefixes still cover everything below them;
    (2) abort_step() after some ranges of a per-bucket step were already updated leaves parameters one step apart: every further step is refused."""
    from types import SimpleNamespace
    from unite_amd.optim_factory import FusedAdamW
    names = ["head.weight", "head.weight_g", "clip_decoder.0.w", "blocks.0.w"]
    params = [torch.nn.Parameter(torch.zeros(4)) for _ in names]
    fake = SimpleNamespace(names=names, params=params, unused_prefixes=(), total=4096, chunk_groups=lambda group_of: dict(group_of))
    opt = FusedAdamW([{"params": params}], fake, lr=1e-3)
    params[0].requires_grad_(False)                              # frozen behind create_optimizer (run_stage2.py:711-746)
    opt.set_unused(("clip_decoder.",))
    table = opt._build_chunk_table()
    assert table == {"head.weight": 1, "head.weight_g": 0, "clip_decoder.0.w": 1, "blocks.0.w": 0} and opt._frozen_group == 1
    params[0].requires_grad_(True)                               # --lp_ft_epochs switches it back on
    assert opt._effective_unused() != opt._table_unused
    assert opt._build_chunk_table() == {"head.weight": 0, "head.weight_g": 0, "clip_decoder.0.w": 1, "blocks.0.w": 0}
    # an aborted step with nothing updated yet is harmless ...
    opt._step, opt._open = 3, dict(hp=None, done=[])
    opt.abort_step()
    assert opt._step == 2 and opt._open is None
    opt._check_usable()
    # ... one that had already updated a range is the end of the run
    opt._step, opt._open = 3, dict(hp=None, done=[(0, 1024)])
    opt.abort_step()
    with pytest.raises(RuntimeError, match="aborted after 1"):
        opt.begin_step()
    with pytest.raises(RuntimeError, match="restore a checkpoint"):
        opt.step()
