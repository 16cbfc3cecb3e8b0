"""unite_amd/cli.py against the flag tables read from the reference's own get_args() sources (oracle/make_golden_cli.py ->
tests/golden/cli_flags.json), and the precedence rules of run_stage1.py:231-247: command line > --config YAML > default, then the
--dataset mapping over everything.  CPU only.  (The reference's parsers cannot be executed here -- the driver scripts import wandb /
decord -- so the behavioural half of this file is pinned on argparse's documented semantics of parse_args(namespace=...), which is what
the reference calls; parity of the tables is exact.)"""
import json
import os

import pytest
import yaml

from unite_amd import cli


def _golden(golden_dir):
    return json.load(open(os.path.join(golden_dir, "cli_flags.json")))


def _norm_kw(kw):
    out = {}
    for k, v in kw.items():
        if k == "type":
            v = getattr(v, "__name__", str(v))
        out[k] = list(v) if isinstance(v, tuple) else v
    return out


@pytest.mark.parametrize("stage", ["stage1", "stage2", "stage3"])
def test_flag_tables_equal_the_reference(golden_dir, stage):
    ref = _golden(golden_dir)[stage]
    mine = {opts[0]: (list(opts), _norm_kw(kw)) for opts, kw in cli.flag_table(stage)}
    want = {}
    for f in ref["flags"]:
        kw = {k: v for k, v in f.items() if k not in ("opts", "line")}
        if "type" in kw:
            kw["type"] = kw["type"].split(".")[-1]                      # utils.str2bool -> str2bool
        want[f["opts"][0]] = (f["opts"], kw)
    assert sorted(mine) == sorted(want)
    for name in want:
        assert mine[name] == want[name], name
    assert cli.SET_DEFAULTS[stage] == ref["set_defaults"]


@pytest.mark.parametrize("stage", ["stage1", "stage2", "stage3"])
def test_defaults_with_no_flags(golden_dir, stage):
    """every dest of the reference parser exists with the reference's default (set_defaults wins over add_argument's default)"""
    ref = _golden(golden_dir)[stage]
    a = vars(cli.get_args(stage, []))
    expect = {}
    for f in ref["flags"]:
        dest = f.get("dest") or f["opts"][-1].lstrip("-").replace("-", "_")
        act = f.get("action")
        if "default" in f:
            d = f["default"]
        elif act == "store_true":
            d = False
        elif act == "store_false":
            d = True
        else:
            d = None
        if dest not in expect or "default" in f:
            expect.setdefault(dest, d)
    expect.update(ref["set_defaults"])
    for k, v in expect.items():
        assert k in a, k
        if isinstance(v, dict):              # an expression in the source (none today)
            continue
        assert a[k] == v, (k, a[k], v)


def test_precedence_cli_over_yaml_over_default(tmp_path):
    cfg = tmp_path / "c.yaml"
    cfg.write_text(yaml.safe_dump({"batch_size": 7, "mask_ratio": 0.8, "clip_return_layers": [6], "opt_betas": [0.9, 0.95], "dataset": ""}))
    a = cli.get_args("stage1", ["--config", str(cfg)])
    assert a.batch_size == 7 and a.mask_ratio == 0.8 and a.clip_return_layers == [6] and a.opt_betas == [0.9, 0.95]
    assert a.epochs == 800 and a.lr == 1.5e-4                                 # untouched defaults
    b = cli.get_args("stage1", ["--config", str(cfg), "--batch_size", "3", "--epochs", "2"])
    assert b.batch_size == 3 and b.epochs == 2 and b.mask_ratio == 0.8         # explicit flags win, the YAML keeps the rest
    c = cli.get_args("stage3", ["--config", str(cfg), "--checkpoints_disabled", "--no_auto_resume"])
    assert c.checkpoints_enabled is False and c.auto_resume is False and c.batch_size == 7


def test_dataset_mapping_overrides_everything(tmp_path):
    maps = tmp_path / "dataset_mappings.yaml"
    maps.write_text(yaml.safe_dump({"toy_a2b": {"nb_classes": 12, "ann_file_train": "/d/a.txt", "ann_file_train_target": "/d/b.txt", "batch_size": 5}}))
    a = cli.get_args("stage1", ["--dataset", "toy_a2b", "--batch_size", "9", "--nb_classes", "3"], mappings_path=str(maps))
    assert a.nb_classes == 12 and a.batch_size == 5 and a.ann_file_train_target == "/d/b.txt"      # run_stage1.py:259-261: setattr over the parse
    with pytest.raises(KeyError):
        cli.get_args("stage1", ["--dataset", "missing"], mappings_path=str(maps))
    with pytest.raises(FileNotFoundError):
        cli.get_args("stage1", ["--dataset", "toy_a2b"], mappings_path=str(tmp_path / "nope.yaml"))


def test_reference_configs_parse(tmp_path):
    """the three YAML files shaped like configs/stage{1,2,3}_config.yaml load through --config without unknown-key errors
    (a YAML key that is no flag simply becomes an attribute, as in the reference)."""
    cfg = tmp_path / "s2.yaml"
    cfg.write_text(yaml.safe_dump({"model": "vit_base_patch16_224", "num_frames": 16, "tubelet_size": 1, "layer_decay": 0.65,
                                   "update_freq": 2, "some_future_key": 1}))
    a = cli.get_args("stage2", ["--config", str(cfg), "--eval"])
    assert a.model == "vit_base_patch16_224" and a.update_freq == 2 and a.some_future_key == 1 and a.eval is True
