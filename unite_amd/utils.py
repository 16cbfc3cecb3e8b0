"""Host-side helpers the engines call -- the hot subset of reference src/utils.py with the same names and call
signatures: loss scaler facade, gradient norm, schedules, meters, distributed
init, checkpoint dict layout.  Device work goes through unite_amd.ops (C ABI); nothing here computes on ATen
in the per-step path except reading scalars back at log time.
"""
from __future__ import annotations

import datetime
import math
import statistics
import os
import time
from collections import defaultdict, deque
from pathlib import Path
from typing import Optional

import numpy as np
import torch
import torch.distributed as dist

from . import ops


# ----------------------------------------------------------------------------- scaler / grad norm (utils.py:602-643)
class NativeScalerWithGradNormCount:
    """Same call signature and state_dict()['scale'] as the reference (utils.py:602-628).  The kernels compute in
    bf16 operands / fp32 accumulation, so there is no loss scaling: scale is the constant 1.0 (SURVEY A-17)."""
    state_dict_key = "amp_scaler"
    RING = 256

    def __init__(self):
        self._scale = 1.0
        self._norm = None
        self._coef = None
        self._ws = None
        # AdamW of each gradient bucket right behind its all-reduce instead of one launch after the whole backward (opt-in; only without
        # gradient clipping: the clip coefficient needs every gradient first).  Same parameters bit for bit (tests/test_ddp_gpu.py).
        self.bucket_adamw = os.environ.get("UNITE_BUCKET_ADAMW", "0") == "1"

    def __call__(self, loss, optimizer, clip_grad=None, parameters=None, create_graph=False, update_grad=True, reducer=None):
        if reducer is not None and not update_grad:
            # a micro-batch of an accumulation step (update_freq > 1): gradients only add up in the flat buffer; the reducer
            # stays quiet until the backward of the last micro-batch, which reduces the accumulated sum once
            with reducer.no_sync():
                loss.backward(create_graph=create_graph)
            reducer.reset()
            return None
        per_bucket = (self.bucket_adamw and reducer is not None and update_grad and not (clip_grad is not None and clip_grad > 0)
                      and hasattr(optimizer, "begin_step") and getattr(optimizer, "_step_params", None) is None)
        if per_bucket:
            optimizer.begin_step()
            reducer.after_bucket = optimizer.step_range
        try:
            loss.backward(create_graph=create_graph)
            if not update_grad:
                return None
            if reducer is not None:
                reducer.finish()                      # join the gradient all-reduce (side stream) before reading grads
        except BaseException:
            if per_bucket:
                optimizer.abort_step()                # a backward that raised leaves no half-open optimizer step behind
            raise
        finally:
            if per_bucket:
                reducer.after_bucket = None
        flat = getattr(optimizer, "_flat", None)
        if flat is None:
            raise RuntimeError("optimizer is not a unite_amd FusedAdamW bound to a flat parameter buffer")
        norm = self.grad_norm(flat, clip_grad, optimizer)
        optimizer.step(grad_scale=self._coef if (clip_grad is not None and clip_grad > 0) else None)
        return norm

    def grad_norm(self, flat, clip_grad=None, optimizer=None):
        dev = flat.grad.device
        if self._norm is None:
            # a ring of result slots: the engines keep the returned 0-dim tensors un-read until the next log line (no host sync per
            # step), so consecutive steps must not write the same element
            self._norm = torch.zeros(self.RING, device=dev)
            self._slot = 0
            self._coef = torch.ones(1, device=dev)
            self._ws = torch.empty(ops.grad_norm_workspace(flat.grad.numel()), dtype=torch.uint8, device=dev)
        self._slot = (self._slot + 1) % self.RING
        out = self._norm[self._slot:self._slot + 1]
        # parameters that have no gradient in this step (frozen, or in layers that were not run) are in a group of their own that the
        # optimizer skips: they stay out of the norm as well (the reference's norm runs over `p.grad is not None`)
        table, skip = optimizer.no_grad_chunks() if hasattr(optimizer, "no_grad_chunks") else (None, -1)
        ops.grad_norm_flat(flat.grad, out, self._ws, max_norm=float(clip_grad or 0.0), clip_coef_out=self._coef, chunk_group=table, skip_group=skip)
        return out[0]                                 # 0-dim device tensor: no host sync here

    def state_dict(self):
        return {"scale": self._scale}

    def load_state_dict(self, state_dict):
        self._scale = 1.0


def get_grad_norm_(parameters, norm_type: float = 2.0) -> torch.Tensor:
    """utils.py:631-643 for models on a flat buffer: sqrt(sum of squares) in one reduction."""
    if isinstance(parameters, torch.Tensor):
        parameters = [parameters]
    parameters = [p for p in parameters if p.grad is not None]
    if len(parameters) == 0:
        return torch.tensor(0.)
    if norm_type != 2.0:
        raise NotImplementedError("only the L2 norm is built")
    dev = parameters[0].grad.device
    out = torch.zeros(1, device=dev)
    sq = torch.zeros(1, device=dev)
    ws = None
    total = None
    for p in parameters:       # generic (non-flat) path: one reduction per tensor, combined on device
        g = p.grad.contiguous().view(-1)
        if ws is None or ws.numel() < ops.grad_norm_workspace(g.numel()):
            ws = torch.empty(ops.grad_norm_workspace(g.numel()), dtype=torch.uint8, device=dev)
        ops.grad_norm_flat(g, out, ws)
        total = out * out if total is None else total + out * out
    return total.sqrt()[0]


# ----------------------------------------------------------------------------- zero-shot CLIP similarities (utils.py:55-68)
def clip_infer(model, videos, text_features):
    """reference utils.clip_infer: per-frame image embeddings (here: unite_amd.clip.VisionTransformer.encode_image, i.e. the frozen
    CLIP image tower on the HIP kernels), cosine similarity x 100 against the class text embeddings, soft-max, mean over the frames.
    ``text_features`` (n_classes, C): what setup_clip returned (unite_amd.clip_text, or a file of embeddings)."""
    B, T = videos.shape[0], videos.shape[2]
    img = model.encode_image(videos)
    text = text_features.float()
    text = (text / text.norm(dim=-1, keepdim=True)).contiguous()           # (the reference normalises the caller's tensor in place)
    out = torch.empty(B, text.shape[0], dtype=torch.float32, device=videos.device)
    return ops.clip_similarity(img, text, out, T, 100.0)


# ----------------------------------------------------------------------------- schedules (utils.py:646-686)
def _warmup_ramp(base_value, niter_per_ep, warmup_epochs, start_warmup_value, warmup_steps):
    """the linear ramp both schedules start with, and the number of iterations it stands for.  As in the reference the ramp exists only when
    warmup_epochs > 0, while its LENGTH is warmup_steps whenever that is positive (so warmup_steps without warmup_epochs leaves a gap that the
    length check of the caller rejects, utils.py:649-654)."""
    n = warmup_steps if warmup_steps > 0 else warmup_epochs * niter_per_ep
    ramp = np.linspace(start_warmup_value, base_value, n) if warmup_epochs > 0 else np.empty(0)
    return ramp, n


def cosine_scheduler(base_value, final_value, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1):
    """per-iteration values: linear warm-up to base_value, then half a cosine period down to final_value (reference utils.py:646-663; pinned by
    tests/golden/utils.npz)."""
    total = epochs * niter_per_ep
    ramp, n_warm = _warmup_ramp(base_value, niter_per_ep, warmup_epochs, start_warmup_value, warmup_steps)
    print(f"Set warmup steps = {n_warm:d}")
    n_cos = total - n_warm
    phase = np.arange(n_cos, dtype=np.float64) / max(n_cos, 1)                      # 0 .. (n-1)/n: the last value stays above final_value
    values = np.concatenate([ramp, final_value + 0.5 * (base_value - final_value) * (1.0 + np.cos(np.pi * phase))])
    assert len(values) == total, (len(values), total)
    return values


def step_scheduler(base_value, step_fraction, epochs, niter_per_ep, warmup_epochs=0, start_warmup_value=0, warmup_steps=-1, steps=None):
    """constant base_value after the warm-up, or -- with `steps` (epoch numbers) -- a staircase of cumulative factors step_fraction[i] from epoch
    steps[i] on, over the whole run and relative to 1.0 as the reference builds it (utils.py:666-686)."""
    total = epochs * niter_per_ep
    ramp, n_warm = _warmup_ramp(base_value, niter_per_ep, warmup_epochs, start_warmup_value, warmup_steps)
    if steps is None:
        body = np.full(total - n_warm, float(base_value))
    else:
        epoch_of = np.arange(total) // niter_per_ep
        body = np.ones(total)
        for first_epoch, factor in zip(steps, step_fraction):
            body = np.where(epoch_of >= first_epoch, body * factor, body)
    values = np.concatenate([ramp, body])
    assert len(values) == total, (len(values), total)
    return values


def get_greedy_masks(attn, mask_ratio, k):
    """utils.py:89-120: committee member i unmasks attention ranks i, i+k, ... ; bool (k, BT, N), True = masked.
    (stage 3; index bookkeeping on the device through torch's sort -- not on the stage-1 hot path.)"""
    BT, N = attn.shape
    n_unmask = N - int(N * mask_ratio)
    _, order = attn.sort(dim=1, descending=True)
    masks = torch.ones((k, BT, N), dtype=torch.bool, device=attn.device)
    for i in range(k):
        masks[i].scatter_(1, order[:, i::k][:, :n_unmask], False)
    return masks


# ----------------------------------------------------------------------------- meters (utils.py:215-423)
class SmoothedValue(object):
    """a series seen through a window of the last `window_size` values plus its running mean over everything (reference utils.py:215-277: same
    properties and format fields; the window statistics are plain Python here -- nothing of this runs on the device)."""

    def __init__(self, window_size=20, fmt=None):
        self.fmt = "{median:.4f} ({global_avg:.4f})" if fmt is None else fmt
        self.deque = deque(maxlen=window_size)          # (attribute name kept: callers of the reference read it)
        self.count = 0
        self.total = 0.0

    def update(self, value, n=1):
        self.deque.append(value)
        self.total += n * value
        self.count += n

    def synchronize_between_processes(self):
        """count and total summed over the ranks; the window stays local (as in the reference)"""
        if not is_dist_avail_and_initialized():
            return
        pair = torch.tensor([self.count, self.total], dtype=torch.float64, device='cuda' if dist.get_backend() == 'nccl' else 'cpu')
        dist.barrier()
        dist.all_reduce(pair)
        self.count, self.total = int(pair[0].item()), float(pair[1].item())

    def _window(self):
        return [float(v) for v in self.deque]

    median = property(lambda self: statistics.median_low(self._window()))          # the lower middle value, as torch.median picks it
    avg = property(lambda self: math.fsum(self._window()) / len(self.deque))
    global_avg = property(lambda self: self.total / max(self.count, 1))
    max = property(lambda self: max(self.deque))
    value = property(lambda self: self.deque[-1])

    def __str__(self):
        fields = {name: getattr(self, name) for name in ("median", "avg", "global_avg", "max", "value") if "{" + name in self.fmt}
        return self.fmt.format(**fields)


class MetricLogger(object):
    """named SmoothedValue meters, created on first update, reachable as attributes (reference utils.py:280-360)"""

    def __init__(self, delimiter="\t"):
        self.delimiter = delimiter
        self.meters = defaultdict(SmoothedValue)

    def update(self, **kwargs):
        for name, v in kwargs.items():
            if v is not None:                      # (the reference skips None the same way: a loss that was not computed this step)
                self.meters[name].update(float(v.item() if isinstance(v, torch.Tensor) else v))

    def __getattr__(self, attr):
        meters = self.__dict__.get("meters", {})
        if attr not in meters:
            raise AttributeError(attr)
        return meters[attr]

    def __str__(self):
        return self.delimiter.join(f"{name}: {meter}" for name, meter in self.meters.items())

    def synchronize_between_processes(self):
        for meter in self.meters.values():
            meter.synchronize_between_processes()

    def add_meter(self, name, meter):
        self.meters[name] = meter

    def log_every(self, iterable, print_freq, n_epochs=None, epoch=None, ipe=None, header=None):
        i = 0
        header = header or ''
        start = time.time()
        end = time.time()
        iter_time, data_time = SmoothedValue(fmt='{avg:.4f}'), SmoothedValue(fmt='{avg:.4f}')
        n = len(iterable)
        for obj in iterable:
            data_time.update(time.time() - end)
            yield obj
            iter_time.update(time.time() - end)
            if print_freq and (i % print_freq == 0 or i == n - 1):
                eta = str(datetime.timedelta(seconds=int(iter_time.global_avg * (n - i))))
                print(self.delimiter.join([header, f"[{i}/{n}]", f"eta: {eta}", str(self), f"time: {iter_time}", f"data: {data_time}"]))
            i += 1
            end = time.time()
        total = time.time() - start
        print('{} Total time: {} ({:.4f} s / it)'.format(header, str(datetime.timedelta(seconds=int(total))), total / max(n, 1)))


# ----------------------------------------------------------------------------- distributed (utils.py:450-551)
def is_dist_avail_and_initialized():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_dist_avail_and_initialized() else 1


def get_rank():
    return dist.get_rank() if is_dist_avail_and_initialized() else 0


def is_main_process():
    return get_rank() == 0


def save_on_master(obj, path, **kwargs):
    """rank 0 writes `obj` to `path` -- into a temporary file beside it first, renamed over the target once complete, so that a reader on
    another rank (or a resumed job) never sees a half-written checkpoint (the reference writes in place and sleeps before reading back,
    run_stage2.py:825-829)"""
    if not is_main_process():
        return
    if isinstance(path, (str, os.PathLike)):
        tmp = f"{os.fspath(path)}.tmp{os.getpid()}"
        try:
            torch.save(obj, tmp, **kwargs)
            os.replace(tmp, path)
        finally:
            if os.path.exists(tmp):
                os.remove(tmp)
    else:                                           # a file object: the caller owns it
        torch.save(obj, path, **kwargs)


def init_distributed_mode(args):
    """env:// rendezvous as torchrun provides it (utils.py:532-551); backend 'nccl' is RCCL on ROCm."""
    if 'RANK' in os.environ and 'WORLD_SIZE' in os.environ:
        args.rank = int(os.environ["RANK"])
        args.world_size = int(os.environ['WORLD_SIZE'])
        args.gpu = int(os.environ['LOCAL_RANK'])
    else:
        print('Not using distributed mode')
        args.distributed = False
        return
    args.distributed = True
    backend = getattr(args, "dist_backend", "nccl")
    if backend == "nccl":
        torch.cuda.set_device(args.gpu)
    print('| distributed init (rank {}): {}, gpu {}'.format(args.rank, getattr(args, "dist_url", "env://"), args.gpu), flush=True)
    dist.init_process_group(backend=backend, init_method=getattr(args, "dist_url", "env://"), world_size=args.world_size, rank=args.rank)
    if backend == "nccl":
        dist.barrier(device_ids=[args.gpu])
        if os.environ.get("UNITE_COMM_NATIVE", "0") == "1":
            # libunite_comm.so's communicator at the point where torch creates its own (bench.py has the note on what this does and does not fix)
            from .ddp import native_comm
            native_comm()
    else:
        dist.barrier()


# ----------------------------------------------------------------------------- checkpoints (utils.py:689-776)
def load_state_dict(model, state_dict, prefix='', ignore_missing="relative_position_index"):
    """non-strict load with a key prefix that REPORTS instead of raising (reference utils.py:554-599): every module takes what it finds under its
    own prefix through nn.Module._load_from_state_dict; afterwards the missing keys are split into ignorable ones (a '|'-separated list of
    substrings) and the rest, and the unused keys are listed.  Returns (missing keys that matter, unused keys)."""
    src = state_dict.copy()
    meta = getattr(state_dict, '_metadata', None)
    if meta is not None:
        src._metadata = meta
    missing, unused, errors = [], [], []
    for name, module in model.named_modules():                 # parents before children, the order a recursive walk visits them in
        where = prefix + (name + '.' if name else '')
        module._load_from_state_dict(src, where, {} if meta is None else meta.get(where[:-1], {}), True, missing, unused, errors)
    patterns = ignore_missing.split('|')
    ignored = [k for k in missing if any(pat in k for pat in patterns)]
    wanted = [k for k in missing if k not in ignored]
    cls = model.__class__.__name__
    for keys, text in ((wanted, "Weights of {} not initialized from pretrained model: {}"),
                       (unused, "Weights from pretrained model not used in {}: {}"),
                       (ignored, "Ignored weights of {} not initialized from pretrained model: {}")):
        if keys:
            print(text.format(cls, keys))
    if errors:
        print('\n'.join(errors))
    return wanted, unused        # (the bf16 shadow follows by itself: FlatParams watches tensor versions)


def _plain(obj):
    """numpy scalars / arrays -> Python numbers / lists (the lr schedule is a numpy array, so param_groups and args pick up
    numpy.float64): what is saved must load with the tensor-only loader, which executes nothing from the file"""
    if isinstance(obj, np.generic):
        return obj.item()
    if isinstance(obj, np.ndarray):
        return obj.tolist()
    if isinstance(obj, dict):
        return {k: _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return type(obj)(_plain(v) for v in obj)
    return obj


def save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None, tag=None):
    """{'model','optimizer','epoch','scaler','args'} in checkpoint-{epoch|tag}.pth on rank 0 (utils.py:689-736)."""
    output_dir = Path(args.output_dir)
    path = output_dir / ('checkpoint-%s.pth' % (tag if tag is not None else str(epoch)))
    to_save = {'model': model_without_ddp.state_dict(), 'optimizer': _plain(optimizer.state_dict()), 'epoch': epoch,
               'scaler': loss_scaler.state_dict(), 'args': _plain(dict(vars(args))) if hasattr(args, "__dict__") else args}
    save_on_master(to_save, path)
    return path


def save_latest_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema=None, model_name='latest'):
    return save_model(args, epoch, model, model_without_ddp, optimizer, loss_scaler, model_ema, tag=model_name)


def auto_load_model(args, model, model_without_ddp, optimizer, loss_scaler, model_ema=None):
    """prefers checkpoint-latest.pth, then -best, then the highest epoch number (utils.py:739-776)."""
    output_dir = Path(args.output_dir)
    if getattr(args, "auto_resume", False) and len(getattr(args, "resume", "")) == 0:
        import glob
        if (output_dir / 'checkpoint-latest.pth').exists():
            args.resume = str(output_dir / 'checkpoint-latest.pth')
        elif (output_dir / 'checkpoint-best.pth').exists():
            args.resume = str(output_dir / 'checkpoint-best.pth')
        else:
            latest = -1
            for ckpt in glob.glob(os.path.join(output_dir, 'checkpoint-*.pth')):
                t = ckpt.split('-')[-1].split('.')[0]
                if t.isdigit():
                    latest = max(int(t), latest)
            if latest >= 0:
                args.resume = os.path.join(output_dir, 'checkpoint-%d.pth' % latest)
    if getattr(args, "resume", ""):
        from .checkpoint import read_checkpoint
        ckpt = read_checkpoint(args.resume)       # tensor-only loader: save_model stores vars(args), tensors and plain numbers
        model_without_ddp.load_state_dict(ckpt['model'])
        if 'optimizer' in ckpt and 'epoch' in ckpt:
            optimizer.load_state_dict(ckpt['optimizer'])
            args.start_epoch = ckpt['epoch'] + 1
            if 'scaler' in ckpt:
                loss_scaler.load_state_dict(ckpt['scaler'])
        print("Resume checkpoint %s" % args.resume)


# ----------------------------------------------------------------------------- small driver-side helpers (utils.py:70-88,426-480,854-925)
# Not on the hot path; present so that ``from unite_amd import utils`` covers every ``utils.*`` name the reference drivers touch.
def str2bool(v):
    return v if isinstance(v, bool) else v.lower() in ("yes", "true", "t", "1")


_CLASS_NAMES = {       # label sets of the three benchmarks (utils.py:70-82): class prompts of the zero-shot CLIP classifier
    8: ['drink', 'jump', 'pick', 'pour', 'push', 'run', 'walk', 'wave'],
    12: ['climb', 'fencing', 'golf', 'soccer', 'pullup', 'boxing', 'pushup', 'riding bike', 'horse riding', 'basketball', 'archery',
         'walking'],
    23: ['archery', 'baseball', 'basketball', 'biking', 'bowling', 'swimming', 'diving', 'fencing', 'field hockey', 'gymnastics', 'golf',
         'horse riding', 'kayaking', 'rock climbing', 'climbing rope', 'skateboarding', 'skiing', 'sumo wrestling', 'surfing', 'tai chi',
         'tennis', 'trampoline jumping', 'volleyball'],
}


def get_class_names(args):
    if args.nb_classes not in _CLASS_NAMES:
        raise NotImplementedError
    return list(_CLASS_NAMES[args.nb_classes])


def setup_clip(args, device):
    """(image tower, class text features) for the zero-shot side of stage 3 -- src/utils.py:44-53, where ``clip.load("ViT-B/16")`` gives both and
    ``clip.tokenize`` / ``encode_text`` make the features.  Here the image tower is unite_amd.clip.clip_b16 on the HIP kernels
    (``args.clip_teacher_weights`` / UNITE_CLIP_PATH for its weights) and the text side is unite_amd.clip_text (OpenAI CLIP's tokenizer and text
    transformer restated; needs ``args.clip_text_weights``: a state dict with OpenAI's text-side keys, and ``args.clip_bpe_vocab``: that package's
    bpe_simple_vocab_16e6.txt.gz -- neither exists offline).  ``args.clip_text_features`` (a (nb_classes, C) .pt / .npy) short-cuts the text side."""
    from . import clip as _clip
    from . import clip_text
    feats = getattr(args, "clip_text_features", "")
    tw, vocab = getattr(args, "clip_text_weights", ""), getattr(args, "clip_bpe_vocab", "")
    if not feats and not (tw and vocab):
        raise NotImplementedError("utils.setup_clip needs the class text embeddings: pass --clip_text_features, or --clip_text_weights (OpenAI CLIP "
                                  "state dict with its text-side keys) together with --clip_bpe_vocab (bpe_simple_vocab_16e6.txt.gz)")
    weights = getattr(args, "clip_teacher_weights", "") or os.environ.get("UNITE_CLIP_PATH", "")
    model = _clip.clip_b16(pretrained=bool(weights), return_attn=False, clip_return_layers=[11]).to(device)
    if feats:
        text = torch.from_numpy(np.load(feats)) if feats.endswith(".npy") else torch.load(feats, map_location="cpu", weights_only=True)
    else:
        text = clip_text.class_text_features(get_class_names(args), clip_text.BpeTokenizer(vocab), clip_text.load_text_tower(tw, device))
    return model, text.float().to(device)


def create_ds_config(args):
    raise NotImplementedError("DeepSpeed is out of scope (enable_deepspeed: false in every UNITE config)")


class TensorboardLogger(object):
    """scalar logger with the reference's interface (set_step / update(head, step, **scalars) / flush); needs tensorboardX or
    torch.utils.tensorboard at construction time"""

    def __init__(self, log_dir):
        try:
            from tensorboardX import SummaryWriter
            self.writer = SummaryWriter(logdir=log_dir)
        except ImportError:
            from torch.utils.tensorboard import SummaryWriter        # raises ImportError if tensorboard is absent as well
            self.writer = SummaryWriter(log_dir=log_dir)
        self.step = 0

    def set_step(self, step=None):
        self.step = step if step is not None else self.step + 1

    def update(self, head='scalar', step=None, **kwargs):
        for k, v in kwargs.items():
            if v is None:
                continue
            if isinstance(v, torch.Tensor):
                v = v.item()
            assert isinstance(v, (float, int))
            self.writer.add_scalar(head + "/" + k, v, self.step if step is None else step)

    def flush(self):
        self.writer.flush()


def seed_worker(worker_id):
    """DataLoader worker_init_fn: numpy / random seeded from the worker's torch seed"""
    import random
    s = torch.initial_seed() % 2 ** 32
    np.random.seed(s)
    random.seed(s)


def setup_for_distributed(is_master):
    """print() only on the master process (``force=True`` overrides)"""
    import builtins
    plain = builtins.print

    def _print(*args, **kwargs):
        if kwargs.pop('force', False) or is_master:
            plain(*args, **kwargs)

    builtins.print = _print


def _flatten_collate(columns, flat):
    from torch.utils.data.dataloader import default_collate
    return [default_collate([x for sub in col for x in sub] if i in flat else list(col)) for i, col in enumerate(columns)]


def multiple_samples_collate(batch, fold=False):
    """repeated augmentation: every dataset item carries several samples; flatten inputs / labels / indices, keep extra_data per item"""
    inputs, labels, video_idx, extra = _flatten_collate(list(zip(*batch)), flat={0, 1, 2})
    return ([inputs] if fold else inputs), labels, video_idx, extra


def multiple_pretrain_samples_collate(batch, fold=False):
    data, mask = _flatten_collate(list(zip(*batch)), flat={0, 1})
    return ([data] if fold else data), mask


def count_parameters(model):
    return sum(p.numel() for p in model.parameters() if p.requires_grad)


def experiment_exists(dir):
    """a directory (whose path does not contain 'scrap') that already holds a .pth checkpoint"""
    return os.path.isdir(dir) and 'scrap' not in dir and any(f.endswith(".pth") for f in os.listdir(dir))


def confirm_exp_overwrite(output_dir):
    answer = input("Experiment already exists in {}. Overwrite? (y/n): ".format(output_dir)).lower()
    if answer == "y":
        return output_dir
    if answer == "n":
        return str(Path(output_dir).parent / input("Enter new directory name: "))
    raise ValueError("Invalid input. Enter 'y' or 'n'.")
