"""Zero-shot CLIP, text side: the class-prompt embeddings of stage 3 (reference src/utils.py:44-53 ``setup_clip``: ``clip.load("ViT-B/16")``,
``clip.tokenize(f"a video of a person {c}")``, ``model.encode_text(...)``; the result feeds ``clip_infer``, src/utils.py:55-68).

The algorithm lives in a third-party dependency that is NOT in /root/reference: OpenAI CLIP, installed by the reference from
``git+https://github.com/openai/CLIP.git`` (environment.yaml:353, unpinned; with ftfy==6.1.1 and regex==2023.10.3, :184,291).  This module restates its
published algorithm in this build's own structure:

* ``BpeTokenizer`` -- the lower-cased byte-level BPE of ``clip/simple_tokenizer.py``: UTF-8 bytes mapped to printable code points, a word's last symbol
  marked ``</w>``, merges applied lowest rank first; vocabulary = 256 byte symbols, their ``</w>`` forms, the merges in file order,
  ``<|startoftext|>``, ``<|endoftext|>``.  ``tokenize`` frames each text as [sot] ids [eot], zero-padded to the context length.
* ``TextTower`` -- ``CLIP.encode_text`` of ``clip/model.py``: token + position embeddings, pre-LN residual blocks (multi-head attention under a causal
  mask, c_fc -> QuickGELU -> c_proj), ``ln_final``, the row at each text's end-of-text token (the largest id) times ``text_projection``.

It runs ONCE per job on nb_classes <= 23 prompts of 77 tokens (setup, not the hot path), in fp32 torch on whatever device it is given; the per-step
zero-shot work (image tower + similarity) is on the HIP kernels (unite_amd.clip, ``unite_clip_similarity``).

PARITY UNPINNED: neither the BPE vocabulary (``bpe_simple_vocab_16e6.txt.gz``, a data file of that package) nor CLIP weights exist offline, and the
reference holds no vector for this path.  tests/test_clip_text.py checks the tokenizer's mechanics on a synthetic merge table and the tower against
an independent loop restatement (oracle/clip_text_oracle.py) on seeded weights."""
import gzip
import html
import os
from typing import Dict, Iterable, List, Sequence, Tuple, Union

import torch

CONTEXT_LENGTH = 77
PROMPT = "a video of a person {}"          # src/utils.py:48


_PRINTABLE = list(range(ord("!"), ord("~") + 1)) + list(range(0xA1, 0xAC + 1)) + list(range(0xAE, 0xFF + 1))


def _byte_symbols() -> Dict[int, str]:
    """byte -> printable code point: the printable Latin-1 ranges stand for themselves, every other byte gets 256, 257, ... in byte order.
    The dict is in VOCABULARY order: the self-mapped bytes first (range by range), then the remapped ones."""
    table = {b: chr(b) for b in _PRINTABLE}
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + len(table) - len(_PRINTABLE))
    return table


def _read_merges(path: str, limit: int) -> List[Tuple[str, str]]:
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rt", encoding="utf-8") as f:
        lines = f.read().split("\n")
    out = []
    for line in lines[1:1 + limit]:          # the first line is a version header
        parts = line.split()
        if len(parts) == 2:
            out.append((parts[0], parts[1]))
    return out


class BpeTokenizer:
    SOT, EOT = "<|startoftext|>", "<|endoftext|>"

    def __init__(self, vocab_path: str, n_merges: int = 49152 - 256 - 2):
        if not vocab_path or not os.path.exists(vocab_path):
            raise FileNotFoundError(f"BPE merge table not found: {vocab_path!r} (OpenAI CLIP's bpe_simple_vocab_16e6.txt.gz; pass --clip_bpe_vocab)")
        import regex
        self._bytes = _byte_symbols()
        merges = _read_merges(vocab_path, n_merges)
        symbols = list(self._bytes.values())
        vocab = symbols + [s + "</w>" for s in symbols] + [a + b for a, b in merges] + [self.SOT, self.EOT]
        self.ids = {tok: i for i, tok in enumerate(vocab)}
        self.text_of = {i: tok for tok, i in self.ids.items()}
        self.rank = {pair: i for i, pair in enumerate(merges)}
        self._memo = {self.SOT: self.SOT, self.EOT: self.EOT}
        self._split = regex.compile(r"<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+", regex.IGNORECASE)

    @property
    def sot_id(self) -> int:
        return self.ids[self.SOT]

    @property
    def eot_id(self) -> int:
        return self.ids[self.EOT]

    def _merge_word(self, word: str) -> str:
        """one pre-token (already in byte symbols) -> its BPE symbols joined by spaces"""
        hit = self._memo.get(word)
        if hit is not None:
            return hit
        parts = list(word[:-1]) + [word[-1] + "</w>"]
        while len(parts) > 1:
            best = min(zip(parts, parts[1:]), key=lambda p: self.rank.get(p, float("inf")))
            if best not in self.rank:
                break
            a, b = best
            merged, i = [], 0
            while i < len(parts):
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        out = " ".join(parts)
        self._memo[word] = out
        return out

    @staticmethod
    def clean(text: str) -> str:
        """html entities undone twice, whitespace runs collapsed, lower case.  (The package also runs ftfy.fix_text first, which repairs mojibake;
        ftfy is not installed here and the class names of src/utils.py:70-82 are plain ASCII.)"""
        text = html.unescape(html.unescape(text)).strip()
        return " ".join(text.split()).lower()

    def encode(self, text: str) -> List[int]:
        out = []
        for piece in self._split.findall(self.clean(text)):
            word = "".join(self._bytes[b] for b in piece.encode("utf-8"))
            out.extend(self.ids[s] for s in self._merge_word(word).split(" "))
        return out

    def decode(self, ids: Iterable[int]) -> str:
        back = {s: b for b, s in self._bytes.items()}
        raw = bytearray()
        for i in ids:
            tok = self.text_of[int(i)]
            if tok in (self.SOT, self.EOT):
                raw += tok.encode() + b" "
                continue
            end = tok.endswith("</w>")
            raw += bytes(back[c] for c in (tok[:-4] if end else tok))
            if end:
                raw += b" "
        return raw.decode("utf-8", errors="replace")

    def tokenize(self, texts: Union[str, Sequence[str]], context_length: int = CONTEXT_LENGTH, truncate: bool = False) -> torch.Tensor:
        """(n, context_length) int64: [sot] ids [eot] per text, zero-padded; a text that does not fit raises (or, with ``truncate``, is cut and
        ends in eot) -- ``clip.tokenize``"""
        if isinstance(texts, str):
            texts = [texts]
        out = torch.zeros(len(texts), context_length, dtype=torch.long)
        for i, t in enumerate(texts):
            ids = [self.sot_id] + self.encode(t) + [self.eot_id]
            if len(ids) > context_length:
                if not truncate:
                    raise RuntimeError(f"Input {t} is too long for context length {context_length}")
                ids = ids[:context_length]
                ids[-1] = self.eot_id
            out[i, :len(ids)] = torch.tensor(ids)
        return out


class TextTower:
    """CLIP's text encoder over a state dict with OpenAI's key names (``token_embedding.weight``, ``positional_embedding``,
    ``transformer.resblocks.<i>.{ln_1,attn.in_proj_weight,attn.in_proj_bias,attn.out_proj,ln_2,mlp.c_fc,mlp.c_proj}``, ``ln_final``,
    ``text_projection``); keys of the image side are ignored."""

    def __init__(self, state_dict: Dict[str, torch.Tensor], device="cpu"):
        need = ("token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias", "text_projection")
        missing = [k for k in need if k not in state_dict]
        if missing:
            raise KeyError(f"not a CLIP text-side state dict: missing {missing}")
        self.w = {k: v.detach().to(device=device, dtype=torch.float32) for k, v in state_dict.items()
                  if k in need or k.startswith("transformer.resblocks.")}
        self.width = self.w["token_embedding.weight"].shape[1]
        self.context = self.w["positional_embedding"].shape[0]
        self.layers = 1 + max(int(k.split(".")[2]) for k in self.w if k.startswith("transformer.resblocks."))
        self.heads = self.width // 64          # clip/model.py: transformer_heads = transformer_width // 64
        self.output_dim = self.w["text_projection"].shape[1]
        self.device = device

    def _block(self, x: torch.Tensor, i: int, mask: torch.Tensor) -> torch.Tensor:
        w, p = self.w, f"transformer.resblocks.{i}."
        n, L, D = x.shape
        H, dh = self.heads, D // self.heads
        h = torch.nn.functional.layer_norm(x, (D,), w[p + "ln_1.weight"], w[p + "ln_1.bias"], 1e-5)
        qkv = h @ w[p + "attn.in_proj_weight"].t() + w[p + "attn.in_proj_bias"]
        q, k, v = (t.reshape(n, L, H, dh).transpose(1, 2) for t in qkv.split(D, dim=-1))
        att = torch.softmax(q @ k.transpose(-1, -2) * dh ** -0.5 + mask, dim=-1)
        o = (att @ v).transpose(1, 2).reshape(n, L, D)
        x = x + o @ w[p + "attn.out_proj.weight"].t() + w[p + "attn.out_proj.bias"]
        h = torch.nn.functional.layer_norm(x, (D,), w[p + "ln_2.weight"], w[p + "ln_2.bias"], 1e-5)
        a = h @ w[p + "mlp.c_fc.weight"].t() + w[p + "mlp.c_fc.bias"]
        a = a * torch.sigmoid(1.702 * a)
        return x + a @ w[p + "mlp.c_proj.weight"].t() + w[p + "mlp.c_proj.bias"]

    @torch.no_grad()
    def encode_text(self, tokens: torch.Tensor) -> torch.Tensor:
        """(n, context) token ids -> (n, output_dim) embeddings (not normalised: clip_infer normalises, src/utils.py:62)"""
        tokens = tokens.to(self.device)
        n, L = tokens.shape
        if L != self.context:
            raise ValueError(f"context length {L} != the tower's {self.context}")
        x = self.w["token_embedding.weight"][tokens] + self.w["positional_embedding"]
        mask = torch.full((L, L), float("-inf"), device=x.device).triu_(1)          # a token sees itself and the tokens before it
        for i in range(self.layers):
            x = self._block(x, i, mask)
        x = torch.nn.functional.layer_norm(x, (self.width,), self.w["ln_final.weight"], self.w["ln_final.bias"], 1e-5)
        eot = tokens.argmax(dim=-1)                                                  # end-of-text has the largest id
        return x[torch.arange(n, device=x.device), eot] @ self.w["text_projection"]


def class_text_features(class_names: Sequence[str], tokenizer: BpeTokenizer, tower: TextTower, prompt: str = PROMPT) -> torch.Tensor:
    """(n_classes, C) float32: ``model.encode_text(cat([tokenize(f"a video of a person {c}") ...])).float()`` -- src/utils.py:47-51"""
    tokens = tokenizer.tokenize([prompt.format(c) for c in class_names], context_length=tower.context)
    return tower.encode_text(tokens).float()


def load_text_tower(weights_path: str, device="cpu") -> TextTower:
    """a state-dict file (``torch.save(model.state_dict())`` of an OpenAI CLIP, or just its text-side keys), read with the tensor-only loader"""
    sd = torch.load(weights_path, map_location="cpu", weights_only=True)
    if isinstance(sd, dict) and "state_dict" in sd and isinstance(sd["state_dict"], dict):
        sd = sd["state_dict"]
    return TextTower(sd, device)
