"""Zero-shot CLIP text side (unite_amd/clip_text.py; reference call site src/utils.py:44-53).  PARITY UNPINNED: OpenAI CLIP is a third-party dependency
of the reference, absent from /root/reference together with its BPE vocabulary and weights; the reference holds no vector for this path.  What can be
checked offline: the tokenizer's mechanics on a synthetic merge table (byte mapping, vocabulary layout, rank-ordered merging against an independent
formulation, framing / padding / truncation), and the text transformer against an independent float64 loop restatement on seeded weights."""
import gzip
import os

import numpy as np
import pytest
import torch

from oracle import clip_text_oracle as TO
from unite_amd import clip_text

MERGES = [("t", "h"), ("th", "e</w>"), ("i", "n"), ("a", "n"), ("r", "u"), ("ru", "n</w>"), ("in", "g</w>"), ("o", "f</w>"), ("p", "e"),
          ("v", "i"), ("vi", "d"), ("vid", "e"), ("vide", "o</w>"), ("Ġ", "x"), ("k", "i"), ("ki", "ng</w>"), ("an", "d</w>"), ("a</w>", "b")]


@pytest.fixture(scope="module")
def tok(tmp_path_factory):
    path = os.path.join(tmp_path_factory.mktemp("bpe"), "merges.txt.gz")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("#version: synthetic\n" + "\n".join(a + " " + b for a, b in MERGES) + "\n")
    return clip_text.BpeTokenizer(path)


def test_byte_symbols_and_vocabulary_layout(tok):
    b = clip_text._byte_symbols()
    assert len(b) == 256 and len(set(b.values())) == 256
    assert b[ord("a")] == "a" and b[ord("!")] == "!" and b[0] == chr(256) and b[32] == chr(0x120) and b[0xAD] == chr(0x143)
    first = list(b.values())
    assert first[0] == "!" and first[93] == "~" and first[94] == "¡" and first[188] == chr(256)      # vocabulary order: printable ranges, then the rest
    n = len(MERGES)
    assert tok.ids["!"] == 0 and tok.ids["!</w>"] == 256 and tok.ids["th"] == 512 and tok.ids["the</w>"] == 513
    assert tok.sot_id == 512 + n and tok.eot_id == 512 + n + 1          # 49406 / 49407 with the package's 48 894 merges
    assert len(tok.ids) == 512 + n + 2


def test_merging_matches_the_sequential_formulation(tok):
    rng = np.random.default_rng(0)
    words = ["the", "then", "running", "run", "of", "video", "king", "kings", "and", "a", "in", "inn", "thethe", "person", "pe"]
    words += ["".join(rng.choice(list("thenrugivdokapf"), size=int(rng.integers(1, 9)))) for _ in range(200)]
    for w in words:
        got = tok._merge_word(w).split(" ")
        want = TO.bpe_sequential(list(w[:-1]) + [w[-1] + "</w>"], MERGES)
        assert got == want, (w, got, want)
    assert tok._merge_word("the") == "the</w>" and tok._merge_word("running") == "ru n n ing</w>"


def test_encode_decode_and_framing(tok):
    ids = tok.encode("A  video of  THE king &amp; running!")
    assert tok.decode(ids) == "a video of the king & running ! "
    assert ids[0] == tok.ids["a</w>"] and ids[1] == tok.ids["video</w>"] and ids[2] == tok.ids["of</w>"] and ids[3] == tok.ids["the</w>"]
    assert tok.encode("it's") == [tok.ids["i"], tok.ids["t</w>"], tok.ids["'"], tok.ids["s</w>"]]          # contractions split off
    assert tok.encode("r2d2") == [tok.ids["r</w>"], tok.ids["2</w>"], tok.ids["d</w>"], tok.ids["2</w>"]]  # digits one by one
    t = tok.tokenize(["a video of a person run", "the"], context_length=12)
    assert t.shape == (2, 12) and t.dtype == torch.long
    assert t[0, 0] == tok.sot_id and t[1, 0] == tok.sot_id and t[1, 1] == tok.ids["the</w>"] and t[1, 2] == tok.eot_id and (t[1, 3:] == 0).all()
    assert int(t[0].argmax()) == int((t[0] == tok.eot_id).nonzero()[0])                                    # eot has the largest id
    long = " ".join(["the"] * 20)
    with pytest.raises(RuntimeError):
        tok.tokenize(long, context_length=12)
    cut = tok.tokenize(long, context_length=12, truncate=True)
    assert cut[0, -1] == tok.eot_id and cut[0, 0] == tok.sot_id
    one = tok.tokenize("the king")          # a bare string is one text; the default context is CLIP's 77
    assert one.shape == (1, 77)
    with pytest.raises(FileNotFoundError):
        clip_text.BpeTokenizer("/nonexistent/bpe.txt.gz")


def _seeded_text_weights(width=128, layers=2, vocab=600, context=12, out=64, seed=0):
    g = torch.Generator().manual_seed(seed)
    r = lambda *s, scale=1.0: torch.randn(*s, generator=g) * scale      # noqa: E731
    sd = {"token_embedding.weight": r(vocab, width, scale=0.5), "positional_embedding": r(context, width, scale=0.1),
          "ln_final.weight": 1 + r(width, scale=0.1), "ln_final.bias": r(width, scale=0.1), "text_projection": r(width, out, scale=width ** -0.5),
          "visual.proj": r(4, 4), "logit_scale": torch.tensor(4.6)}                                        # image-side keys are ignored
    for i in range(layers):
        p = f"transformer.resblocks.{i}."
        sd.update({p + "ln_1.weight": 1 + r(width, scale=0.1), p + "ln_1.bias": r(width, scale=0.1),
                   p + "attn.in_proj_weight": r(3 * width, width, scale=width ** -0.5), p + "attn.in_proj_bias": r(3 * width, scale=0.1),
                   p + "attn.out_proj.weight": r(width, width, scale=width ** -0.5), p + "attn.out_proj.bias": r(width, scale=0.1),
                   p + "ln_2.weight": 1 + r(width, scale=0.1), p + "ln_2.bias": r(width, scale=0.1),
                   p + "mlp.c_fc.weight": r(4 * width, width, scale=width ** -0.5), p + "mlp.c_fc.bias": r(4 * width, scale=0.1),
                   p + "mlp.c_proj.weight": r(width, 4 * width, scale=(4 * width) ** -0.5), p + "mlp.c_proj.bias": r(width, scale=0.1)})
    return sd


def test_text_tower_matches_the_loop_restatement(tok, tmp_path):
    sd = _seeded_text_weights()
    tower = clip_text.TextTower(sd)
    assert (tower.width, tower.layers, tower.heads, tower.context, tower.output_dim) == (128, 2, 2, 12, 64)
    tokens = tok.tokenize(["a video of a person run", "the king", "running and running and"], context_length=12)
    got = tower.encode_text(tokens)
    want = TO.text_forward({k: v.numpy() for k, v in sd.items() if k.startswith(("token", "positional", "ln_final", "text_proj", "transformer."))},
                           tokens.numpy(), heads=2)
    np.testing.assert_allclose(got.numpy(), want, rtol=2e-4, atol=2e-5)
    # causal: what follows a text's end-of-text token (padding) cannot change its embedding
    t2 = tokens.clone()
    t2[1, 5:] = 7
    assert int(t2[1].argmax()) == int(tokens[1].argmax())
    torch.testing.assert_close(tower.encode_text(t2)[1], got[1], rtol=1e-5, atol=1e-6)
    with pytest.raises(ValueError):
        tower.encode_text(tokens[:, :8])
    with pytest.raises(KeyError):
        clip_text.TextTower({"visual.proj": torch.zeros(2, 2)})
    # the file route of setup_clip: tensor-only loader, wrapped or bare state dict
    path = os.path.join(tmp_path, "text.pt")
    torch.save({"state_dict": sd}, path)
    torch.testing.assert_close(clip_text.load_text_tower(path).encode_text(tokens), got)


def test_class_text_features_follow_the_reference_prompt(tok):
    tower = clip_text.TextTower(_seeded_text_weights(context=77))
    names = ["run", "king"]
    feats = clip_text.class_text_features(names, tok, tower)
    assert feats.shape == (2, 64) and feats.dtype == torch.float32
    want = tower.encode_text(tok.tokenize(["a video of a person run", "a video of a person king"]))          # src/utils.py:48
    torch.testing.assert_close(feats, want)
