"""TEST INFRASTRUCTURE ONLY (imported by tests/ alone): an independent restatement of the two pieces of OpenAI CLIP's text side that
unite_amd/clip_text.py implements, written differently on purpose so that the two can check each other.

PARITY UNPINNED -- the algorithm is a third-party dependency of the reference (``git+https://github.com/openai/CLIP.git``, environment.yaml:353;
call site src/utils.py:44-53) that is absent from /root/reference, as are its BPE vocabulary and weights; the reference holds no vector for it.

* ``bpe_sequential``: byte-pair merging by walking the merge table in rank order and merging every occurrence of each pair (equivalent to
  lowest-rank-pair-first: a pair that contains a symbol created by merge r has a rank above r).
* ``text_forward``: the text transformer with explicit per-text, per-head loops in float64 numpy."""
import numpy as np


def bpe_sequential(symbols, merges):
    """symbols: list of str (last one already carries '</w>'); merges: list of (a, b) in rank order"""
    parts = list(symbols)
    for a, b in merges:
        if len(parts) < 2:
            break
        out, i = [], 0
        while i < len(parts):
            if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                out.append(a + b)
                i += 2
            else:
                out.append(parts[i])
                i += 1
        parts = out
    return parts


def _ln(x, g, b):
    mu = x.mean(-1, keepdims=True)
    var = ((x - mu) ** 2).mean(-1, keepdims=True)
    return (x - mu) / np.sqrt(var + 1e-5) * g + b


def text_forward(sd, tokens, heads):
    """sd: dict of float arrays with OpenAI's key names; tokens: (n, L) ints -> (n, C)"""
    w = {k: np.asarray(v, dtype=np.float64) for k, v in sd.items()}
    layers = 1 + max(int(k.split(".")[2]) for k in w if k.startswith("transformer.resblocks."))
    n, L = tokens.shape
    D = w["token_embedding.weight"].shape[1]
    dh = D // heads
    outs = []
    for t in range(n):
        x = w["token_embedding.weight"][tokens[t]] + w["positional_embedding"]
        for i in range(layers):
            p = f"transformer.resblocks.{i}."
            h = _ln(x, w[p + "ln_1.weight"], w[p + "ln_1.bias"])
            qkv = h @ w[p + "attn.in_proj_weight"].T + w[p + "attn.in_proj_bias"]
            o = np.zeros((L, D))
            for hd in range(heads):
                q = qkv[:, hd * dh:(hd + 1) * dh]
                k = qkv[:, D + hd * dh:D + (hd + 1) * dh]
                v = qkv[:, 2 * D + hd * dh:2 * D + (hd + 1) * dh]
                for r in range(L):                      # row r attends to rows 0..r
                    s = q[r] @ k[:r + 1].T / np.sqrt(dh)
                    e = np.exp(s - s.max())
                    o[r, hd * dh:(hd + 1) * dh] = (e / e.sum()) @ v[:r + 1]
            x = x + o @ w[p + "attn.out_proj.weight"].T + w[p + "attn.out_proj.bias"]
            h = _ln(x, w[p + "ln_2.weight"], w[p + "ln_2.bias"])
            a = h @ w[p + "mlp.c_fc.weight"].T + w[p + "mlp.c_fc.bias"]
            a = a / (1.0 + np.exp(-1.702 * a))
            x = x + a @ w[p + "mlp.c_proj.weight"].T + w[p + "mlp.c_proj.bias"]
        x = _ln(x, w["ln_final.weight"], w["ln_final.bias"])
        outs.append(x[int(np.argmax(tokens[t]))] @ w["text_projection"])
    return np.stack(outs)
