"""CLIP byte-level BPE tokenizer: `tokenize(texts, context_length=77, truncate=False)`
(clip/clip.py:197-237) over `SimpleTokenizer` (clip/simple_tokenizer.py:62-127).

Host-side text processing (the reference does it on the CPU as well); the token matrix it
returns feeds `CLIPViTFM.model.encode_text`.  The merges file is NOT shipped: pass the path of
OpenAI's `bpe_simple_vocab_16e6.txt.gz` (or set HYBRIDGL_BPE_VOCAB); tests pin the algorithm on
a small synthetic merges file against token ids produced by the reference class.
"""
import gzip
import html
import os
from functools import lru_cache

import numpy as np
import regex as re

try:  # the reference calls ftfy.fix_text; it is the identity on well-formed ASCII/UTF-8 input
    import ftfy
    _fix = ftfy.fix_text
except ImportError:  # pragma: no cover
    _fix = lambda s: s

_PAT = re.compile(r"""<\|startoftext\|>|<\|endoftext\|>|'s|'t|'re|'ve|'m|'ll|'d|[\p{L}]+|[\p{N}]|[^\s\p{L}\p{N}]+""",
                  re.IGNORECASE)
_N_MERGES = 49152 - 256 - 2  # simple_tokenizer.py:67


@lru_cache()
def byte_table():
    """printable stand-ins for the 256 byte values (simple_tokenizer.py:15-35)."""
    keep = list(range(ord("!"), ord("~") + 1)) + list(range(ord("¡"), ord("¬") + 1)) + list(range(ord("®"), ord("ÿ") + 1))
    table, extra = {}, 0
    for b in keep:
        table[b] = chr(b)
    for b in range(256):
        if b not in table:
            table[b] = chr(256 + extra)
            extra += 1
    # the reference orders its vocabulary by (kept bytes in order, then the remapped ones)
    order = keep + [b for b in range(256) if b not in keep]
    return table, [table[b] for b in order]


class SimpleTokenizer:
    def __init__(self, bpe_path=None):
        bpe_path = bpe_path or os.environ.get("HYBRIDGL_BPE_VOCAB")
        if not bpe_path or not os.path.exists(bpe_path):
            raise FileNotFoundError("BPE merges file not found: pass bpe_path or set HYBRIDGL_BPE_VOCAB to "
                                    "bpe_simple_vocab_16e6.txt.gz (not redistributed with this package)")
        lines = gzip.open(bpe_path).read().decode("utf-8").split("\n")[1:_N_MERGES + 1]
        merges = [tuple(l.split()) for l in lines]
        self.byte_encoder, base = byte_table()
        vocab = base + [c + "</w>" for c in base] + ["".join(m) for m in merges] + ["<|startoftext|>", "<|endoftext|>"]
        self.encoder = {tok: i for i, tok in enumerate(vocab)}  # later duplicates win, as dict(zip(...)) does
        self.decoder = {i: tok for tok, i in self.encoder.items()}
        self.rank = {m: i for i, m in enumerate(merges)}
        self.sot, self.eot = self.encoder["<|startoftext|>"], self.encoder["<|endoftext|>"]
        self._cache = {"<|startoftext|>": ["<|startoftext|>"], "<|endoftext|>": ["<|endoftext|>"]}

    def _bpe(self, token):
        """greedy lowest-rank pair merging of one pre-token (simple_tokenizer.py:80-118)."""
        hit = self._cache.get(token)
        if hit is not None:
            return hit
        parts = list(token[:-1]) + [token[-1] + "</w>"]
        while len(parts) > 1:
            best, best_rank = None, None
            for a, b in zip(parts, parts[1:]):
                r = self.rank.get((a, b))
                if r is not None and (best_rank is None or r < best_rank):
                    best, best_rank = (a, b), r
            if best is None:
                break
            a, b = best
            merged, i = [], 0
            while i < len(parts):  # merge every non-overlapping occurrence, left to right
                if i + 1 < len(parts) and parts[i] == a and parts[i + 1] == b:
                    merged.append(a + b)
                    i += 2
                else:
                    merged.append(parts[i])
                    i += 1
            parts = merged
        self._cache[token] = parts
        return parts

    def encode(self, text):
        text = html.unescape(html.unescape(_fix(text))).strip()       # basic_clean
        text = re.sub(r"\s+", " ", text).strip().lower()               # whitespace_clean + lower
        out = []
        for tok in _PAT.findall(text):
            sym = "".join(self.byte_encoder[b] for b in tok.encode("utf-8"))
            out.extend(self.encoder[p] for p in self._bpe(sym))
        return out

    def decode(self, tokens):
        inv = {c: b for b, c in self.byte_encoder.items()}
        text = "".join(self.decoder[int(t)] for t in tokens)
        return bytearray(inv[c] for c in text).decode("utf-8", errors="replace").replace("</w>", " ")


_default = None


def tokenize(texts, context_length=77, truncate=False, tokenizer=None):
    """clip.tokenize (clip/clip.py:197-237) -> int32 numpy [len(texts), context_length]."""
    global _default
    if isinstance(texts, str):
        texts = [texts]
    tk = tokenizer
    if tk is None:
        if _default is None:
            _default = SimpleTokenizer()
        tk = _default
    out = np.zeros((len(texts), context_length), dtype=np.int32)
    for i, t in enumerate(texts):
        ids = [tk.sot] + tk.encode(t) + [tk.eot]
        if len(ids) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {t} is too long for context length {context_length}")
            ids = ids[:context_length]
            ids[-1] = tk.eot
        out[i, :len(ids)] = ids
    return out
