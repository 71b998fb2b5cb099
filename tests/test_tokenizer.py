"""CPU: the BPE tokenizer against token ids produced by the reference's SimpleTokenizer
(oracle/gen_golden.py:gen_tokenizer) -- on the committed tiny synthetic merges file always, and on
OpenAI's real merges file when HYBRIDGL_BPE_VOCAB points at it (it is not redistributed here)."""
import os

import numpy as np
import pytest

from hybridgl_amd.tokenizer import SimpleTokenizer, tokenize


@pytest.fixture(scope="module")
def g(golden_dir):
    return np.load(os.path.join(golden_dir, "tokenizer.npz"))


def _split(g, tag):
    ids, out, o = g[f"{tag}_ids"], [], 0
    for n in g[f"{tag}_len"]:
        out.append([int(v) for v in ids[o:o + n]])
        o += n
    return out


def test_tiny_vocab_matches_reference(g, golden_dir):
    tk = SimpleTokenizer(os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"))
    assert [tk.sot, tk.eot] == [int(v) for v in g["tiny_sot_eot"]]
    for s, ref in zip(g["strings"], _split(g, "tiny")):
        assert tk.encode(str(s)) == ref, s
    assert tk.decode(_split(g, "tiny")[2]) == str(g["tiny_decoded0"])


def test_tokenize_layout_and_errors(g, golden_dir):
    tk = SimpleTokenizer(os.path.join(golden_dir, "tiny_bpe_vocab.txt.gz"))
    m = tokenize(["the cat on left", ""], context_length=16, tokenizer=tk)
    assert m.shape == (2, 16) and m.dtype == np.int32
    assert m[0, 0] == tk.sot and m[1, 0] == tk.sot and m[1, 1] == tk.eot and m[1, 2:].sum() == 0
    assert m[0].argmax() == 1 + len(tk.encode("the cat on left"))     # EOT is the arg-max token (pooling index)
    long = "the cat on the left of the dog " * 8
    with pytest.raises(RuntimeError):
        tokenize(long, context_length=16, tokenizer=tk)
    t = tokenize(long, context_length=16, truncate=True, tokenizer=tk)
    assert t[0, -1] == tk.eot
    with pytest.raises(FileNotFoundError):
        SimpleTokenizer("/nonexistent/merges.txt.gz")


def test_real_vocab_matches_reference(g):
    path = os.environ.get("HYBRIDGL_BPE_VOCAB")
    if not path or not os.path.exists(path):
        pytest.skip("set HYBRIDGL_BPE_VOCAB to bpe_simple_vocab_16e6.txt.gz to run")
    tk = SimpleTokenizer(path)
    assert [tk.sot, tk.eot] == [49406, 49407] == [int(v) for v in g["real_sot_eot"]]
    for s, ref in zip(g["strings"], _split(g, "real")):
        assert tk.encode(str(s)) == ref, s
    assert tk.encode("the cat on left") == [518, 2368, 525, 1823]     # SURVEY.md known answer
