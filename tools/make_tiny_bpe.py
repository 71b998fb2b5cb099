#!/usr/bin/env python3
"""Writes tests/golden/tiny_bpe_vocab.txt.gz: a small byte-level BPE merges file in the format of
OpenAI's bpe_simple_vocab_16e6.txt.gz (header line, then one 'left right' merge per line), learnt
from a fixed toy corpus with a plain most-frequent-pair trainer.  Our own data, not the reference's."""
import collections
import gzip
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from hybridgl_amd.tokenizer import byte_table

CORPUS = ("the cat on the left of the dog . a photo of the bigger elephant behind the small "
          "giraffe's tail ; person in blue shirt holding an umbrella , second banana from right "
          "woman's red hat isn't there they're we've i'm you'll he'd 3 zebras 42 café naïve über "
          "the man standing inside the doorway near the larger window closest to the camera").split()


def main():
    enc, _ = byte_table()
    words = collections.Counter()
    for w in CORPUS:
        sym = [enc[b] for b in w.lower().encode("utf-8")]
        sym[-1] += "</w>"
        words[tuple(sym)] += 1
    merges = []
    for _ in range(180):
        pairs = collections.Counter()
        for w, c in words.items():
            for a, b in zip(w, w[1:]):
                pairs[(a, b)] += c
        if not pairs:
            break
        (a, b), _c = max(sorted(pairs.items()), key=lambda kv: kv[1])
        merges.append((a, b))
        new = collections.Counter()
        for w, c in words.items():
            out, i = [], 0
            while i < len(w):
                if i + 1 < len(w) and w[i] == a and w[i + 1] == b:
                    out.append(a + b); i += 2
                else:
                    out.append(w[i]); i += 1
            new[tuple(out)] += c
        words = new
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tiny_bpe_vocab.txt.gz")
    with gzip.GzipFile(path, "wb", mtime=0) as f:
        f.write(("#version: tiny synthetic\n" + "\n".join(f"{a} {b}" for a, b in merges) + "\n").encode("utf-8"))
    print(len(merges), "merges ->", path)


if __name__ == "__main__":
    main()
