#!/usr/bin/env python3
"""Producer of the --parse_json records hybridgl_amd.main consumes, for a box that has the
reference checkout and its spaCy model (spacy==3.7.6, en_core_web_lg==3.7.1: environment.yaml:199,259).  Neither is in this
image and nothing of the reference is copied here: the script IMPORTS the reference's own extractors (utils.py:31-133,198-238)
from --reference at run time and applies them exactly as Hybridgl_main.py:131-157 / Hybridgl_main_PhraseCut.py:118-140 do.

    python tools/make_parse_records.py --reference /path/to/HybridGL --refer_data_root ./refer/data --dataset refcocog \\
           --split val --out parse_refcocog_val.json
    python tools/make_parse_records.py --reference /path/to/HybridGL --phrasecut_root ./VGPhraseCut_v0 --split test \\
           --out parse_phrasecut_test.json

One record per sentence, keyed by str(sent_id) (REFER) or by the phrase string (PhraseCut):
    {"sentence_for_spacy": str, "noun_phrase": str, "other_nouns": [str, ...], "dirflag": str, "relaflag": str}
"""
import argparse
import json
import os
import sys


def records_for(sentence, U, nlp):
    """Hybridgl_main.py:131-157 for one raw sentence -> the record"""
    s = sentence.lower()
    toks = [t.text for t in nlp(s) if t.text != " "]                            # :133-139
    sfs = " ".join(toks)
    noun_phrase, _, _ = U.extract_noun_phrase(sfs, nlp, need_index=True)          # :146
    others, _ = U.extract_nouns(sfs, nlp)                                         # :155 (bare phrases; the driver adds 'a photo of ')
    return {"sentence_for_spacy": sfs, "noun_phrase": noun_phrase, "other_nouns": list(others),
            "dirflag": U.extract_dir_phrase(sfs, nlp, False),                     # :141
            "relaflag": U.extract_rela_word(sfs, nlp)}                            # :173


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference", required=True, help="checkout of fhgyuanshen/HybridGL (utils.py is imported from it)")
    ap.add_argument("--refer_data_root", default="")
    ap.add_argument("--dataset", default="refcocog")
    ap.add_argument("--split", default="val")
    ap.add_argument("--phrasecut_root", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    import spacy
    sys.path.insert(0, a.reference)
    import utils as U                              # the reference's extractors
    nlp = spacy.load("en_core_web_lg")             # Hybridgl_main.py:50
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = {}
    if a.phrasecut_root:
        tasks = json.load(open(os.path.join(a.phrasecut_root, f"refer_{a.split}.json")))
        for t in tasks:
            out.setdefault(t["phrase"], records_for(t["phrase"], U, nlp))
    else:
        from hybridgl_amd.refer_io import ReferDataset
        ds = ReferDataset(a.refer_data_root, a.dataset, "umd" if a.dataset == "refcocog" else "unc", a.split)
        for i, rid in enumerate(ds.ref_ids):
            ref = ds.refer.Refs[rid]
            for sent_id, raw in zip(ref["sent_ids"], ds.sentence_raws[i]):
                out[str(sent_id)] = records_for(raw, U, nlp)
    json.dump(out, open(a.out, "w"))
    print(len(out), "records ->", a.out)


if __name__ == "__main__":
    main()
