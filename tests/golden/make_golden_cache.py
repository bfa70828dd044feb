"""Golden fixture for the MIND TSV -> cache pipeline, made by running the REAL
reference's cache_news / cache_behaviors (utils/dataset.py:298-427, utils/util.py:92-123) in the build
container on small synthetic TSV files and a synthetic WordPiece vocabulary.

    python tests/golden/make_golden_cache.py      ->  tests/golden/cache_pipeline.json

The fixture is data only: the TSV lines and vocabulary this script invented, and the lists / dicts the
reference wrote into its pickles.  /root/reference is not needed to run the tests.
"""
import json
import os
import pickle
import random
import sys
import tempfile
import types

import numpy as np

np.bool8 = np.bool_
REF = "/root/reference/src"
OUT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REF)

WORDS = ("the a of and to in for on with new says after over from at by is as how why what this that his her "
         "trump biden game season team win loss nfl nba star video show first year years day best top make "
         "police man woman dies dead fire home house city state school health covid weather storm football "
         "recipe easy chicken dinner stock market deal price car cars review photos week big little "
         "coach player players score final report reports update live watch shooting crash found killed").split()
PIECES = "s ed ing er est ly ness ment tion able 's n't re un pre".split()


def make_vocab():
    vocab = ["[PAD]"] + ["[unused%d]" % i for i in range(99)] + ["[UNK]", "[CLS]", "[SEP]", "[MASK]"]
    vocab += list("!\"#$%&'()*+,-./:;<=>?@[\\]^_`{|}~")
    vocab += list("0123456789") + list("abcdefghijklmnopqrstuvwxyz")
    vocab += ["##" + c for c in "0123456789abcdefghijklmnopqrstuvwxyz"]
    vocab += ["中", "国", "##é"]                       # two CJK characters and one accented continuation
    vocab += sorted(set(WORDS)) + ["##" + p for p in PIECES if p.isalpha()]
    return list(dict.fromkeys(vocab))


def make_sentence(rng, n):
    out = []
    for _ in range(n):
        r = rng.random()
        w = rng.choice(WORDS)
        if r < 0.15:
            w = w + rng.choice(["s", "ed", "ing", "er", "ly"])
        elif r < 0.22:
            w = w.capitalize()
        elif r < 0.27:
            w = w.upper()
        elif r < 0.32:
            w = str(rng.randint(0, 2021))
        elif r < 0.36:
            w = w + rng.choice([",", ".", "!", "?", ":", "'s", "-" + rng.choice(WORDS)])
        elif r < 0.39:
            w = rng.choice(["café", "naïve", "中国", "中文", "résumé", "£" + "5", "x" * 120, "zzqj", " ",
                            "—", "it’s", "$1.5m", "u.s.", "#metoo", "a\tb" .replace("\t", " "), "é"])
        out.append(w)
    return " ".join(out)


def make_split(rng, root, split, n_news, n_impr, users, prefix):
    d = os.path.join(root, "MIND", "MINDlarge_" + split)
    os.makedirs(d)
    nids = []
    news_lines = []
    for i in range(n_news):
        nid = "%s%d" % (prefix, rng.randint(10000, 99999) * 100 + i)
        nids.append(nid)
        title = make_sentence(rng, rng.choice([0, 1, 3, 6, 9, 14, 30]) if i % 7 == 0 else rng.randint(4, 14))
        abst = "" if i % 5 == 0 else make_sentence(rng, rng.randint(5, 60))
        news_lines.append("\t".join([nid, "news", "newsus", title, abst, "https://example.invalid/%d.html" % i, "[]", "[]"]))
    with open(os.path.join(d, "news.tsv"), "w", encoding="utf-8") as f:
        f.write("\n".join(news_lines) + "\n")
    beh_lines = []
    for k in range(n_impr):
        uid = rng.choice(users)
        his = " ".join(rng.sample(nids, rng.choice([0, 1, 3, 8, 20])))
        cands = rng.sample(nids, rng.choice([1, 2, 5, 9, 23]))
        if split == "test":
            impr = " ".join(cands)
        else:
            labs = [int(rng.random() < 0.3) for _ in cands]
            if split == "train" and sum(labs) == 0 and k % 3:
                labs[0] = 1
            impr = " ".join("%s-%d" % (c, l) for c, l in zip(cands, labs))
        beh_lines.append("\t".join([str(k + 1), uid, "11/13/2019 8:36:57 AM", his, impr]))
    with open(os.path.join(d, "behaviors.tsv"), "w", encoding="utf-8") as f:
        f.write("\n".join(beh_lines) + "\n")
    return news_lines, beh_lines


def main():
    from utils import dataset as R_ds                      # the reference's module
    rng = random.Random(20240611)
    vocab = make_vocab()
    impr_size = 4
    with tempfile.TemporaryDirectory() as tmp:
        data_root, cache_root = os.path.join(tmp, "Data"), os.path.join(tmp, "cache")
        plm_dir = os.path.join(data_root, "PLM", "bert")
        os.makedirs(plm_dir)
        with open(os.path.join(plm_dir, "vocab.txt"), "w", encoding="utf-8") as f:
            f.write("\n".join(vocab) + "\n")
        with open(os.path.join(plm_dir, "config.json"), "w") as f:
            json.dump({"model_type": "bert", "vocab_size": len(vocab)}, f)
        with open(os.path.join(plm_dir, "tokenizer_config.json"), "w") as f:
            json.dump({"do_lower_case": True, "tokenizer_class": "BertTokenizer"}, f)
        users = ["U%d" % rng.randint(1, 99999) for _ in range(12)]
        fixture = {"vocab": vocab, "impr_size": impr_size, "max_title_length": 12, "max_abs_length": 24, "splits": {}}
        manager = types.SimpleNamespace(plm_dir=plm_dir, max_title_length=12, max_abs_length=24, impr_size=impr_size,
                                        data_root=data_root, cache_root=cache_root)
        tsv = {}
        for split, n_news, n_impr, prefix in (("train", 40, 25, "N"), ("dev", 30, 15, "N"), ("test", 30, 15, "N")):
            tsv[split] = make_split(rng, data_root, split, n_news, n_impr, users + (["U7", "U8"] if split != "train" else []), prefix)
        for split in ("train", "dev", "test"):
            d = os.path.join(data_root, "MIND", "MINDlarge_" + split)
            news_cache = os.path.join(cache_root, "MIND", "MINDlarge_" + split, "news")
            beh_cache = os.path.join(cache_root, "MIND", "MINDlarge_" + split, "behaviors")
            if split != "train":
                beh_cache = os.path.join(beh_cache, str(impr_size))
            R_ds.cache_news(os.path.join(d, "news.tsv"), news_cache, manager)
            with open(os.path.join(news_cache, "nid2index.pkl"), "rb") as f:
                nid2index = pickle.load(f)
            R_ds.cache_behaviors(os.path.join(d, "behaviors.tsv"), beh_cache, nid2index, manager)
            blob = {"news_tsv": tsv[split][0], "behaviors_tsv": tsv[split][1], "nid2index": nid2index}
            for name in ("title_token_ids", "abs_token_ids"):
                with open(os.path.join(news_cache, name + ".pkl"), "rb") as f:
                    blob[name] = pickle.load(f)
            with open(os.path.join(beh_cache, "behaviors.pkl"), "rb") as f:
                beh = pickle.load(f)
            blob["behaviors"] = {k: [list(x) if isinstance(x, tuple) else x for x in v] for k, v in beh.items()}
            fixture["splits"][split] = blob
        with open(os.path.join(cache_root, "MIND", "uid2index.pkl"), "rb") as f:
            fixture["uid2index"] = pickle.load(f)
    with open(os.path.join(OUT, "cache_pipeline.json"), "w", encoding="utf-8") as f:
        json.dump(fixture, f, ensure_ascii=True, separators=(",", ":"))
    n_tok = sum(len(t) for s in fixture["splits"].values() for t in s["title_token_ids"])
    print("cache pipeline golden ok:", {k: len(v["news_tsv"]) for k, v in fixture["splits"].items()}, "title tokens", n_tok,
          "users", len(fixture["uid2index"]))


if __name__ == "__main__":
    main()
