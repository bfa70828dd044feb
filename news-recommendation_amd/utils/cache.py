"""MIND TSV -> cache pipeline: the on-disk pickles are the interchange format with the
reference's caches, so a cache written here loads in the reference and vice versa.

Mirrors, by behaviour, the reference's utils/dataset.py:298-427 (tokenize_news, cache_news,
cache_behaviors) and utils/util.py:92-123 (construct_nid2index, construct_uid2index).  The reference
tokenises with `transformers.AutoTokenizer.from_pretrained(plm_dir).encode(text, max_length=L)`; the
uncased BERT WordPiece algorithm that call runs is restated below over `plm_dir/vocab.txt`, so caching
needs neither `transformers` nor a network.  Pinned by tests/golden/cache_pipeline.json (outputs of the
reference's own functions on synthetic TSV files).
"""
import os
import re
import unicodedata

from utils.util import load_pickle, save_pickle

_SPECIALS = ("[UNK]", "[SEP]", "[PAD]", "[CLS]", "[MASK]")
# token separators: what BERT's cleaner turns into a space (" \t\n\r", the Zs category) plus the line / paragraph
# separators it keeps and the whitespace split then breaks on; every other separator is a control character: dropped
_WS_RE = re.compile("[ \t\n\r\u00a0\u1680\u2000-\u200a\u2028\u2029\u202f\u205f\u3000]+")


def _is_control(ch):
    if ch in "\t\n\r":
        return False
    return unicodedata.category(ch).startswith("C")


def _is_punctuation(ch):
    cp = ord(ch)
    if 33 <= cp <= 47 or 58 <= cp <= 64 or 91 <= cp <= 96 or 123 <= cp <= 126:
        return True
    return unicodedata.category(ch).startswith("P")


def _is_cjk(cp):
    return (0x4E00 <= cp <= 0x9FFF or 0x3400 <= cp <= 0x4DBF or 0x20000 <= cp <= 0x2A6DF or 0x2A700 <= cp <= 0x2B73F
            or 0x2B740 <= cp <= 0x2B81F or 0x2B820 <= cp <= 0x2CEAF or 0xF900 <= cp <= 0xFAFF or 0x2F800 <= cp <= 0x2FA1F)


class WordPieceTokenizer:
    """uncased BERT tokenizer: clean -> space out CJK -> lower-case + strip accents -> split on
    whitespace and punctuation -> greedy longest-match WordPiece ("##" continuations, words over 100
    characters or without a match become [UNK])."""

    def __init__(self, vocab_path, do_lower_case=True, max_chars_per_word=100):
        self.vocab = {}
        with open(vocab_path, "r", encoding="utf-8") as f:
            for i, line in enumerate(f):
                self.vocab.setdefault(line.rstrip("\n"), i)
        self.do_lower_case = do_lower_case
        self.max_chars = max_chars_per_word
        self.unk, self.cls, self.sep = self.vocab["[UNK]"], self.vocab["[CLS]"], self.vocab["[SEP]"]
        self._special_re = re.compile("(" + "|".join(re.escape(t) for t in _SPECIALS if t in self.vocab) + ")")
        self._word_cache = {}
        self._token_cache = {}

    @classmethod
    def from_pretrained(cls, plm_dir):
        path = os.path.join(plm_dir, "vocab.txt")
        if not os.path.exists(path):
            raise FileNotFoundError(f"{path} missing: the TSV -> cache step needs the PLM's WordPiece vocabulary")
        return cls(path)

    def _basic(self, tok):
        """one whitespace-delimited token -> lower-cased, accent-stripped words and punctuation marks"""
        out = []
        for ch in tok:
            cp = ord(ch)
            if cp == 0 or cp == 0xFFFD or _is_control(ch):
                continue
            out.append(" " + ch + " " if _is_cjk(cp) else ch)
        words = []
        for tok in unicodedata.normalize("NFC", "".join(out)).split(" "):
            if self.do_lower_case:
                tok = "".join(c for c in unicodedata.normalize("NFD", tok.lower()) if unicodedata.category(c) != "Mn")
            cur = []
            for ch in tok:
                if _is_punctuation(ch):
                    if cur:
                        words.append("".join(cur))
                        cur = []
                    words.append(ch)
                else:
                    cur.append(ch)
            if cur:
                words.append("".join(cur))
        return words

    def _token_ids(self, tok):
        hit = self._token_cache.get(tok)
        if hit is None:
            hit = []
            for w in self._basic(tok):
                hit.extend(self._wordpiece(w))
            if len(self._token_cache) < (1 << 20):
                self._token_cache[tok] = hit
        return hit

    def _wordpiece(self, word):
        hit = self._word_cache.get(word)
        if hit is not None:
            return hit
        if len(word) > self.max_chars:
            ids = [self.unk]
        else:
            ids, start, n = [], 0, len(word)
            while start < n:
                end, cur = n, None
                while start < end:
                    piece = word[start:end] if start == 0 else "##" + word[start:end]
                    cur = self.vocab.get(piece)
                    if cur is not None:
                        break
                    end -= 1
                if cur is None:
                    ids = [self.unk]
                    break
                ids.append(cur)
                start = end
        if len(self._word_cache) < (1 << 20):
            self._word_cache[word] = ids
        return ids

    def tokenize_ids(self, text):
        ids = []
        for part in self._special_re.split(text):
            if part in self.vocab and part in _SPECIALS:
                ids.append(self.vocab[part])
                continue
            for tok in _WS_RE.split(part):
                if tok:
                    ids.extend(self._token_ids(tok))
        return ids

    def encode(self, text, max_length=None):
        """[CLS] pieces [SEP], pieces cut from the end so that the total is <= max_length"""
        ids = self.tokenize_ids(text)
        if max_length is not None and len(ids) + 2 > max_length:
            ids = ids[:max(max_length - 2, 0)]
        return [self.cls] + ids + [self.sep]


def _count_lines(path):
    n = 0
    with open(path, "rb") as f:
        for chunk in iter(lambda: f.read(1 << 20), b""):
            n += chunk.count(b"\n")
    return n


def tokenize_news(news_path, cache_dir, news_num, tokenizer, max_title_length, max_abs_length):
    """title_token_ids.pkl / abs_token_ids.pkl: list (file order) of [CLS] … [SEP] id lists"""
    title_token_ids = [[] for _ in range(news_num)]
    abs_token_ids = [[] for _ in range(news_num)]
    with open(news_path, "r", encoding="utf-8") as f:
        for idx, line in enumerate(f):
            _, _, _, title, abstract, _, _, _ = line.strip("\n").split("\t")
            title_token_ids[idx] = tokenizer.encode(title, max_length=max_title_length)
            abs_token_ids[idx] = tokenizer.encode(abstract, max_length=max_abs_length)
    save_pickle(title_token_ids, os.path.join(cache_dir, "title_token_ids.pkl"))
    save_pickle(abs_token_ids, os.path.join(cache_dir, "abs_token_ids.pkl"))


def _first_seen(values, base):
    index = {}
    for v in values:
        if v not in index:
            index[v] = len(index) + base
    return index


def _column(path, col):
    with open(path, "r", encoding="utf-8") as f:
        for line in f:
            line = line.rstrip("\n")
            if line:
                yield line.split("\t")[col]


def construct_nid2index(news_path, cache_dir):
    """news ID -> news index, first-seen order, from 1 (0 is the padded news)"""
    nid2index = _first_seen(_column(news_path, 0), 1)
    save_pickle(nid2index, os.path.join(cache_dir, "nid2index.pkl"))
    return nid2index


def construct_uid2index(data_root, cache_root, scale="large"):
    """user ID -> user index from 0, first-seen order over the train, dev, test behaviour files.  The
    reference always reads the MINDlarge files; when none of them exist (reference: FileNotFoundError)
    the files of the scale in use are read instead."""
    def files(s):
        return [os.path.join(data_root, "MIND", f"MIND{s}_{split}", "behaviors.tsv") for split in ("train", "dev", "test")]
    paths = [p for p in files("large") if os.path.exists(p)]
    if len(paths) < 3 and scale != "large":
        paths = [p for p in files(scale) if os.path.exists(p)]
    elif len(paths) < 3:
        raise FileNotFoundError("uid2index needs " + ", ".join(files("large")))
    if not paths:
        raise FileNotFoundError("uid2index needs behaviors.tsv under " + os.path.join(data_root, "MIND"))
    def uids():
        for p in paths:
            yield from _column(p, 1)
    uid2index = _first_seen(uids(), 0)
    os.makedirs(os.path.join(cache_root, "MIND"), exist_ok=True)
    save_pickle(uid2index, os.path.join(cache_root, "MIND", "uid2index.pkl"))
    return uid2index


def cache_news(news_path, news_cache_root, manager):
    news_num = _count_lines(news_path)
    os.makedirs(news_cache_root, exist_ok=True)
    tokenizer = WordPieceTokenizer.from_pretrained(manager.plm_dir)
    tokenize_news(news_path, news_cache_root, news_num, tokenizer, manager.max_title_length, manager.max_abs_length)
    if not os.path.exists(os.path.join(news_cache_root, "nid2index.pkl")):
        construct_nid2index(news_path, news_cache_root)


def cache_behaviors(behaviors_path, cache_dir, nid2index, manager):
    """behaviors.pkl.  train: one (impr_index, clicked news) pair per click + the impression's negatives;
    dev / test: the candidates of an impression in chunks of impr_size (dev: with their labels)."""
    uid_path = os.path.join(manager.cache_root, "MIND", "uid2index.pkl")
    uid2index = load_pickle(uid_path) if os.path.exists(uid_path) else \
        construct_uid2index(manager.data_root, manager.cache_root, getattr(manager, "scale", "large"))
    os.makedirs(cache_dir, exist_ok=True)
    kind = next((k for k in ("train", "dev", "test") if k in behaviors_path), None)
    if kind is None:
        raise ValueError(f"cannot tell train / dev / test from {behaviors_path}")
    step = manager.impr_size
    imprs, histories, user_indices, negatives, labels = [], [], [], [], []
    with open(behaviors_path, "r", encoding="utf-8") as f:
        for impr_index, line in enumerate(f):
            _, uid, _, history, impression = line.strip("\n").split("\t")
            histories.append([nid2index[x] for x in history.split()])
            user_indices.append(uid2index[uid])
            items = impression.split()
            if kind == "test":
                news = [nid2index[x] for x in items]
            else:
                news = [nid2index[x.split("-")[0]] for x in items]
                label = [int(x.split("-")[1]) for x in items]
            if kind == "train":
                negative = []
                for n, lab in zip(news, label):
                    if lab == 1:
                        imprs.append((impr_index, n))
                    else:
                        negative.append(n)
                negatives.append(negative)
            else:
                for i in range(0, len(news), step):
                    imprs.append((impr_index, news[i:i + step]))
                    if kind == "dev":
                        labels.append(label[i:i + step])
    if kind == "train":
        out = {"imprs": imprs, "user_indices": user_indices, "histories": histories, "negatives": negatives}
    elif kind == "dev":
        out = {"imprs": imprs, "labels": labels, "histories": histories, "user_indices": user_indices}
    else:
        out = {"imprs": imprs, "histories": histories, "user_indices": user_indices}
    save_pickle(out, os.path.join(cache_dir, "behaviors.pkl"))
    return out


def ensure_caches(manager, data_dir, news_cache_root, behaviors_cache_dir):
    """what MIND.__init__ does on rank 0 before loading (reference utils/dataset.py:47-53): build whichever
    cache is missing from data_dir/{news,behaviors}.tsv; returns False when a cache is missing and so is its TSV"""
    ok = True
    if not os.path.exists(os.path.join(news_cache_root, "title_token_ids.pkl")):
        news_path = os.path.join(data_dir, "news.tsv")
        if os.path.exists(news_path):
            cache_news(news_path, news_cache_root, manager)
        else:
            ok = False
    if not os.path.exists(os.path.join(behaviors_cache_dir, "behaviors.pkl")):
        beh_path = os.path.join(data_dir, "behaviors.tsv")
        nid_path = os.path.join(news_cache_root, "nid2index.pkl")
        if os.path.exists(beh_path) and os.path.exists(nid_path):
            cache_behaviors(beh_path, behaviors_cache_dir, load_pickle(nid_path), manager)
        else:
            ok = False
    return ok
