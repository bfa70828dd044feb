"""Write MIND-shaped synthetic caches in the reference's pickle formats (tests, bench, smoke runs).
Not part of the reference: there is no network here, hence no MIND download."""
import os
import random

from utils.util import save_pickle


def write_synthetic_caches(manager, n_train_impr=64, n_dev_impr=24, vocab=None, seed=0, splits=("train", "dev")):
    rng = random.Random(seed)
    V = vocab or manager.vocab_size
    out = {}
    for split in splits:
        name = f"MIND{manager.scale}_{split}"
        n_news = manager.news_nums[name]
        root = os.path.join(manager.cache_root, "MIND", name)
        os.makedirs(os.path.join(root, "news"), exist_ok=True)
        titles = []
        for i in range(n_news):
            if i % 97 == 13:
                titles.append([101, 102])                  # too short: stays an empty row
                continue
            n = rng.randint(4, manager.max_title_length)
            titles.append([101] + [rng.randint(min(1000, V // 2), V - 1) for _ in range(n - 2)] + [102])
        save_pickle(titles, os.path.join(root, "news", "title_token_ids.pkl"))
        abstracts = []
        arng = random.Random(f"{seed}-{split}-abs")        # its own stream: the title / behaviour draws stay what they were
        for i in range(n_news):
            if i % 5 == 0:
                abstracts.append([101, 102])               # no abstract
                continue
            n = arng.randint(4, manager.max_abs_length)
            abstracts.append([101] + [arng.randint(min(1000, V // 2), V - 1) for _ in range(n - 2)] + [102])
        save_pickle(abstracts, os.path.join(root, "news", "abs_token_ids.pkl"))
        save_pickle({f"N{i}": i for i in range(1, n_news + 1)}, os.path.join(root, "news", "nid2index.pkl"))
        n_impr = n_train_impr if split == "train" else n_dev_impr
        histories = [[rng.randint(1, n_news) for _ in range(rng.choice([0, 3, 17, 50, 70]))] for _ in range(n_impr)]
        users = [rng.randint(0, 999) for _ in range(n_impr)]
        if split == "train":
            bdir = os.path.join(root, "behaviors")
            imprs, negatives = [], []
            for k in range(n_impr):
                negatives.append([rng.randint(1, n_news) for _ in range(rng.choice([1, 2, 3, 9, 40]))])
                for _ in range(rng.randint(1, 2)):
                    imprs.append((k, rng.randint(1, n_news)))
            blob = {"imprs": imprs, "user_indices": users, "histories": histories, "negatives": negatives}
        else:
            bdir = os.path.join(root, "behaviors", str(manager.impr_size))
            imprs, labels = [], []
            for k in range(n_impr):
                n = rng.randint(2, 3 * manager.impr_size)
                news = [rng.randint(1, n_news) for _ in range(n)]
                lab = [1 if rng.random() < 0.2 else 0 for _ in range(n)]
                lab[0], lab[1] = 1, 0
                for s in range(0, n, manager.impr_size):
                    imprs.append((k, news[s:s + manager.impr_size]))
                    labels.append(lab[s:s + manager.impr_size])
            blob = {"imprs": imprs, "histories": histories, "user_indices": users}
            if split == "dev":
                blob["labels"] = labels
        os.makedirs(bdir, exist_ok=True)
        save_pickle(blob, os.path.join(bdir, "behaviors.pkl"))
        out[split] = root
    return out
