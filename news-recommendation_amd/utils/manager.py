"""Run manager of the two-tower path: flags, data-parallel runtime (one process per MI355X,
torch.distributed "nccl" == RCCL over xGMI), loaders, train loop, checkpoints, logging.

Flag names, destinations and defaults are the reference's (utils/manager.py:41-89) so the same
command lines work; methods keep its names (setup / prepare / save / load / _log / _train / train).
Differences, all deliberate: no model download (no network: a missing PLM dir is an error unless
`synthetic_plm`), no SMTP e-mail in `_log`, rendezvous on 127.0.0.1, a gloo process group when
`-d cpu` (host-logic tests), the loss is accumulated on the device (one host sync per 5 steps
instead of every step) and `plm_dim` / `vocab_size` follow the local PLM's config.json when present
(the reference pins 768 / 30522), which makes the 300-d NRMS shape reachable from the flags.
"""
import argparse
import json
import logging
import os
import random
from datetime import timedelta

import numpy as np
import torch
import torch.distributed as dist
from torch.utils.data import DataLoader
from torch.utils.data.distributed import DistributedSampler
from tqdm import tqdm

from utils.util import Sequential_Sampler

logger = logging.getLogger("Manager")
logging.basicConfig(level=logging.INFO, format="[%(asctime)s] %(levelname)s (%(name)s) %(message)s")
logging.getLogger("torch.distributed.distributed_c10d").setLevel(logging.WARNING)


def _device_arg(x):
    return "cpu" if x == "cpu" else int(x)


# (flags, dest, kwargs) — one row per reference flag
_FLAGS = [
    (("-s", "--scale"), "scale", dict(type=str, choices=["demo", "small", "large", "whole"], default="large", help="data scale")),
    (("-e", "--epoch"), "epochs", dict(type=int, default=10, help="epochs to train the model")),
    (("-m", "--mode"), "mode", dict(default="train", help="train | dev | test")),
    (("-d", "--device"), "device", dict(type=_device_arg, default=0, help="gpu index, or 'cpu'")),
    (("-bs", "--batch-size"), "batch_size", dict(type=int, default=32, help="batch size in training")),
    (("-bse", "--batch-size-eval"), "batch_size_eval", dict(type=int, default=200, help="batch size in encoding / scoring")),
    (("-ck", "--checkpoint"), "checkpoint", dict(type=str, default="none", help="none | best | <step> | <path>")),
    (("-vs", "--validate-step"), "validate_step", dict(type=str, default="0", help="evaluate every N steps ('0.5e' = half an epoch)")),
    (("-hst", "--hold-step"), "hold_step", dict(type=str, default="0", help="do not evaluate before this step")),
    (("-sav", "--save-at-validate"), "save_at_validate", dict(action="store_true", default=False)),
    (("-vb", "--verbose"), "verbose", dict(type=str, default=None, help="variant name appended to the model name")),
    (("--metrics",), "metrics", dict(nargs="+", action="extend", default=["auc", "mean_mrr", "ndcg@5", "ndcg@10"])),
    (("-hs", "--his_size"), "his_size", dict(type=int, default=50)),
    (("-is", "--impr_size"), "impr_size", dict(type=int, default=20)),
    (("-nn", "--negative-num"), "negative_num", dict(type=int, default=4)),
    (("-dp", "--dropout-p"), "dropout_p", dict(type=float, default=0.1)),
    (("-lr", "--learning-rate"), "learning_rate", dict(type=float, default=1e-5)),
    (("-sch", "--scheduler"), "scheduler", dict(choices=["linear", "none"], default="none")),
    (("--warmup",), "warmup", dict(type=float, default=0.1)),
    (("-pth", "--preprocess-threads"), "preprocess_threads", dict(type=int, default=32)),
    (("-dr", "--data-root"), "data_root", dict(default="../../../Data")),
    (("-cr", "--cache-root"), "cache_root", dict(default="data/cache")),
    (("-tl", "--title-length"), "title_length", dict(type=int, default=32)),
    (("-al", "--abs-length"), "abs_length", dict(type=int, default=64)),
    (("-mtl", "--max-title-length"), "max_title_length", dict(type=int, default=64)),
    (("-mal", "--max-abs-length"), "max_abs_length", dict(type=int, default=128)),
    (("-ef", "--enable-fields"), "enable_fields", dict(nargs="+", action="extend", choices=["title", "abs"], default=[])),
    (("-ne", "--news-encoder"), "newsEncoder", dict(default="cnn")),
    (("-ue", "--user-encoder"), "userEncoder", dict(default="rnn")),
    (("-hd", "--hidden-dim"), "hidden_dim", dict(type=int, default=768)),
    (("-hn", "--head-num"), "head_num", dict(type=int, default=12)),
    (("-k",), "k", dict(type=int, default=4, help="gate number (one-tower models; unused here)")),
    (("-plm",), "plm", dict(type=str, default="bert")),
    (("--seed",), "seed", dict(type=int, default=3407)),
    (("-ws", "--world-size"), "world_size", dict(type=int, default=1, help="number of GPUs (one process each)")),
    (("-br", "--base-rank"), "base_rank", dict(type=int, default=0, help="first device index")),
    (("--debug",), "debug", dict(action="store_true", default=False)),
    # extensions (not in the reference)
    (("--synthetic-plm",), "synthetic_plm", dict(action="store_true", default=False, help="random token table instead of a PLM dir")),
    (("--bf16",), "bf16", dict(action="store_true", default=False,
                               help="bf16-operand matrix cores (f32 accumulate, tensors stay fp32) for every large dense product: "
                                    "the 1e-2-parity build meant for -ne bert; default is exact fp32")),
    (("--host-assembly",), "host_assembly", dict(action="store_true", default=False,
                                                 help="ship token ids/masks in every batch like the reference (default on a GPU: "
                                                      "batches carry news indices only and the token rows are gathered on the device)")),
]

_PLM = {
    "bert": dict(full_name="bert-base-uncased", dim=768, vocab_size=30522),
    "distilbert": dict(full_name="distilbert-base-uncased", dim=768, vocab_size=30522),
}
_SPECIAL = {"[PAD]": 0, "[CLS]": 101, "[SEP]": 102, "punctuations": {}}
_LOADERS = {"train": ["train", "dev", "news"], "dev": ["dev", "news"], "test": ["test", "news"]}
NEWS_NUMS = {"MINDdemo_train": 51282, "MINDdemo_dev": 42416, "MINDsmall_train": 51282, "MINDsmall_dev": 42416,
             "MINDlarge_train": 101527, "MINDlarge_dev": 72023, "MINDlarge_test": 120961}


def _with_next(it):
    """(item, next item or None) pairs of an iterable: one element of look-ahead"""
    it = iter(it)
    try:
        cur = next(it)
    except StopIteration:
        return
    for nxt in it:
        yield cur, nxt
        cur = nxt
    yield cur, None


class Manager():
    def __init__(self, config=None, command=None):
        parser = argparse.ArgumentParser()
        for flags, dest, kw in _FLAGS:
            parser.add_argument(*flags, dest=dest, **kw)
        if config:
            parser.set_defaults(**config)
        args = vars(parser.parse_args(command) if command is not None else parser.parse_args())

        if args["debug"]:
            args["hold_step"], args["validate_step"] = "0", "2"
        if args["mode"] != "train" and args["checkpoint"] == "none":
            args["checkpoint"] = "best"
        seq = (args["title_length"] if "title" in args["enable_fields"] else 0) + \
              (args["abs_length"] if "abs" in args["enable_fields"] else 0)
        if seq == 0:
            raise ValueError("Include at least one field!")
        args["sequence_length"] = seq

        if args["seed"] is not None:
            seed = args["seed"]
            random.seed(seed)
            os.environ["PYTHONHASHSEED"] = str(seed)
            np.random.seed(seed)
            torch.manual_seed(seed)
            if torch.cuda.is_available():
                torch.cuda.manual_seed_all(seed)

        for k, v in args.items():
            setattr(self, k, v)

        plm = _PLM[self.plm]
        self.plm_dir = os.path.join(self.data_root, "PLM", self.plm)
        self.plm_dim, self.vocab_size, self.plm_full_name = plm["dim"], plm["vocab_size"], plm["full_name"]
        cfg = os.path.join(self.plm_dir, "config.json")
        if os.path.isfile(cfg):
            with open(cfg) as f:
                c = json.load(f)
            self.plm_dim = int(c.get("hidden_size", c.get("dim", self.plm_dim)))
            self.vocab_size = int(c.get("vocab_size", self.vocab_size))
        elif self.synthetic_plm and self.newsEncoder == "tfm":
            self.plm_dim = self.hidden_dim           # synthetic table: whatever the transformer needs
        self.special_token_ids = dict(_SPECIAL)
        self.news_nums = dict(NEWS_NUMS)
        self.dataloaders = _LOADERS[self.mode]
        self.rank = 0
        self.distributed = self.world_size > 1
        if self.distributed:                       # inherited by the mp.spawn children before their first GPU call (see setup)
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        self.exclude_hparams = set([
            "news_nums", "vocab_size_map", "metrics", "plm_dim", "plm_dir", "data_root", "cache_root", "distributed",
            "exclude_hparams", "rank", "epochs", "mode", "debug", "special_token_ids", "validate_step", "hold_step",
            "device", "save_at_validate", "preprocess_threads", "base_rank", "max_title_length", "max_abs_length"])
        logger.info("Hyper Parameters are:\n{}\n".format({k: v for k, v in args.items() if k not in self.exclude_hparams}))

    # ------------------------------------------------------------------ runtime
    def setup(self, rank):
        """one process per GPU; RCCL process group when world_size > 1 (reference manager.py:189-217).
        The environment RCCL needs on this platform is set here, so `python -m main.twotower -ws 8` has what
        `bench.py --gpus 8` has: HSA_ENABLE_IPC_MODE_LEGACY=0 (the host driver only offers dmabuf IPC; without it RCCL's
        buffer exchange fails with hipIpcGetMemHandle: invalid argument) — set before this process makes its first GPU
        call (Manager.__init__ sets it too, so that mp.spawn children inherit it) — and the process group is bound to its
        device (device_id=), so the communicator is created eagerly on the right GPU and barriers need no device guess.
        NR_REHEARSE_ONE_GPU=1 (tests only, never set by the product): every rank on device 0 over gloo, the multi-rank
        logic without N GPUs."""
        os.environ["TOKENIZERS_PARALLELISM"] = "True"
        if self.world_size > 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(12355 + self.base_rank))
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            rehearse = os.environ.get("NR_REHEARSE_ONE_GPU") == "1"
            backend = "gloo" if (self.device == "cpu" or rehearse) else "nccl"
            self.rank = rank
            kw = {}
            if self.device != "cpu":
                self.device = self.base_rank if rehearse else rank + self.base_rank
                torch.cuda.set_device(self.device)
                if backend == "nccl":
                    kw["device_id"] = torch.device("cuda", self.device)
            dist.init_process_group(backend, rank=rank, world_size=self.world_size, timeout=timedelta(0, 1000000), **kw)
        else:
            self.rank = 0
            if self.device != "cpu":
                torch.cuda.set_device(self.device)
        if self.device != "cpu":
            from newsrec_hip import ops
            ops.set_gemm_compute("bf16" if getattr(self, "bf16", False) else "fp32")

    def barrier(self):
        if self.distributed:
            if isinstance(self.device, int) and dist.get_backend() == "nccl":
                dist.barrier(device_ids=[self.device])
            else:
                dist.barrier()

    def prepare(self):
        from utils.dataset import MIND_Train, MIND_Dev, MIND_Test, MIND_News
        if self.rank == 0 and not self.synthetic_plm and not os.path.exists(self.plm_dir):
            raise FileNotFoundError(f"{self.plm_dir} not found; this build never downloads {self.plm_full_name} "
                                    "(no network). Copy the PLM directory there or pass --synthetic-plm.")
        self.barrier()
        loaders = {}
        if "train" in self.dataloaders:
            ds = MIND_Train(self)
            if self.distributed:
                sampler, shuffle = DistributedSampler(ds, num_replicas=self.world_size, rank=self.rank, seed=self.seed), False
            else:
                sampler, shuffle = None, True
            loaders["train"] = DataLoader(ds, batch_size=self.batch_size, shuffle=shuffle, sampler=sampler)
        for name, cls in (("dev", MIND_Dev), ("test", MIND_Test)):
            if name in self.dataloaders:
                ds = cls(self)
                sampler = Sequential_Sampler(len(ds), num_replicas=self.world_size, rank=self.rank)
                loaders[name] = DataLoader(ds, batch_size=self.batch_size_eval, sampler=sampler, drop_last=False)
        if "news" in self.dataloaders:
            loaders["news"] = DataLoader(MIND_News(self), batch_size=self.batch_size_eval, drop_last=False)
        return loaders

    # ------------------------------------------------------------------ checkpoints / log
    @staticmethod
    def _unwrap(model):
        from utils.dp import FlatGradSync
        return model.module if isinstance(model, (torch.nn.parallel.DistributedDataParallel, FlatGradSync)) else model

    def _ckpt_path(self, model, tag):
        return f"data/ckpts/{model.name}/{self.scale}/{tag}.model"

    def save(self, model, step, best=False):
        model = self._unwrap(model)
        path = self._ckpt_path(model, "best" if best else step)
        os.makedirs(os.path.dirname(path), exist_ok=True)
        logger.info("saving model at {}...".format(path))
        hparams = {k: v for k, v in vars(self).items() if k not in self.exclude_hparams}
        torch.save({"manager": hparams, "model": model.state_dict()}, path)

    def load(self, model):
        """-ck none | best | <step> | <file>; weights only (no optimizer state), strict=False — like the
        reference.  Files are read with weights_only=True (tensors + plain python containers)."""
        model = self._unwrap(model)
        ck = self.checkpoint
        if ck == "none":
            return
        path = ck if os.path.isfile(ck) else self._ckpt_path(model, ck)
        if not os.path.exists(path):
            if self.rank == 0:
                logger.warning(f"Checkpoint {path} Not Found, Not Loading Any Checkpoints!")
            return
        if self.rank == 0:
            logger.info("loading model from {}...".format(path))
        dev = torch.device("cpu") if model.device == "cpu" else torch.device("cuda", model.device)
        state = torch.load(path, map_location=dev, weights_only=True)
        if self.rank == 0:
            mine = vars(self)
            for k, v in state.get("manager", {}).items():
                if k in {"dataloaders", "checkpoint"}:
                    continue
                if k not in mine:
                    logger.info(f"manager settings {k} not found!")
                elif v != mine[k]:
                    logger.info(f"manager settings {k} of the checkpoint is {v}, while it's {mine[k]} in current setting!")
        missing, unexpected = model.load_state_dict(state["model"], strict=False)
        if self.rank == 0:
            if missing:
                logger.warning(f"Missing Keys: {missing}")
            if unexpected:
                logger.warning(f"Unexpected Keys: {unexpected}")

    def _log(self, model_name, metrics):
        hp = {k: v for k, v in vars(self).items() if k not in self.exclude_hparams}
        with open("performance.log", "a+") as f:
            f.write("{} : {}\n{}\n\n".format(model_name, str(hp), str(metrics)))

    # ------------------------------------------------------------------ training
    def _train(self, model, loaders, validate_step, hold_step, optimizer, scheduler, save_at_validate=False):
        core = self._unwrap(model)
        loader = loaders["train"]
        best = {"main": -1.}
        total_steps = 1
        if self.rank == 0:
            logger.info("training {}...".format(core.name))
        # The datasets, the model and the optimizer are millions of long-lived Python objects; a full (generation 2)
        # garbage collection walks all of them and was measured at 65-80 ms, every few hundred steps, in the middle of a
        # 2.3 ms step (tools/host_slack.py).  Moving what exists now to the permanent generation keeps collections to the
        # objects a step creates.
        from utils.util import frozen_gc
        with frozen_gc():
            return self._train_loop(model, core, loader, loaders, validate_step, hold_step, optimizer, scheduler, save_at_validate, best, total_steps)

    def _train_loop(self, model, core, loader, loaders, validate_step, hold_step, optimizer, scheduler, save_at_validate, best, total_steps):
        for epoch in range(self.epochs):
            if self.distributed and hasattr(loader.sampler, "set_epoch"):
                loader.sampler.set_epoch(epoch)
            epoch_loss = None
            bar = tqdm(loader, ncols=120, disable=self.rank != 0)
            for step, (x, x_next) in enumerate(_with_next(bar), 1):
                if hasattr(model, "sync"):                 # FlatGradSync: gradients live in one flat buffer
                    model.zero_grad()
                else:
                    optimizer.zero_grad(set_to_none=True)
                loss = model(x)
                if x_next is not None and hasattr(core, "plan_ahead"):
                    core.plan_ahead(x_next)                # the next batch's pack plan rides along with this step
                loss.backward()
                if hasattr(model, "sync"):
                    model.sync()
                optimizer.step()
                if scheduler:
                    scheduler.step()
                d = loss.detach()
                epoch_loss = d if epoch_loss is None else epoch_loss + d        # stays on the device
                if step % 5 == 0:
                    if hasattr(core, "_check_indices"):
                        core._check_indices("train step")         # the float() below syncs anyway
                    bar.set_description("epoch: {:d}, step: {:d}, loss: {:.4f}".format(epoch + 1, step, float(epoch_loss) / step))
                if total_steps > hold_step and total_steps % validate_step == 0:
                    result = core.dev(self, loaders)
                    if self.rank == 0:
                        result["step"] = total_steps
                        if save_at_validate:
                            self.save(model, total_steps)
                        if result["main"] >= best["main"]:
                            best = result
                            self.save(model, total_steps, best=True)
                            self._log(core.name, result)
                    self.barrier()
                    model.train()
                total_steps += 1
        return best

    def train(self, model, loaders):
        model.train()
        core = self._unwrap(model)
        if self.rank == 0:
            os.makedirs(f"data/ckpts/{core.name}/{self.scale}", exist_ok=True)
        n = len(loaders["train"])
        optimizer, scheduler = core.get_optimizer(self, n)
        self.load(model)

        def steps(spec, as_int):
            if spec[-1] == "e":
                v = n * float(spec[:-1])
                return int(v) if as_int else round(v)
            return int(spec)
        validate_step = n if self.validate_step == "0" else steps(self.validate_step, False)
        hold_step = steps(self.hold_step, True)
        result = self._train(model, loaders, validate_step, hold_step, optimizer, scheduler=scheduler,
                             save_at_validate=self.save_at_validate)
        if self.rank in (-1, 0):
            logger.info("Best result: {}".format(result))
            self._log(core.name, result)
        return result
