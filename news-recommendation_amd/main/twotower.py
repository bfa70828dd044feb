"""python -m main.twotower [flags] — train / evaluate a two-tower recommender on MI355X.

Same entry, flags and encoder selection as the reference's main/twotower.py:8-63: one process per
GPU (mp.spawn), data-parallel gradient averaging over the "nccl" backend (RCCL over xGMI on ROCm; utils/dp.py).
"""
import torch.multiprocessing as mp
from utils.dp import FlatGradSync

from utils.manager import Manager
from models.TwoTower import TwoTowerModel
from models.modules import encoder as E

NEWS_ENCODERS = {"cnn": E.CnnNewsEncoder, "bert": E.AllBertNewsEncoder, "tfm": E.TfmNewsEncoder}
USER_ENCODERS = {"rnn": E.RnnUserEncoder, "sum": E.SumUserEncoder, "avg": E.AvgUserEncoder, "attn": E.AttnUserEncoder,
                 "tfm": E.TfmUserEncoder}


def main(rank, manager):
    manager.setup(rank)
    loaders = manager.prepare()
    if manager.newsEncoder not in NEWS_ENCODERS or manager.userEncoder not in USER_ENCODERS:
        raise ValueError(f"unknown encoder pair -ne {manager.newsEncoder} -ue {manager.userEncoder}")
    model = TwoTowerModel(manager, NEWS_ENCODERS[manager.newsEncoder](manager), USER_ENCODERS[manager.userEncoder](manager))
    model = model.to(manager.device)
    if "train" in loaders and not loaders["train"].dataset.ship_tokens:
        ds = loaders["train"].dataset           # token rows are gathered on the device from these resident tables
        model.attach_news_tokens(ds.token_ids, ds.attn_masks)
    if manager.mode == "train":
        if manager.world_size > 1:
            # every gradient is a view of one flat buffer that is all-reduced (averaged) once per step
            model = FlatGradSync(model, manager.world_size)
        return manager.train(model, loaders)
    if manager.mode == "dev":
        manager.load(model)
        return model.dev(manager, loaders, log=True)
    if manager.mode == "test":
        manager.load(model)
        return model.test(manager, loaders)
    raise ValueError(f"unknown mode {manager.mode}")


if __name__ == "__main__":
    manager = Manager({"enable_fields": ["title"], "newsEncoder": "cnn", "userEncoder": "rnn"})
    if manager.world_size > 1:
        mp.spawn(main, args=(manager,), nprocs=manager.world_size, join=True)
    else:
        main(manager.device, manager)
