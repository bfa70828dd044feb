"""Two-rank rehearsal of the product's own multi-GPU entry on ONE GPU: `main.twotower.main(rank,
manager)` under mp.spawn — Manager.setup -> process group -> prepare (DistributedSampler) -> FlatGradSync -> >= 5 train steps
-> sharded evaluation — with NR_REHEARSE_ONE_GPU=1 (gloo instead of RCCL, both ranks on device 0).  The ranks are fresh
processes started before anything in this parent touches the GPU (reference main/twotower.py:55-61 does the same with
mp.spawn).  Mirrors reference utils/manager.py:196-208.

usage: python tests/rehearse_two_ranks.py OUTDIR      -> OUTDIR/rank{0,1}.json; exit status 0 = both ranks finished
Started by tests/conftest.py at session start of a `-m gpu` run (the pytest process itself must not have initialised the
GPU yet); tests/test_gpu_two_rank_rehearsal.py checks what it wrote.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "news-recommendation_amd")):
    if p not in sys.path:
        sys.path.insert(0, p)


def worker(rank, manager, outdir):
    import torch
    import torch.distributed as dist
    from utils.manager import Manager
    from main import twotower
    os.chdir(outdir)
    kept = {}
    orig_train = Manager.train

    def train(self, model, loaders):
        kept["model"] = model
        kept["steps"] = len(loaders["train"])
        return orig_train(self, model, loaders)
    Manager.train = train
    best = twotower.main(rank, manager)
    model = kept["model"]
    core = model.module
    flat = torch.cat([p.detach().reshape(-1) for p in core.parameters()]).cpu()
    both = [torch.empty_like(flat) for _ in range(manager.world_size)]
    dist.all_gather(both, flat)
    res = {
        "rank": rank, "backend": dist.get_backend(), "device": str(manager.device), "world": manager.world_size,
        "steps": kept["steps"], "wrapper": type(model).__name__,
        "replicas_identical": all(bool(torch.equal(both[0], b)) for b in both),
        "param_abs_sum": float(flat.double().abs().sum()),
        "n_early": int(model.n_early), "n_flat": int(model.flat.numel()),
        "ipc_env": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"),
        "best": {k: float(v) for k, v in (best or {}).items()},
        "so_loaded": any("libnewsrec_hip.so" in line for line in open("/proc/self/maps")),
    }
    json.dump(res, open(os.path.join(outdir, f"rank{rank}.json"), "w"))
    dist.barrier()
    dist.destroy_process_group()


def main():
    outdir = os.path.abspath(sys.argv[1])
    os.makedirs(outdir, exist_ok=True)
    os.environ["NR_REHEARSE_ONE_GPU"] = "1"
    os.environ.pop("HSA_ENABLE_IPC_MODE_LEGACY", None)          # Manager must put it back for its children
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    os.environ["MASTER_PORT"] = str(s.getsockname()[1])
    s.close()
    import torch.multiprocessing as mp
    from utils.manager import Manager
    from utils.synthetic import write_synthetic_caches
    os.chdir(outdir)
    m = Manager({"enable_fields": ["title"]},
                command=(f"-ws 2 -bs 8 -e 1 -ne tfm -ue tfm -hd 48 -hn 4 -lr 3e-3 -s demo --synthetic-plm "
                         f"-cr {outdir}/cache -dr {outdir}/data").split())
    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0", "Manager.__init__ must export the RCCL IPC setting for -ws > 1"
    m.news_nums = {"MINDdemo_train": 700, "MINDdemo_dev": 500}
    write_synthetic_caches(m, n_train_impr=80, n_dev_impr=30, seed=3)       # >= 80 train items / 2 ranks / batch 8: at least 5 steps each
    mp.spawn(worker, args=(m, outdir), nprocs=2, join=True)


if __name__ == "__main__":
    main()
