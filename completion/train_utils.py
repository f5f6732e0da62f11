"""Training/eval utilities -- counterpart of the reference's
completion/train_utils.py (AverageValueMeter :3-16, set_requires_grad :19-26,
save_model :29-34), plus the helpers that replace its single-process
nn.DataParallel with one process per GPU (torch.distributed: RCCL on ROCm,
gloo on CPU)."""
import os

import torch
import torch.distributed as dist


class AttrDict(dict):
    """Attribute-access dict: stand-in for `munch`, which the reference uses to
    wrap the YAML config (train.py:200).  Missing keys read as None so the
    optional synthetic-data keys may be omitted."""

    def __getattr__(self, name):
        return self.get(name)

    def __setattr__(self, name, value):
        self[name] = value


class AverageValueMeter(object):
    """Running (optionally weighted) average; same fields as the reference's."""

    def __init__(self):
        self.reset()

    def reset(self):
        self.val = 0
        self.avg = 0
        self.sum = 0
        self.count = 0.0

    def update(self, val, n=1):
        self.val = val
        self.sum += val * n
        self.count += n
        self.avg = self.sum / self.count

    def all_reduce(self, device=None):
        """Sum (sum, count) over all ranks -- the eval loop's only collective
        (one small sum all-reduce per epoch)."""
        if is_distributed():
            t = torch.tensor([self.sum, self.count], dtype=torch.float64, device=device)
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            self.sum, self.count = float(t[0]), float(t[1])
            self.avg = self.sum / self.count if self.count else 0
        return self


def set_requires_grad(nets, requires_grad=False):
    if not isinstance(nets, list):
        nets = [nets]
    for net in nets:
        if net is not None:
            for param in net.parameters():
                param.requires_grad = requires_grad


def unwrap(net):
    """The bare module behind DistributedDataParallel / DataParallel."""
    return net.module if hasattr(net, "module") else net


def save_model(path, net, net_d=None):
    """Same checkpoint layout as the reference (`net_state_dict`
    [, `D_state_dict`]) so checkpoints are interchangeable; rank 0 writes."""
    if get_rank() != 0:
        return
    state = {'net_state_dict': unwrap(net).state_dict()}
    if net_d is not None:
        state['D_state_dict'] = unwrap(net_d).state_dict()
    torch.save(state, path)


def load_model(path, net, net_d=None, map_location="cpu"):
    ckpt = torch.load(path, map_location=map_location)
    unwrap(net).load_state_dict(ckpt['net_state_dict'])
    if net_d is not None and 'D_state_dict' in ckpt:
        unwrap(net_d).load_state_dict(ckpt['D_state_dict'])
    return ckpt


# ----------------------------------------------------------------- distributed
def is_distributed():
    return dist.is_available() and dist.is_initialized()


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def init_distributed(backend=None):
    """Join the process group described by RANK / WORLD_SIZE / MASTER_* (set by
    `python -m torch.distributed.run`).  Returns (rank, world, device).
    backend defaults to nccl (= RCCL over xGMI on ROCm) when a GPU is visible,
    gloo otherwise."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    device = torch.device("cuda", local_rank) if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not is_distributed():
        backend = backend or ("nccl" if use_gpu else "gloo")
        kwargs = {"device_id": device} if backend == "nccl" else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, world, device


def shard_indices(n, rank, world, shuffle=False, seed=0, epoch=0):
    """Split range(n) over ranks (DistributedSampler semantics): an optional
    seeded permutation, wrapped padding so every rank gets ceil(n/world)
    entries, rank r takes a contiguous slice.  Returns (indices, valid) where
    `valid` marks non-padding entries so evaluation counts every sample once."""
    order = torch.arange(n)
    if shuffle:
        order = torch.randperm(n, generator=torch.Generator().manual_seed(seed + epoch))
    per = -(-n // world)
    total = per * world
    valid = torch.ones(total, dtype=torch.bool)
    if total > n:
        order = torch.cat([order, order[: total - n]])
        valid[n:] = False
    sl = slice(rank * per, (rank + 1) * per)
    return order[sl].tolist(), valid[sl].tolist()
