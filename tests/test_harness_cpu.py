"""CPU tests of the completion harness: schedules,
meters, dataset sharding, checkpoint layout, and -- with a world_size-2 gloo
process group -- the DDP train step, the sharded validation with its small sum
all-reduce, and the sharded test/submission writer.  The GPU operators are not
involved here (fake models); their parity is covered by the -m gpu tests."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

COMPLETION = os.path.join(ROOT, "completion")
if COMPLETION not in sys.path:
    sys.path.insert(0, COMPLETION)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def test_cfg_keys_match_reference_set():
    import yaml
    want = {'batch_size', 'workers', 'nepoch', 'model_name', 'load_model', 'start_epoch', 'num_points',
            'work_dir', 'flag', 'loss', 'manual_seed', 'use_mean_feature', 'step_interval_to_print',
            'epoch_interval_to_save', 'epoch_interval_to_val', 'varying_constant',
            'varying_constant_epochs', 'lr', 'lr_decay', 'lr_decay_interval', 'lr_decay_rate',
            'lr_step_decay_epochs', 'lr_step_decay_rates', 'lr_clip', 'optimizer', 'weight_decay',
            'betas', 'save_vis', 'eval_emd'}
    for name in ("pcn", "ecg", "vrcnet"):
        cfg = yaml.safe_load(open(os.path.join(COMPLETION, "cfgs", name + ".yaml")))
        assert want <= set(cfg), name
        assert cfg["model_name"] == name and cfg["batch_size"] == 32 and cfg["num_points"] == 2048
    v = yaml.safe_load(open(os.path.join(COMPLETION, "cfgs", "vrcnet.yaml")))
    assert {'layers', 'distribution_loss', 'knn_list', 'pk', 'local_folding', 'points_label',
            'num_coarse_raw', 'num_fps', 'num_coarse'} <= set(v)


def test_schedules_and_meter():
    import train
    from train_utils import AttrDict, AverageValueMeter
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))
    # varying_constant 0.01,0.1,0.5,1 at epochs 5,15,30 (train.py:101-108)
    assert [train.alpha_for_epoch(args, e) for e in (0, 4, 5, 14, 15, 29, 30, 99)] == \
        [0.01, 0.01, 0.1, 0.1, 0.5, 0.5, 1.0, 1.0]
    lr = args.lr
    seen = []
    for e in range(0, 121):
        lr = train.lr_for_epoch(args, e, lr)
        seen.append(lr)
    assert seen[0] == seen[39] == 1e-4
    assert math.isclose(seen[40], 0.7e-4) and math.isclose(seen[80], 0.49e-4) and math.isclose(seen[120], 0.343e-4)
    a = AttrDict(lr_decay=True, lr_decay_interval=None, lr_step_decay_epochs="2, 4", lr_step_decay_rates="0.5, 0.1",
                 lr_clip=1e-3)
    out, lr = [], 1.0
    for e in range(6):
        lr = train.lr_for_epoch(a, e, lr)
        out.append(lr)
    assert out == [1.0, 1.0, 0.5, 0.5, 0.05, 0.05]
    m = AverageValueMeter()
    m.update(1.0, 32)
    m.update(3.0, 8)          # weighted by batch size (train.py:172-173)
    assert math.isclose(m.avg, (32 + 24) / 40)


def test_shard_indices_cover_every_sample_once():
    from train_utils import shard_indices
    for n, world in [(10, 4), (64, 8), (7, 2), (5, 8)]:
        got = []
        for r in range(world):
            idx, valid = shard_indices(n, r, world)
            assert len(idx) == -(-n // world)
            got += [i for i, v in zip(idx, valid) if v]
        assert sorted(got) == list(range(n))
    a, _ = shard_indices(100, 0, 4, shuffle=True, seed=3, epoch=1)
    b, _ = shard_indices(100, 0, 4, shuffle=True, seed=3, epoch=2)
    assert a != b and len(set(a)) == 25


def test_synthetic_dataset_matches_mvp_layout():
    from dataset import SyntheticMVP, VIEWS_PER_SHAPE
    ds = SyntheticMVP("train", num_shapes=3, num_points=4096)
    assert len(ds) == 3 * VIEWS_PER_SHAPE == 78
    l0, p0, c0 = ds[0]
    l1, p1, c1 = ds[25]
    l2, p2, c2 = ds[26]
    assert p0.shape == (2048, 3) and c0.shape == (4096, 3) and p0.dtype == torch.float32
    assert torch.equal(c0, c1) and not torch.equal(c0, c2)        # index // 26 pairing
    assert not torch.equal(p0, p1)
    assert torch.equal(ds[0][1], p0)                               # deterministic
    t = SyntheticMVP("test", num_shapes=1)
    assert t[0].shape == (2048, 3)


def test_pcn_shapes_and_checkpoint_layout(tmp_path):
    import train
    from models import pcn
    from train_utils import load_model, save_model
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))
    net = pcn.Model(args)
    assert sum(p.numel() for p in net.parameters()) == 6861059     # 6.86 M
    keys = set(net.state_dict())
    assert {'encoder.conv1.weight', 'encoder.conv4.bias', 'decoder.fc1.weight', 'decoder.fc3.bias',
            'decoder.conv1.weight', 'decoder.conv3.bias'} <= keys
    assert not any('grid' in k for k in keys)                      # reference keeps grid out of the ckpt
    out = net(torch.rand(2, 3, 2048), prefix="test")
    assert out['result'].shape == (2, 2048, 3)
    path = str(tmp_path / "network.pth")
    save_model(path, net)
    assert set(torch.load(path)) == {'net_state_dict'}             # train_utils.py:29-34
    other = pcn.Model(args)
    load_model(path, other)
    assert all(torch.equal(a, b) for a, b in zip(net.state_dict().values(), other.state_dict().values()))
    args16 = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn_eval16k.yaml"))
    assert pcn.Model(args16)(torch.rand(1, 3, 2048), prefix="test")['result'].shape == (1, 16384, 3)


# ------------------------------------------------------------ world_size 2
class _FakeNet(torch.nn.Module):
    """Stands in for a completion model without touching the GPU ops."""

    def __init__(self):
        super().__init__()
        self.lin = torch.nn.Linear(3, 3)

    def forward(self, x, gt=None, prefix="train", alpha=None):
        pred = self.lin(x.transpose(2, 1))                        # (B, N, 3)
        if prefix == "train":
            loss = ((pred - gt[:, :pred.shape[1]]) ** 2).mean(dim=(1, 2))
            return pred, loss, loss.mean() * (alpha or 1.0)
        if prefix == "val":
            d = ((pred - gt[:, :pred.shape[1]]) ** 2).mean(dim=(1, 2))
            return {'cd_p': d.sqrt(), 'cd_t': d, 'f1': 1.0 / (1.0 + d), 'emd': 2 * d}
        return {'result': pred}


def _worker(rank, world, port, tmp, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, COMPLETION)
    import train
    from dataset import SyntheticMVP
    from train_utils import AttrDict, AverageValueMeter, init_distributed, unwrap
    torch.manual_seed(0)
    r, w, device = init_distributed("gloo")
    assert (r, w) == (rank, world) and device.type == "cpu"
    args = AttrDict(batch_size=8, workers=0)
    ds = SyntheticMVP("val", num_shapes=1, num_points=2048)        # 26 samples: uneven over 2 ranks x batch 4
    net = _FakeNet()
    ddp = torch.nn.parallel.DistributedDataParallel(net)
    # -- validation: sharded + one sum all-reduce
    loader, valid = train.make_loader(ds, args, r, w, shuffle=False)
    meters = {m: AverageValueMeter() for m in ['cd_p', 'cd_t', 'emd', 'f1']}
    best = {m: (0, 0) if m == 'f1' else (0, math.inf) for m in meters}
    res = train.val(ddp, 0, meters, loader, valid, best, device, log_dir=tmp)
    # -- one DDP training step on this rank's shard
    opt = torch.optim.SGD(unwrap(ddp).parameters(), lr=0.1)
    tl, _ = train.make_loader(ds, AttrDict(batch_size=26 * 2, workers=0), r, w, shuffle=False)
    meter = AverageValueMeter()
    train.train_one_epoch(ddp, opt, tl, device, 1.0, meter)
    q.put((rank, res, [p.detach().numpy().tolist() for p in net.parameters()], meters['cd_t'].count))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gloo_val_and_ddp_step(tmp_path):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single-process reference
    sys.path.insert(0, COMPLETION)
    import train
    from dataset import SyntheticMVP
    from train_utils import AttrDict, AverageValueMeter
    torch.manual_seed(0)
    net = _FakeNet()
    ds = SyntheticMVP("val", num_shapes=1, num_points=2048)
    loader, valid = train.make_loader(ds, AttrDict(batch_size=8, workers=0), 0, 1, shuffle=False)
    meters = {m: AverageValueMeter() for m in ['cd_p', 'cd_t', 'emd', 'f1']}
    best = {m: (0, 0) if m == 'f1' else (0, math.inf) for m in meters}
    ref = train.val(net, 0, meters, loader, valid, best, torch.device("cpu"))
    for rank, res, params, count in out:
        assert count == 26                       # every sample counted exactly once across ranks
        for k in ref:
            assert math.isclose(res[k], ref[k], rel_tol=1e-5), (k, res[k], ref[k])
    # both ranks hold identical parameters after the DDP step ...
    for a, b in zip(out[0][2], out[1][2]):
        assert a == b
    # ... equal to one full-batch step in a single process
    opt = torch.optim.SGD(net.parameters(), lr=0.1)
    tl, _ = train.make_loader(ds, AttrDict(batch_size=26, workers=0), 0, 1, shuffle=False)
    train.train_one_epoch(net, opt, tl, torch.device("cpu"), 1.0, AverageValueMeter())
    for a, b in zip(out[0][2], net.parameters()):
        assert torch.allclose(torch.tensor(a), b.detach(), rtol=1e-5, atol=1e-6)
    assert os.path.exists(os.path.join(str(tmp_path), "best_cd_t_network.pth"))   # rank 0 wrote it


class _BranchyNet(torch.nn.Module):
    """A model with a branch that never contributes to the loss -- the shape of
    MSAP_SKN_decoder at cfgs/vrcnet.yaml (num_fps == num_coarse == num_points:
    conv_s1..3, expansion2, conv_f1..2 get no gradient)."""

    def __init__(self):
        super().__init__()
        self.used = torch.nn.Linear(3, 3)
        self.unused = torch.nn.Linear(3, 3)

    def forward(self, x, gt=None, prefix="train", alpha=None):
        pred = self.used(x.transpose(2, 1))
        loss = ((pred - gt[:, :pred.shape[1]]) ** 2).mean(dim=(1, 2))
        return pred, loss, loss.mean()


def _worker_unused(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, COMPLETION)
    import train
    from dataset import SyntheticMVP
    from train_utils import AttrDict, AverageValueMeter, init_distributed, unwrap
    torch.manual_seed(0)
    r, w, device = init_distributed("gloo")
    net = train.wrap_ddp(_BranchyNet(), device, w)
    assert isinstance(net, torch.nn.parallel.DistributedDataParallel)
    opt = torch.optim.SGD(unwrap(net).parameters(), lr=0.05)
    ds = SyntheticMVP("train", num_shapes=1, num_points=2048)
    args = AttrDict(batch_size=8, workers=0)
    scale = train.loss_scale(args, w)
    assert scale == 2.0 and train.loss_scale(AttrDict(ddp_grad_scale="mean"), w) == 1.0
    loader, _ = train.make_loader(ds, args, r, w, shuffle=False)          # 13 samples per rank -> 4 steps
    train.train_one_epoch(net, opt, loader, device, 1.0, AverageValueMeter(), scale=scale)
    q.put((rank, [p.detach().numpy().tolist() for p in unwrap(net).parameters()]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_ddp_with_unused_parameters_and_sum_gradient():
    """a plain DDP wrapper dies in the second step when a branch gets
    no gradient (cfgs/vrcnet.yaml does that); train.wrap_ddp must survive several
    steps.  Also pins the gradient convention: loss * world under DDP's average ==
    the reference's sum over replicas (train.py:141 backward(ones(ngpu)))."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_unused, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for a, b in zip(out[0][1], out[1][1]):
        assert a == b
    # single-process replay of the reference's rule: per step, sum over the two replicas'
    # mean losses (each replica = one rank's batch of 4)
    import train
    from dataset import SyntheticMVP
    from train_utils import AttrDict
    torch.manual_seed(0)
    net = _BranchyNet()
    unused_before = [p.detach().clone() for p in net.unused.parameters()]
    opt = torch.optim.SGD(net.parameters(), lr=0.05)
    ds = SyntheticMVP("train", num_shapes=1, num_points=2048)
    loaders = [train.make_loader(ds, AttrDict(batch_size=8, workers=0), r, 2, shuffle=False)[0] for r in range(2)]
    for (_, x0, g0), (_, x1, g1) in zip(*loaders):
        opt.zero_grad()
        total = sum(net(x.float().transpose(2, 1).contiguous(), g.float())[2] for x, g in ((x0, g0), (x1, g1)))
        total.backward()
        opt.step()
    for a, b in zip(out[0][1], net.parameters()):
        assert torch.allclose(torch.tensor(a), b.detach(), rtol=1e-5, atol=1e-6)
    for a, b in zip(unused_before, net.unused.parameters()):
        assert torch.equal(a, b.detach())


def _worker_test_entry(rank, world, port, tmp, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                      MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    sys.path.insert(0, COMPLETION)
    import test as test_entry
    import train
    from train_utils import init_distributed
    init_distributed("gloo")
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))
    args.update(batch_size=4, data_dir=os.path.join(tmp, "data"), load_model=os.path.join(tmp, "network.pth"),
                step_interval_to_print=1000)
    res = test_entry.test(args, tmp)
    q.put((rank, None if res is None else res.shape))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_test_entry_writes_results_h5_in_dataset_order(tmp_path):
    """completion/test.py:23-64 sharded over two ranks: MVP_ExtraTest_Shuffled_CP.h5
    (written + read through the HDF5 layer) -> PCN prefix="test" -> results.h5 /
    `results` with the clouds in dataset order, padding dropped (5 clouds over 2
    ranks), submission.zip next to it."""
    import zipfile
    import h5lite
    import train
    from models import pcn
    from train_utils import save_model
    rng = np.random.default_rng(5)
    partial = rng.random((5, 2048, 3), dtype=np.float32)
    os.makedirs(tmp_path / "data")
    with h5lite.File(str(tmp_path / "data" / "MVP_ExtraTest_Shuffled_CP.h5"), "w") as f:
        f.create_dataset("incomplete_pcds", data=partial)
    args = train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))
    torch.manual_seed(11)
    net = pcn.Model(args).eval()
    save_model(str(tmp_path / "network.pth"), net)
    with torch.no_grad():
        want = net(torch.from_numpy(partial).transpose(2, 1).contiguous(), prefix="test")["result"].numpy()

    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker_test_entry, args=(r, world, port, str(tmp_path), q)) for r in range(world)]
    for p in procs:
        p.start()
    out = dict(q.get(timeout=240) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert out[0] == (5, 2048, 3) and out[1] is None
    with h5lite.File(str(tmp_path / "results.h5"), "r") as f:
        got = f["results"][()]
    assert got.dtype == np.float32
    np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-6)
    with zipfile.ZipFile(str(tmp_path / "submission.zip")) as z:
        assert z.namelist() == ["results.h5"]


def test_models_state_dict_layout_matches_reference():
    """Names and shapes of every parameter of PCN / ECG / VRCNet equal the
    reference models' (tests/golden/model_state_keys.json, recorded from the
    reference by tests/golden/make_model_keys.py) => checkpoints interchange."""
    import json
    import train
    from models import ecg, pcn, vrcnet
    ref = json.load(open(os.path.join(ROOT, "tests", "golden", "model_state_keys.json")))
    for name, mod, nparam in (("pcn", pcn, 6861059), ("ecg", ecg, 14195987), ("vrcnet", vrcnet, 17221879)):
        net = mod.Model(train.load_config(os.path.join(COMPLETION, "cfgs", name + ".yaml")))
        mine = {k: list(v.shape) for k, v in net.state_dict().items()}
        assert mine == ref[name], name
        assert sum(p.numel() for p in net.parameters()) == nparam


def test_pcn_forward_matches_reference_golden():
    """Same seed => same initial weights (identical construction order); the
    forward pass reproduces the reference PCN's output on the golden input."""
    import train
    from models import pcn
    g = np.load(os.path.join(ROOT, "tests", "golden", "pcn_forward_golden.npz"))
    torch.manual_seed(1234)
    net = pcn.Model(train.load_config(os.path.join(COMPLETION, "cfgs", "pcn.yaml"))).eval()
    with torch.no_grad():
        res = net(torch.tensor(g["x"]), prefix="test")["result"]
    np.testing.assert_allclose(res.numpy(), g["result"], rtol=1e-5, atol=1e-6)


def test_pcn_folded_conv1_equals_concatenated_formulation():
    """PCN_decoder._folded_conv1 splits conv1 over its three kinds of input channels (grid patch, coarse
    point, global feature) instead of convolving the concatenated (B, 1029, Nf) tensor the reference builds
    (pcn.py:60-68).  Same parameters, same function: compare with the concatenation written out, in float64,
    values and every gradient."""
    from models.pcn import PCN_decoder
    torch.manual_seed(5)
    B, Nc, S = 3, 8, 4
    dec = PCN_decoder(Nc, Nc * S, S, 2 + 3 + 1024).double()
    x = torch.randn(B, 1024, dtype=torch.float64, requires_grad=True)
    coarse = torch.randn(B, 3, Nc, dtype=torch.float64, requires_grad=True)
    got = dec._folded_conv1(x, coarse)
    center = coarse.unsqueeze(3).expand(-1, -1, -1, S).reshape(B, 3, Nc * S)
    grid_feat = dec.grid.detach().unsqueeze(0).repeat(B, 1, Nc)
    feat = torch.cat((grid_feat, center, x.unsqueeze(2).expand(-1, -1, Nc * S)), 1)
    ref = torch.relu(torch.nn.functional.conv1d(feat, dec.conv1.weight, dec.conv1.bias))
    assert got.shape == ref.shape == (B, 512, Nc * S)
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    params = (x, coarse, dec.conv1.weight, dec.conv1.bias)
    for a, b in zip(torch.autograd.grad(got.square().sum(), params), torch.autograd.grad(ref.square().sum(), params)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_pcn_encoder_split_conv3_equals_concatenated_formulation():
    """PCN_encoder applies conv3 to the per-point half of its input and adds the pooled half's product as one
    vector per cloud; the reference (pcn.py:25-29) tiles the pooled feature, concatenates and convolves."""
    from models.pcn import PCN_encoder
    torch.manual_seed(6)
    enc = PCN_encoder().double()
    x = torch.randn(2, 3, 50, dtype=torch.float64, requires_grad=True)
    got = enc(x)
    F = torch.nn.functional
    local = F.conv1d(torch.relu(F.conv1d(x, enc.conv1.weight, enc.conv1.bias)), enc.conv2.weight, enc.conv2.bias)
    cat = torch.cat((local, local.max(dim=2, keepdim=True)[0].expand(-1, -1, 50)), 1)
    ref = F.conv1d(torch.relu(F.conv1d(cat, enc.conv3.weight, enc.conv3.bias)), enc.conv4.weight, enc.conv4.bias).max(dim=2)[0]
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    params = (x,) + tuple(enc.parameters())
    for a, b in zip(torch.autograd.grad(got.square().sum(), params), torch.autograd.grad(ref.square().sum(), params)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_pointwise_conv_max_matches_autograd():
    """pointwise_conv_max (values, and the gather / scatter backward through the winning positions only) against
    conv(x).max over the positions under plain autograd; 1-D and (B, C, 1, N) inputs, with and without bias,
    repeated maxima included (the gradient goes to the position torch.max reports)."""
    from mvp_benchmark_amd.pointwise import PointwiseConv1d, PointwiseConv2d
    torch.manual_seed(11)
    for layer, shape in ((PointwiseConv1d(12, 20), (3, 12, 37)), (PointwiseConv2d(7, 9, bias=False), (2, 7, 1, 16)),
                         (PointwiseConv1d(5, 6), (2, 5, 8))):
        layer = layer.double()
        x = torch.randn(*shape, dtype=torch.float64)
        if shape == (2, 5, 8):
            x[..., 4:] = x[..., :4]                        # every maximum is attained twice
        x.requires_grad_()
        got = layer.max_over_positions(x)
        ref = layer(x).flatten(2).max(dim=2)[0]
        assert torch.equal(got, ref)
        go = torch.randn_like(ref)
        params = (x,) + tuple(layer.parameters())
        for a, b in zip(torch.autograd.grad(got, params, go), torch.autograd.grad(ref, params, go)):
            assert torch.allclose(a, b, rtol=1e-12, atol=1e-12)
    with torch.no_grad():
        assert torch.equal(layer.max_over_positions(x), ref)


def test_conv_global_concat_equals_concatenated_formulation():
    """models/_common.py: conv_global_concat (the global feature's share of a 1x1 convolution as one vector per
    cloud) against the convolution of the concatenated tensor the reference builds (vrcnet.py conv6, ecg.py conv5:
    global first; pcn.py conv3: global last), float64, values and every gradient, 1-D and (B, C, 1, N) layouts."""
    from models._common import conv_global_concat, pointwise1d, pointwise2d
    torch.manual_seed(9)
    F = torch.nn.functional
    for make, shape, first in ((pointwise1d, (3, 10, 17), True), (pointwise1d, (2, 6, 9), False), (pointwise2d, (2, 8, 1, 12), True)):
        B, cf, n, cg = shape[0], shape[1], shape[-1], 5
        conv = make(cg + cf, 7).double()
        g = torch.randn(B, cg, dtype=torch.float64, requires_grad=True)
        f = torch.randn(*shape, dtype=torch.float64, requires_grad=True)
        tiled = g.view(B, cg, *([1] * (len(shape) - 2))).expand(B, cg, *shape[2:])
        cat = torch.cat((tiled, f) if first else (f, tiled), 1)
        for relu in (False, True):
            got = conv_global_concat(conv, g, f, relu=relu, global_first=first)
            ref = conv(cat)
            ref = torch.relu(ref) if relu else ref
            assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
            params = (g, f) + tuple(conv.parameters())
            for a, b in zip(torch.autograd.grad(got.square().sum(), params), torch.autograd.grad(ref.square().sum(), params)):
                assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_vrcnet_folding_equals_concatenated_formulation():
    """models/vrcnet.py Folding (conv_folded_concat: global feature, repeated point feature and tiled grid as three tiny
    products) against the reference's formulation written out (vrcnet.py:60-75: tile / repeat / concatenate, one
    convolution, ReLU), float64, values and every gradient."""
    from models.vrcnet import Folding
    torch.manual_seed(12)
    B, C, Nc, S = 2, 6, 5, 4
    fold = Folding(C, 7, S, global_feature_size=9).double()
    pf = torch.randn(B, C, Nc, dtype=torch.float64, requires_grad=True)
    gf = torch.randn(B, 9, dtype=torch.float64, requires_grad=True)
    got = fold(pf, gf)
    total = Nc * S
    point = pf.unsqueeze(3).expand(-1, -1, -1, S).reshape(B, C, total)
    glob = gf.unsqueeze(2).expand(-1, -1, total)
    grid = fold.grid.double().unsqueeze(0).repeat(B, Nc, 1).transpose(1, 2)
    ref = torch.relu(torch.nn.functional.conv1d(torch.cat([glob, point, grid], dim=1), fold.conv.weight, fold.conv.bias))
    assert got.shape == ref.shape == (B, 7, total)
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    params = (pf, gf, fold.conv.weight, fold.conv.bias)
    for a, b in zip(torch.autograd.grad(got.square().sum(), params), torch.autograd.grad(ref.square().sum(), params)):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_sa_module_equals_gather_then_map_formulation():
    """SA_module maps the points with conv2 / conv3 before gathering the
    neighbours; the reference (vrcnet.py:36-57) gathers first.  Same parameters,
    same function: compare against the gather-first formulation written out."""
    from model_utils import get_edge_features
    from models.relational import SA_module
    torch.manual_seed(3)
    B, C, N, k, share = 2, 32, 40, 6, 8
    sam = SA_module(C, C // 16, C // 4, C, share_planes=share, k=k).double()
    x = torch.randn(B, C, 1, N, dtype=torch.float64, requires_grad=True)
    idx = torch.randint(0, N, (B, N, k))
    out, _ = sam([x, idx])

    act = torch.relu(x)
    nbr = get_edge_features(act, idx)                               # (B, C, k, N)
    query = sam.conv1(act)
    keys = sam.conv2(nbr).reshape(B, -1, 1, N)
    values = sam.conv3(nbr)
    w = sam.conv_w(torch.cat([query, keys], 1)).view(B, -1, k, N).repeat(1, share, 1, 1)
    ref = sam.conv_out(torch.relu((w * values).sum(dim=2, keepdim=True))) + x
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-10)
    g1, = torch.autograd.grad(out.square().sum(), x, retain_graph=True)
    g2, = torch.autograd.grad(ref.square().sum(), x)
    assert torch.allclose(g1, g2, rtol=1e-9, atol=1e-9)


def test_singleton_sk_unit_is_bit_identical_to_the_attention_formulation():
    """SKN_Res_unit with ONE neighbourhood size (cfgs/vrcnet.yaml: knn_list "16"): the softmax over a stack of one kernel
    is identically 1, so op_config's `singleton_sk` route (no stack / sums / means / product, no second ReLU) must give
    the reference formulation's (vrcnet.py:138-173) output, input gradient and every parameter gradient BIT FOR BIT --
    exact zeros for the squeeze-excite layers included."""
    import op_config
    from models.relational import SKN_Res_unit
    torch.manual_seed(11)
    B, C, N, k = 2, 32, 48, 5
    unit = SKN_Res_unit(16, C, k=[k], layers=2)
    x0 = torch.randn(B, 16, 1, N)
    idx = [torch.randint(0, N, (B, N, k))]
    runs = {}
    for on in (False, True):
        old = op_config.configure(singleton_sk=on)
        try:
            unit.zero_grad()
            x = x0.clone().requires_grad_()
            out = unit(x, idx)
            out.square().sum().backward()
            runs[on] = (out.detach().clone(), x.grad.clone(), {n: p.grad.clone() for n, p in unit.named_parameters()})
        finally:
            op_config.configure(**old)
    assert torch.equal(runs[True][0], runs[False][0])
    assert torch.equal(runs[True][1], runs[False][1])
    assert runs[True][2].keys() == runs[False][2].keys()
    for name, grad in runs[False][2].items():
        assert torch.equal(runs[True][2][name], grad), name
        if ".fc." in name or ".fcs." in name:
            assert not grad.any(), name          # the reference's autograd hands these layers exact zeros


@pytest.mark.parametrize("dense_n", [1, 3])
def test_dense_conv_equals_edge_tensor_formulation(dense_n):
    """Dense_conv applies the centre columns of every layer per point and
    first_conv's neighbour columns before the gather; the reference
    (ecg.py:36-65) runs every layer on the materialised (B, ., N, k) edge
    tensor.  Same parameters, same function."""
    import torch.nn.functional as F
    from model_utils import get_graph_feature
    from models.edge_unet import Dense_conv
    torch.manual_seed(5)
    B, C, N, k = 2, 12, 50, 5
    dc = Dense_conv(C, growth_rate=8, dense_n=dense_n, k=k).double()
    x = torch.randn(B, C, N, dtype=torch.float64, requires_grad=True)
    out = dc(x)
    assert out.shape == (B, C + 8 * dense_n, N)

    edge = F.relu(dc.first_conv(get_graph_feature(x, k=k)))
    edge = torch.cat((edge, x.unsqueeze(3).expand(-1, -1, -1, k)), 1)
    ref = dc.model(edge).max(dim=3)[0]
    assert torch.allclose(out, ref, rtol=1e-10, atol=1e-10)
    g1, = torch.autograd.grad(out.square().sum(), x, retain_graph=True)
    g2, = torch.autograd.grad(ref.square().sum(), x, retain_graph=True)
    assert torch.allclose(g1, g2, rtol=1e-9, atol=1e-9)
    p1 = torch.autograd.grad(out.square().sum(), list(dc.parameters()))
    p2 = torch.autograd.grad(ref.square().sum(), list(dc.parameters()))
    for a, b in zip(p1, p2):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("step", [1, 3])
def test_ef_expansion_equals_edge_tensor_formulation(step):
    """EF_expansion applies the centre / neighbour columns of conv1 and conv2
    per point and gathers the mapped channels; the reference
    (model_utils.py:26-55) maps the materialised (B, 2C, k, N) edge tensor.
    Same parameters, same function (outputs and all gradients, float64)."""
    import torch.nn.functional as F
    from model_utils import EF_expansion, get_graph_feature
    torch.manual_seed(11)
    B, C, N, k, out = 2, 10, 37, 4, 6
    ef = EF_expansion(C, output_size=out, step_ratio=step, k=k).double()
    x = torch.randn(B, C, N, dtype=torch.float64, requires_grad=True)
    got = ef(x)
    assert got.shape == (B, out, N * step)

    edge_in = get_graph_feature(x, k, minus_center=False).permute(0, 1, 3, 2).contiguous()
    edge = F.relu(torch.cat((ef.conv1(edge_in), edge_in), 1))
    edge = F.relu(ef.conv2(edge))
    edge = edge.permute(0, 2, 3, 1).contiguous().view(B, k, N * step, out).permute(0, 3, 1, 2)
    ref = ef.conv3(edge).max(dim=2)[0]
    assert torch.allclose(got, ref, rtol=1e-10, atol=1e-10)
    params = [x] + list(ef.parameters())
    g1 = torch.autograd.grad(got.square().sum(), params)
    g2 = torch.autograd.grad(ref.square().sum(), params)
    for a, b in zip(g1, g2):
        assert torch.allclose(a, b, rtol=1e-9, atol=1e-9)


def test_uniform_loss_second_neighbour_equals_topk():
    """get_uniform_loss takes the second largest entry per row of the negative
    distance matrix with two max reductions; the reference calls
    knn_point(2, g, g) (model_utils.py:201-227, 250-259).  Same values, duplicates
    (ball_query pads with the first hit) included."""
    from model_utils import knn_point
    torch.manual_seed(4)
    g = torch.rand(50, 12, 3)
    g[:, 7:] = g[:, :1]                      # padded slots repeat the first point
    g[3, :] = g[3, :1]                       # a ball with a single distinct point
    want = knn_point(2, g, g)[0][:, :, 1:]
    inner = -2 * torch.matmul(g, g.transpose(2, 1))
    sq = (g * g).sum(dim=2)
    pairwise = -sq.unsqueeze(2) - inner - sq.unsqueeze(1)
    first = pairwise.argmax(dim=-1, keepdim=True)
    got = pairwise.scatter(-1, first, float('-inf')).max(dim=-1, keepdim=True)[0]
    assert torch.equal(got, want)


def test_geometry_ahead_runs_in_line_without_a_gpu():
    """model_utils.GeometryAhead on the CPU: no side stream, run() evaluates at once (without autograd), take() hands
    the stored value back -- the encoders' coordinate-only work is scheduled the same way with or without a GPU."""
    import sys
    sys.path.insert(0, os.path.join(ROOT, "completion"))
    import torch
    from model_utils import GeometryAhead
    geo = GeometryAhead(torch.device("cpu"))
    assert geo.side is None
    x = torch.arange(6.0, requires_grad=True)
    a = geo.run("a", lambda: (x * 2, [x + 1, x + 2]))
    assert not a[0].requires_grad and torch.equal(a[0], torch.arange(6.0) * 2)
    got = geo.take("a")
    assert got is a and torch.equal(got[1][1], torch.arange(6.0) + 2)
