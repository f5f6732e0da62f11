"""Completion training / validation entry point -- counterpart of the
reference's completion/train.py (`python train.py -c cfgs/<model>.yaml`, run
from completion/; same cfg keys, metric names, checkpoint layout, LR / alpha
schedules).

What changed, MI355X-first: the reference wraps the model in single-process
`torch.nn.DataParallel` (train.py:49) and back-propagates a per-replica loss
vector (`net_loss.backward(ones(ngpu))`, :141).  Here every GPU is its own
process (`python -m torch.distributed.run --nproc-per-node N train.py -c ...`):
the batch dimension is sharded across ranks, the model is wrapped in
DistributedDataParallel (bucketed gradient all-reduce over RCCL/xGMI overlapped
with backward), each rank back-propagates its scalar loss, and the validation
meters are combined with one small sum all-reduce.  `batch_size` in the cfg
stays the GLOBAL batch (the reference's DataParallel also splits it across
GPUs); each rank takes batch_size / world samples per step.
"""
import argparse
import datetime
import importlib
import logging
import math
import os
import random
import sys
import warnings

import torch
import torch.optim as optim
import yaml

from dataset import build_dataset
from model_utils import check_emd_status
from train_utils import (AttrDict, AverageValueMeter, get_rank, get_world_size, init_distributed,
                         load_model, save_model, shard_indices, unwrap)

warnings.filterwarnings("ignore")


def _floats(text):
    return [float(v.strip()) for v in str(text).split(',')]


def _ints(text):
    return [int(v.strip()) for v in str(text).split(',')]


def alpha_for_epoch(args, epoch):
    """varying_constant schedule (train.py:101-108)."""
    if not args.varying_constant:
        return None
    epochs, values = _ints(args.varying_constant_epochs), _floats(args.varying_constant)
    assert len(values) == len(epochs) + 1
    for ind, ep in enumerate(epochs):
        if epoch < ep:
            return values[ind]
    return values[-1]


def lr_for_epoch(args, epoch, lr):
    """Manual LR decay (train.py:110-120); returns the LR to use in `epoch`."""
    if not args.lr_decay:
        return lr
    if args.lr_decay_interval and args.lr_step_decay_epochs:
        raise ValueError('lr_decay_interval and lr_step_decay_epochs are mutually exclusive!')
    if args.lr_decay_interval:
        if epoch > 0 and epoch % args.lr_decay_interval == 0:
            lr = lr * args.lr_decay_rate
    elif args.lr_step_decay_epochs:
        decay_epochs, decay_rates = _ints(args.lr_step_decay_epochs), _floats(args.lr_step_decay_rates)
        if epoch in decay_epochs:
            lr = lr * decay_rates[decay_epochs.index(epoch)]
    if args.lr_clip:
        lr = max(lr, args.lr_clip)
    return lr


def make_loader(dataset, args, rank, world, shuffle, epoch=0, seed=0):
    """This rank's shard of the dataset, batch_size/world samples per step.
    Returns (loader, valid flags per sample of the shard)."""
    indices, valid = shard_indices(len(dataset), rank, world, shuffle=shuffle, seed=seed, epoch=epoch)
    per_rank = max(1, int(args.batch_size) // world)
    subset = torch.utils.data.Subset(dataset, indices)
    loader = torch.utils.data.DataLoader(subset, batch_size=per_rank, shuffle=False,
                                         num_workers=int(args.workers or 0))
    return loader, valid


def _needs_unused_parameter_walk(net, args):
    """True unless the cfg says `ddp_find_unused_parameters: False`.
    cfgs/vrcnet.yaml has num_fps == num_coarse == num_points, so MSAP_SKN_decoder never runs conv_s1..3 /
    expansion2 / conv_f1..2: a plain DDP raises "Expected to have finished reduction" in the second step there.
    cfgs/pcn.yaml and cfgs/ecg.yaml use every parameter in every step and switch the per-step graph walk off
    themselves; a cfg variant that does not say so keeps it (the reference's nn.DataParallel, train.py:49,
    tolerates unused branches)."""
    forced = args.get("ddp_find_unused_parameters") if args is not None else None
    return True if forced is None else bool(forced)


def wrap_ddp(net, device, world, args=None):
    """DistributedDataParallel around `net` when world > 1.

    find_unused_parameters only where a configuration leaves whole branches without a gradient
    (see above; unknown models: on): the reference's nn.DataParallel (train.py:49) tolerates
    that, a plain DDP raises "Expected to have finished reduction in the prior iteration" in the
    second step; for models that use every parameter the per-step graph walk is skipped."""
    if world > 1:
        ids = [device.index] if device.type == "cuda" else None
        net = torch.nn.parallel.DistributedDataParallel(net, device_ids=ids,
                                                        find_unused_parameters=_needs_unused_parameter_walk(net, args))
    return net


def loss_scale(args, world):
    """Factor applied to the rank's scalar loss before backward().

    The reference back-propagates the SUM of the per-replica mean losses
    (`net_loss.backward(ones(ngpu))`, train.py:141): its gradient is ngpu times
    the gradient of the global-batch mean.  DDP averages over ranks, so the same
    gradient needs loss * world -- the default (`ddp_grad_scale: sum`, matters
    for Adam's eps and for weight_decay).  `ddp_grad_scale: mean` keeps DDP's
    own convention (gradient of the global-batch mean)."""
    mode = args.get("ddp_grad_scale") or "sum"
    if mode not in ("sum", "mean"):
        raise ValueError("ddp_grad_scale must be 'sum' or 'mean'")
    return float(world) if mode == "sum" else 1.0


def build_model(args, device, world):
    model_module = importlib.import_module('.%s' % args.model_name, 'models')
    net = model_module.Model(args).to(device)
    if hasattr(model_module, 'weights_init'):
        net.apply(model_module.weights_init)
    return wrap_ddp(net, device, world, args)


class GraphedStep:
    """The training step (forward, loss, backward, optimizer) captured into ONE HIP graph and
    replayed per batch -- opt-in (`hip_graph: True` in the cfg), one rank, Adam / AdamW only
    (built `capturable`, learning rate kept in a device tensor).  Inputs are copied into static
    buffers, `alpha` lives in a device scalar; a batch of another shape (the last, short one) runs
    eagerly.  The step itself is the one of train_one_epoch (train.py:122-142 of the reference).
    Measured on one MI355X (profiles/r3_vrcnet_step_4cell.txt): VRCNet 32.7 ms eager / 32.6 ms
    replayed, ECG 24.8 / 23.0 -- the step is GPU-bound, so this stays an option, not the default."""

    def __init__(self, net, optimizer, device):
        self.net, self.opt, self.device = net, optimizer, device
        self.graph = None
        self.shape = None
        for group in optimizer.param_groups:       # tensor lr: changes between replays without a re-capture
            if not torch.is_tensor(group['lr']):
                group['lr'] = torch.tensor(float(group['lr']), device=device)

    def set_lr(self, lr):
        for group in self.opt.param_groups:
            group['lr'].fill_(float(lr))

    def _step(self):
        self.opt.zero_grad(set_to_none=True)
        out2, loss2, net_loss = self.net(self.inputs, self.gt, alpha=self.alpha)
        net_loss = net_loss.mean()
        net_loss.backward()
        self.opt.step()
        return loss2.mean(), net_loss

    def _capture(self, inputs, gt, alpha):
        self.inputs, self.gt = inputs.clone(), gt.clone()
        self.alpha = torch.tensor(float(alpha), device=self.device)
        # The three warm-up steps a capture needs are real optimizer steps.  So that the first batch is
        # trained on ONCE (by the first replay), as in the eager loop, parameters, buffers (BatchNorm
        # statistics), optimizer state and the random streams are put back afterwards -- in place: the
        # graph records the addresses of the tensors the warm-up created.
        model_snap = {k: v.detach().clone() for k, v in self.net.state_dict().items()}
        opt_snap = {p: {k: v.detach().clone() for k, v in st.items() if torch.is_tensor(v)}
                    for p, st in self.opt.state.items()}
        rng_cpu, rng_dev = torch.get_rng_state(), torch.cuda.get_rng_state(self.device)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):              # warm-up steps off the capturing stream
            for _ in range(3):
                self._step()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        with torch.no_grad():
            for k, v in self.net.state_dict().items():
                v.copy_(model_snap[k])
            for p, st in self.opt.state.items():
                for k, v in st.items():
                    if torch.is_tensor(v):         # state the warm-up created starts from zero (Adam / AdamW: moments, step)
                        v.copy_(opt_snap[p][k]) if p in opt_snap and k in opt_snap[p] else v.zero_()
        torch.set_rng_state(rng_cpu)
        torch.cuda.set_rng_state(rng_dev, self.device)
        logging.info('hip_graph: three warm-up steps run and rolled back (parameters, buffers, optimizer state, RNG)')
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.fine, self.total = self._step()
        self.shape = (tuple(inputs.shape), tuple(gt.shape))

    def __call__(self, inputs, gt, alpha):
        """-> (mean fine loss, total loss) as device scalars of THIS batch's step."""
        if self.graph is None:
            self._capture(inputs, gt, alpha)       # (warm-up steps on this batch, rolled back; then the capture)
        if (tuple(inputs.shape), tuple(gt.shape)) != self.shape:
            self.opt.zero_grad(set_to_none=True)
            out2, loss2, net_loss = self.net(inputs, gt, alpha=torch.tensor(float(alpha), device=self.device))
            net_loss = net_loss.mean()
            net_loss.backward()
            self.opt.step()
            return loss2.mean(), net_loss
        self.inputs.copy_(inputs)
        self.gt.copy_(gt)
        self.alpha.fill_(float(alpha))
        self.graph.replay()
        return self.fine, self.total


def train_one_epoch(net, optimizer, loader, device, alpha, meter, log_fn=None, scale=1.0, graphed=None):
    unwrap(net).train()
    for i, data in enumerate(loader):
        _, inputs, gt = data
        inputs = inputs.float().to(device).transpose(2, 1).contiguous()
        gt = gt.float().to(device)
        if graphed is not None:
            fine, net_loss = graphed(inputs, gt, alpha)
            meter.update(net_loss.item())
            if log_fn:
                log_fn(i, fine.item(), net_loss.item())
            continue
        optimizer.zero_grad()
        out2, loss2, net_loss = net(inputs, gt, alpha=alpha)
        net_loss = net_loss.mean()
        # DDP all-reduces (averages) the gradients; `scale` = loss_scale() restores the
        # reference's sum-over-replicas gradient
        (net_loss * scale if scale != 1.0 else net_loss).backward()
        optimizer.step()
        meter.update(net_loss.item())
        if log_fn:
            log_fn(i, loss2.mean().item(), net_loss.item())


def val(net, curr_epoch_num, val_loss_meters, loader, valid, best_epoch_losses, device, log_dir=None):
    """Validation pass (train.py:156-192): per-batch means weighted by batch
    size, summed over ranks; best-so-far checkpoints per metric."""
    logging.info('Testing...')
    for v in val_loss_meters.values():
        v.reset()
    unwrap(net).eval()
    model = unwrap(net)      # no gradient sync needed in eval
    seen = 0
    with torch.no_grad():
        for data in loader:
            label, inputs, gt = data
            curr = gt.shape[0]
            keep = torch.tensor(valid[seen:seen + curr], dtype=torch.bool)
            seen += curr
            inputs = inputs.float().to(device).transpose(2, 1).contiguous()
            gt = gt.float().to(device)
            result_dict = model(inputs, gt, prefix="val")
            n_keep = int(keep.sum())
            if n_keep == 0:
                continue
            for k, v in val_loss_meters.items():
                r = result_dict[k]
                r = r[keep.to(r.device)] if torch.is_tensor(r) and r.dim() > 0 else r
                v.update(float(r.mean()) if torch.is_tensor(r) else float(r), n_keep)
    # a failed auction launch is an exception here, not a NaN in the log -- on EVERY rank: a rank that raised alone
    # would leave the others waiting in the meters' all-reduce until the collective times out
    failure = None
    try:
        check_emd_status()
    except Exception as e:   # noqa: BLE001 (re-raised below, on all ranks)
        failure = e
    if get_world_size() > 1:
        flag = torch.tensor([1.0 if failure is not None else 0.0], device=device)
        torch.distributed.all_reduce(flag, op=torch.distributed.ReduceOp.MAX)
        if flag.item() > 0 and failure is None:
            failure = RuntimeError("the EMD status check failed on another rank")
    if failure is not None:
        raise failure
    for v in val_loss_meters.values():
        v.all_reduce(device)

    fmt = 'best_%s: %f [epoch %d]; '
    best_log = ''
    for loss_type, (curr_best_epoch, curr_best_loss) in best_epoch_losses.items():
        avg = val_loss_meters[loss_type].avg
        better = avg > curr_best_loss if loss_type == 'f1' else avg < curr_best_loss
        if better:
            best_epoch_losses[loss_type] = (curr_epoch_num, avg)
            if log_dir:
                save_model('%s/best_%s_network.pth' % (log_dir, loss_type), net)
            logging.info('Best %s net saved!' % loss_type)
            best_log += fmt % (loss_type, avg, curr_epoch_num)
        else:
            best_log += fmt % (loss_type, curr_best_loss, curr_best_epoch)
    curr_log = ''.join('curr_%s: %f; ' % (k, m.avg) for k, m in val_loss_meters.items())
    logging.info(curr_log)
    logging.info(best_log)
    return {k: m.avg for k, m in val_loss_meters.items()}


def train(args, log_dir, exp_name):
    rank, world, device = init_distributed()
    scale = loss_scale(args, world)
    logging.info(str(args))
    # the multi-GPU gradient convention, once, where it can be found again (train.log)
    logging.info('ranks: %d, loss scale before backward: %g (ddp_grad_scale: %s -> gradient = %s), '
                 'DDP find_unused_parameters: %s', world, scale, args.get("ddp_grad_scale") or "sum",
                 "sum over ranks of the per-rank mean loss, as the reference's DataParallel" if scale != 1.0 or world == 1
                 else "global-batch mean", _needs_unused_parameter_walk(None, args) if world > 1 else "n/a (one rank)")
    metrics = ['cd_p', 'cd_t', 'emd', 'f1'] if args.eval_emd else ['cd_p', 'cd_t', 'f1']
    best_epoch_losses = {m: (0, 0) if m == 'f1' else (0, math.inf) for m in metrics}
    train_loss_meter = AverageValueMeter()
    val_loss_meters = {m: AverageValueMeter() for m in metrics}

    dataset = build_dataset(args, "train")
    dataset_test = build_dataset(args, "val")
    logging.info('Length of train dataset:%d', len(dataset))
    logging.info('Length of test dataset:%d', len(dataset_test))

    seed = int(args.manual_seed) if args.manual_seed else random.randint(1, 10000)
    logging.info('Random Seed: %d' % seed)
    random.seed(seed)
    torch.manual_seed(seed)               # identical initial weights on every rank
    net = build_model(args, device, world)
    torch.manual_seed(seed + 1000 * (rank + 1))   # per-rank streams for dropout / rsample

    lr = args.lr
    opt_cls = getattr(optim, args.optimizer)
    params = unwrap(net).parameters()
    if args.get("hip_graph"):
        # (checked before the optimizer is chosen: Adagrad keeps its step counter on the host -- captured, it would be
        # baked into the graph; the graphed step applies no loss scale)
        if world != 1 or args.optimizer not in ('Adam', 'AdamW') or device.type != "cuda" or scale != 1.0:
            raise ValueError("hip_graph: True needs one rank on a GPU, the Adam / AdamW optimizer and a loss scale of 1")
    if args.optimizer == 'Adagrad':
        optimizer = opt_cls(params, lr=lr, initial_accumulator_value=args.initial_accum_val)
    else:
        betas = tuple(_floats(args.betas))
        extra = {"capturable": True} if args.get("hip_graph") else {}
        if device.type == "cuda" and args.optimizer in ('Adam', 'AdamW') and args.get("fused_optimizer", True):
            # one kernel over {parameter, gradient, both moments} instead of the foreach implementation's ~8 passes (same
            # update rule; VRCNet's 17 M parameters: 1.0 -> 0.3 ms per step).  `fused_optimizer: False` in the cfg restores it.
            extra["fused"] = True
        optimizer = opt_cls(params, lr=lr, weight_decay=args.weight_decay, betas=betas, **extra)

    if args.load_model:
        load_model(args.load_model, net, map_location=device)
        logging.info("%s's previous weights loaded." % args.model_name)

    loader_test, valid_test = make_loader(dataset_test, args, rank, world, shuffle=False)
    graphed = GraphedStep(net, optimizer, device) if args.get("hip_graph") else None
    if graphed is not None:
        logging.info('training step captured into one HIP graph (hip_graph: True)')
    last = None
    for epoch in range(args.start_epoch, args.nepoch):
        train_loss_meter.reset()
        alpha = alpha_for_epoch(args, epoch)
        lr = lr_for_epoch(args, epoch, lr)
        if graphed is not None:
            graphed.set_lr(lr)
        else:
            for group in optimizer.param_groups:
                group['lr'] = lr
        loader, _ = make_loader(dataset, args, rank, world, shuffle=True, epoch=epoch, seed=seed)

        def log_fn(i, fine, total, _epoch=epoch, _lr=lr, _alpha=alpha):
            if i % args.step_interval_to_print == 0:
                logging.info(exp_name + ' train [%d: %d/%d]  loss_type: %s, fine_loss: %f total_loss: %f lr: %f'
                             % (_epoch, i, len(dataset) / args.batch_size, args.loss, fine, total, _lr)
                             + ' alpha: ' + str(_alpha))

        train_one_epoch(net, optimizer, loader, device, alpha, train_loss_meter, log_fn, scale, graphed)

        if epoch % args.epoch_interval_to_save == 0:
            save_model('%s/network.pth' % log_dir, net)
            logging.info("Saving net...")
        if epoch % args.epoch_interval_to_val == 0 or epoch == args.nepoch - 1:
            last = val(net, epoch, val_loss_meters, loader_test, valid_test, best_epoch_losses, device, log_dir)
    return last


def load_config(path):
    """yaml -> attribute dict; the optional `op_layer:` mapping sets the op-layer switches (op_config.py)."""
    args = AttrDict(yaml.safe_load(open(path)))
    import op_config
    op_config.configure_from_cfg(args)
    return args


def main():
    parser = argparse.ArgumentParser(description='Train config file')
    parser.add_argument('-c', '--config', help='path to config file', required=True)
    arg = parser.parse_args()
    args = load_config(arg.config)

    # one experiment name for the whole job: rank 0's clock, broadcast (ranks started
    # across a second boundary would otherwise create different log directories)
    rank, world, _ = init_distributed()
    stamp = [datetime.datetime.now().isoformat()[:19]]
    if world > 1:
        torch.distributed.broadcast_object_list(stamp, src=0)
    time = stamp[0]
    if args.load_model:
        exp_name = os.path.basename(os.path.dirname(args.load_model))
        log_dir = os.path.dirname(args.load_model)
    else:
        exp_name = args.model_name + '_' + args.loss + '_' + args.flag + '_' + time
        log_dir = os.path.join(args.work_dir, exp_name)
    os.makedirs(log_dir, exist_ok=True)
    handlers = [logging.StreamHandler(sys.stdout)]
    if rank == 0:
        handlers.append(logging.FileHandler(os.path.join(log_dir, 'train.log')))
    logging.basicConfig(level=logging.INFO if rank == 0 else logging.WARNING, handlers=handlers)
    train(args, log_dir, exp_name)


if __name__ == "__main__":
    main()
