"""Completion test / submission entry point -- counterpart of the reference's
completion/test.py:23-64 (`python test.py -c cfgs/<model>.yaml`): load a
checkpoint, run prefix="test", write `results.h5` (dataset `results`) next to
the checkpoint and zip it as `submission.zip`.

One process per GPU: the test set is sharded in order, every rank writes its
predictions to `results.rank<r>.npy` and rank 0 concatenates them in rank
order (no collective on the data path).  `results.h5` is written with h5py when
it is installed, else with the in-tree writer (h5lite.py: same contiguous
float32 dataset `results`, readable by h5py / libhdf5).
"""
import argparse
import logging
import os
import shutil
import subprocess
import sys
import warnings
import zipfile

import numpy as np
import torch
import torch.distributed as dist
import yaml

from dataset import build_dataset, h5_module
from train_utils import AttrDict, init_distributed, is_distributed, load_model, shard_indices, unwrap
from train import build_model

warnings.filterwarnings("ignore")


def test(args, log_dir):
    rank, world, device = init_distributed()
    dataset_test = build_dataset(args, "test")
    n = len(dataset_test)
    logging.info('Length of test dataset:%d', n)
    indices, valid = shard_indices(n, rank, world)
    per_rank = max(1, int(args.batch_size) // world)
    loader = torch.utils.data.DataLoader(torch.utils.data.Subset(dataset_test, indices),
                                         batch_size=per_rank, shuffle=False, num_workers=int(args.workers or 0))

    net = build_model(args, device, 1)   # inference only: no DDP wrapper needed
    load_model(args.load_model, net, map_location=device)
    logging.info("%s's previous weights loaded." % args.model_name)
    net.eval()

    logging.info('Testing...')
    results = []
    with torch.no_grad():
        for i, inputs_cpu in enumerate(loader):
            inputs = inputs_cpu.float().to(device).transpose(2, 1).contiguous()
            results.append(unwrap(net)(inputs, prefix="test")['result'].cpu().numpy())
            if i % args.step_interval_to_print == 0:
                logging.info('test [%d/%d]' % (i, len(loader)))
    mine = np.concatenate(results, axis=0)[np.array(valid, dtype=bool)]
    np.save(os.path.join(log_dir, 'results.rank%d.npy' % rank), mine)
    if is_distributed():
        dist.barrier()
    if rank != 0:
        return None
    parts = [np.load(os.path.join(log_dir, 'results.rank%d.npy' % r)) for r in range(world)]
    all_results = np.concatenate(parts, axis=0)[:n]
    for r in range(world):
        os.remove(os.path.join(log_dir, 'results.rank%d.npy' % r))
    out_name = 'results.h5'
    with h5_module().File(os.path.join(log_dir, out_name), 'w') as f:
        f.create_dataset('results', data=all_results)
    if shutil.which("zip"):
        subprocess.run(["zip", "-r", "submission.zip", out_name], cwd=log_dir,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    else:
        with zipfile.ZipFile(os.path.join(log_dir, "submission.zip"), "w", zipfile.ZIP_DEFLATED) as z:
            z.write(os.path.join(log_dir, out_name), out_name)
    print("Submission file has been saved to %s/submission.zip" % log_dir)
    return all_results


def main():
    parser = argparse.ArgumentParser(description='Test config file')
    parser.add_argument('-c', '--config', help='path to config file', required=True)
    arg = parser.parse_args()
    args = AttrDict(yaml.safe_load(open(arg.config)))
    import op_config
    op_config.configure_from_cfg(args)
    if not args.load_model:
        raise ValueError('Model path must be provided to load model!')
    log_dir = os.path.dirname(args.load_model)
    logging.basicConfig(level=logging.INFO, handlers=[logging.FileHandler(os.path.join(log_dir, 'test.log')),
                                                      logging.StreamHandler(sys.stdout)])
    test(args, log_dir)


if __name__ == "__main__":
    main()
