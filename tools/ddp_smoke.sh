#!/bin/bash
# One-rank DDP/RCCL smoke of the training and test entry points on synthetic data
# (the multi-rank path is covered by the world-size-2 gloo tests on CPU).
# Usage: tools/ddp_smoke.sh <model>   (run from the repo root on a GPU box)
set -e
model=${1:-pcn}
work=$(mktemp -d)
python - "$model" "$work" <<'PY'
import sys, yaml
model, work = sys.argv[1], sys.argv[2]
cfg = yaml.safe_load(open("completion/cfgs/%s.yaml" % model))
cfg.update(nepoch=1, batch_size=8, work_dir=work + "/", synthetic=True, synthetic_train_shapes=2, synthetic_val_shapes=1,
           step_interval_to_print=1, eval_emd=True)
yaml.safe_dump(cfg, open(work + "/cfg.yaml", "w"))
PY
cd completion
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29517 train.py -c $work/cfg.yaml 2>&1 | tail -6
ckpt=$(find $work -name "*.pth" | head -1)
echo "checkpoint: $ckpt"
python - "$work" "$ckpt" <<'PY'
import sys, yaml
work, ckpt = sys.argv[1], sys.argv[2]
cfg = yaml.safe_load(open(work + "/cfg.yaml")); cfg["load_model"] = ckpt
yaml.safe_dump(cfg, open(work + "/cfg_test.yaml", "w"))
PY
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29518 test.py -c $work/cfg_test.yaml 2>&1 | tail -3
