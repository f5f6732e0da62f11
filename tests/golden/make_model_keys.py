"""Record the state_dict layout (parameter names + shapes) of the REFERENCE
completion models, so tests can pin checkpoint compatibility of our models.

Run in the build container only (needs /root/reference):
    python tests/golden/make_model_keys.py
The reference models import the CUDA-only operator packages at import time
(model_utils.py:19-21) and call .cuda() in constructors; here those imports are
satisfied by empty stub modules and .cuda() is a no-op -- only __init__ runs,
no forward.  Output: tests/golden/model_state_keys.json (data: names + shapes).
"""
import json
import os
import sys
import types

import torch
import yaml

REF = "/root/reference/completion"
HERE = os.path.dirname(os.path.abspath(__file__))


class Args(dict):
    __getattr__ = dict.get


def main():
    for name, attrs in (("metrics", ["cd", "fscore", "emd"]),
                        ("mm3d_pn2", ["furthest_point_sample", "gather_points", "grouping_operation",
                                      "ball_query", "three_nn", "three_interpolate"])):
        mod = types.ModuleType(name)
        for a in attrs:
            setattr(mod, a, lambda *x, **k: None)
        sys.modules[name] = mod
    torch.Tensor.cuda = lambda self, *a, **k: self
    os.chdir(REF)
    sys.path.insert(0, REF)
    import importlib
    out = {}
    for model in ("pcn", "ecg", "vrcnet"):
        args = Args(yaml.safe_load(open(os.path.join(REF, "cfgs", model + ".yaml"))))
        net = importlib.import_module("models." + model).Model(args)
        out[model] = {k: list(v.shape) for k, v in net.state_dict().items()}
        print(model, len(out[model]), "tensors", sum(v.numel() for v in net.state_dict().values()), "elements")
    json.dump(out, open(os.path.join(HERE, "model_state_keys.json"), "w"), indent=0)

    # PCN forward (pure PyTorch end to end): seeded construction + seeded input
    # -> the reference's own output, stored as a golden vector.
    import numpy as np
    args = Args(yaml.safe_load(open(os.path.join(REF, "cfgs", "pcn.yaml"))))
    torch.manual_seed(1234)
    net = importlib.import_module("models.pcn").Model(args).eval()
    x = torch.rand(2, 3, 2048, generator=torch.Generator().manual_seed(5))
    with torch.no_grad():
        res = net(x, prefix="test")["result"]
    np.savez_compressed(os.path.join(HERE, "pcn_forward_golden.npz"), x=x.numpy(), result=res.numpy())
    print("pcn forward golden", tuple(res.shape))


if __name__ == "__main__":
    main()
