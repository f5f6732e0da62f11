"""Which formulation the completion networks use where the op layer offers two -- module-level settings, set ONCE (from
the cfg's optional `op_layer:` mapping by train.py / test.py, or by a tool / test through `configure`), never read from the
environment.  Every switch selects between two ways of computing the SAME function (pinned against the reference's own run
by tests/test_model_golden.py); the defaults are the measured-faster ones (profiles/NOTES_r4.md 6).

  gather_sum           SA_module's neighbour gather + shared-weight sum in one kernel (mvp_share_gather_sum)
  gather_max           edge_preserve_sampling's neighbour gather + max in one kernel (mvp_gather_max)
  side_lanes           0: geometry (FPS chain, neighbour searches, three_nn) in line; 1: one side stream; 2: two
  stacked_projections  SA_module's conv1 / conv2 / conv3 (and a residual unit's conv1 / conv_res) as ONE convolution
  skip_full_fps_of_gt  VRCNet training: no FPS of ALL of gt's points in front of the order-blind PCN_encoder
  conv_before_interp   the way up of the U-Nets: the interpolated half convolved at the coarse level
  folded_conv          folding layers as three small products instead of tile / repeat / concatenate / convolve
  fused_activations    pre-activation ReLUs, residual sums + ReLU and per-cloud vectors inside the convolutions' GEMMs
                       (mvp_pointwise_mfma_ex), a residual unit's conv1 / conv_res as one GEMM with two outputs
  fps_beside_losses    VRCNet training: the decoder's FPS on a side lane while the main stream computes the losses of the two
                       outputs that already exist
  singleton_sk         SK_SA_module with ONE kernel (the shipped cfg): attention == 1 exactly, the fusion passes not issued
"""


class OpLayerConfig:
    __slots__ = ("gather_sum", "gather_max", "side_lanes", "stacked_projections", "skip_full_fps_of_gt",
                 "conv_before_interp", "folded_conv", "singleton_sk", "fused_activations", "fps_beside_losses")

    def __init__(self):
        self.reset()

    def reset(self):
        self.gather_sum = True
        self.gather_max = True
        self.side_lanes = 2
        self.stacked_projections = True
        self.skip_full_fps_of_gt = True
        self.conv_before_interp = True
        self.folded_conv = True
        self.singleton_sk = True
        self.fused_activations = True
        self.fps_beside_losses = True

    def as_dict(self):
        return {k: getattr(self, k) for k in self.__slots__}


OPS = OpLayerConfig()


def configure(**switches):
    """Set switches by name (unknown names raise); returns the previous values of the ones given."""
    old = {}
    for k, v in switches.items():
        if k not in OpLayerConfig.__slots__:
            raise KeyError("unknown op-layer switch %r (known: %s)" % (k, ", ".join(OpLayerConfig.__slots__)))
        old[k] = getattr(OPS, k)
        setattr(OPS, k, int(v) if k == "side_lanes" else bool(v))
    return old


def configure_from_cfg(args):
    """cfg key `op_layer: {switch: value, ...}` (optional; absent = defaults)."""
    section = args.get("op_layer") if hasattr(args, "get") else None
    OPS.reset()   # a cfg without the section (or with fewer keys) gets the defaults, not the previous cfg's switches
    if section:
        configure(**dict(section))
