"""ctypes binding of libmvpops.so (the C ABI declared in include/mvpops.h).

This is the binding a maintainer of the reference would write in place of the
8 pybind11 modules (chamfer_3D, emd, furthest_point_sample_ext, ...): raw
device pointers + int sizes + the current HIP stream.  See INTEGRATION.md.
"""
import ctypes
import os

import torch  # noqa: F401  (maps torch's libamdhip64.so.7 before ours resolves it)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmvpops.so")

MVP_OK = 0
_ERR = {-1: "MVP_EBADSHAPE (bad shape / size guard)",
        -2: "MVP_EBADARG (null or undersized buffer)",
        -3: "MVP_ELAUNCH (HIP launch error)"}

# name -> argument kinds: p = device pointer, i = int, f = float, q = int64
SIGNATURES = {
    "mvp_chamfer_forward": "iiipppppp",
    "mvp_chamfer_forward_sorted": "iiipppppppq",
    "mvp_chamfer_backward": "iiipppppppp",
    "mvp_emd_forward": "iippppfipq",
    "mvp_emd_forward_plan": "iippppfipqp",
    "mvp_emd_backward": "iippppp",
    "mvp_furthest_point_sampling": "iiippp",
    "mvp_furthest_point_sampling_sorted": "iiippppq",
    "mvp_furthest_point_sampling_with_dist": "iiippp",
    "mvp_ball_query": "iiiffippp",
    "mvp_knn": "iiiipppp",
    "mvp_knn_sorted": "iiiipppppq",
    "mvp_topk_gram": "iiippp",
    "mvp_three_nn": "iiipppp",
    "mvp_three_interpolate": "iiiipppp",
    "mvp_three_interpolate_grad": "iiiipppp",
    "mvp_gather_points": "iiiippp",
    "mvp_gather_points_grad": "iiiippp",
    "mvp_gather_max": "iiiiipppp",
    "mvp_furthest_point_sampling_cluster": "iiiippppq",
    "mvp_gather_max_grad": "iiiipppi",
    "mvp_group_points": "iiiiippp",
    "mvp_group_points_grad": "iiiiippp",
    "mvp_gather_points_grad_ws": "iiiippppqi",
    "mvp_group_points_grad_ws": "iiiiippppqi",
    "mvp_three_interpolate_grad_ws": "iiiipppppqi",
    "mvp_share_weighted_sum": "iiiiippp",
    "mvp_share_weighted_sum_grad": "iiiiippppp",
    "mvp_share_gather_sum": "iiiiiipppp",
    "mvp_share_gather_sum_grad": "iiiiiipppppp",
    "mvp_pointwise_wgrad": "iiiipppppq",
    "mvp_pointwise_dgrad": "iiiippp",
    "mvp_kabsch_svd3": "ipppppp",
    "mvp_pointwise_mfma": "iiiipppiippiip",
    "mvp_pointwise_mfma_ex": "iiiipppiipipiipip",
    "mvp_pointwise_wgrad_mfma": "iiiippppppq",
    "mvp_pointwise_wgrad_mfma_ex": "iiiipipppppq",
    "mvp_pointwise_max_backward": "iiiippppppppq",
    "mvp_pointwise_mfma_max": "iiiippipipppq",
}
_CT = {"p": ctypes.c_void_p, "i": ctypes.c_int, "f": ctypes.c_float,
       "q": ctypes.c_longlong}

ABI_VERSION = 18  # MVP_ABI_VERSION of include/mvpops.h this binding was written against

# default of mvp_emd_configure's `split` knob (csrc/emd.hip: emd_knobs)
EMD_DEFAULT_SPLIT = 5

_lib = None


class MvpOpsError(RuntimeError):
    pass


def load():
    """Load libmvpops.so; raise loudly if the HIP extension is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise MvpOpsError(
            "libmvpops.so not found at %s -- build it with "
            "`python -c 'import __graft_entry__ as g; g.build()'` or "
            "`make -C mvp_benchmark_amd/csrc`. There is no CPU fallback."
            % LIB_PATH)
    lib = ctypes.CDLL(LIB_PATH)
    lib.mvp_abi_version.restype = ctypes.c_int
    lib.mvp_last_hip_error.restype = ctypes.c_char_p
    lib.mvp_emd_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_emd_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.mvp_emd_configure.restype = ctypes.c_int
    lib.mvp_emd_configure.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mvp_fps_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_fps_cluster_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_fps_cluster_scratch_bytes.argtypes = [ctypes.c_int]
    lib.mvp_fps_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int]
    lib.mvp_chamfer_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_chamfer_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mvp_knn_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_knn_scratch_bytes.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int]
    lib.mvp_scatter_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_scatter_scratch_bytes.argtypes = [ctypes.c_int] * 4
    lib.mvp_pointwise_wgrad_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_pointwise_wgrad_scratch_bytes.argtypes = [ctypes.c_int] * 4
    lib.mvp_pointwise_wgrad_mfma_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_pointwise_wgrad_mfma_scratch_bytes.argtypes = [ctypes.c_int] * 5
    lib.mvp_pointwise_max_backward_scratch_bytes.restype = ctypes.c_longlong
    lib.mvp_pointwise_max_backward_scratch_bytes.argtypes = [ctypes.c_int] * 4
    for name, sig in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = ctypes.c_int
        fn.argtypes = [_CT[k] for k in sig] + [ctypes.c_void_p]  # + stream
    if lib.mvp_abi_version() != ABI_VERSION:
        raise MvpOpsError("libmvpops.so ABI version mismatch")
    _lib = lib
    return lib


def _ptr(t):
    if t is None:
        return None
    if not t.is_cuda:
        raise MvpOpsError("libmvpops ops need GPU tensors (got %s); there is "
                          "no CPU fallback" % t.device)
    if not t.is_contiguous():
        raise MvpOpsError("libmvpops ops need contiguous tensors")
    return t.data_ptr()


_FN = {}   # name -> (ctypes function, signature): one attribute lookup per entry point, not per call

try:
    _raw_stream = torch._C._cuda_getCurrentRawStream   # the current stream's handle without building a Stream object
except AttributeError:   # pragma: no cover
    _raw_stream = None


def call(name, device, *args):
    """Invoke one C-ABI entry point on `device`'s current stream.  (Host cost matters for the small ops: a completion
    network issues ~600 of these per step; everything that can be looked up once is.)"""
    ent = _FN.get(name)
    if ent is None:
        ent = _FN[name] = (getattr(load(), name), SIGNATURES[name])
    fn, sig = ent
    assert len(sig) == len(args), (name, len(sig), len(args))
    cargs = []
    for kind, a in zip(sig, args):
        if kind == "p":
            cargs.append(_ptr(a) if isinstance(a, torch.Tensor) else ctypes.addressof(a) if isinstance(a, ctypes.Structure) else a)
        elif kind == "f":
            cargs.append(float(a))
        else:
            cargs.append(int(a))
    if not isinstance(device, torch.device):   # "cuda:1", 1: accepted like torch does
        device = torch.device("cuda", device) if isinstance(device, int) else torch.device(device)
    idx = device.index
    cur = torch.cuda.current_device()
    if idx is None:
        idx = cur
    if idx == cur:
        stream = _raw_stream(idx) if _raw_stream is not None else torch.cuda.current_stream(idx).cuda_stream
        rc = fn(*cargs, stream)
    else:
        with torch.cuda.device(idx):
            rc = fn(*cargs, torch.cuda.current_stream(idx).cuda_stream)
    if rc != MVP_OK:
        detail = _ERR.get(rc, "code %d" % rc)
        if rc == -3:
            detail += ": " + load().mvp_last_hip_error().decode()
        raise MvpOpsError("%s failed: %s" % (name, detail))


def emd_scratch_bytes(b, n, iters=None):
    """Scratch of mvp_emd_forward (the same for any number of rounds)."""
    return int(load().mvp_emd_scratch_bytes(int(b), int(n)))


class EmdPlan(ctypes.Structure):
    """MvpEmdPlan of include/mvpops.h: the launch plan of ONE mvp_emd_forward_plan call (negative field = compiled-in
    default; nothing process-wide is read or written).  call("mvp_emd_forward_plan", dev, ..., nbytes, EmdPlan(split=2))."""
    _fields_ = [("cluster", ctypes.c_int), ("same_xcd", ctypes.c_int), ("split", ctypes.c_int), ("resident_cap", ctypes.c_int)]

    def __init__(self, cluster=-1, same_xcd=-1, split=-1, resident_cap=-1):
        super().__init__(int(cluster), int(same_xcd), int(split), int(resident_cap))


def emd_configure(cluster=-1, same_xcd=-1, split=-1, resident_cap=-1):
    """Process-wide tuning knobs of mvp_emd_forward (negative = unchanged;
    cluster = 0: automatic).  Results do not depend on them."""
    rc = load().mvp_emd_configure(int(cluster), int(same_xcd), int(split), int(resident_cap))
    if rc != MVP_OK:
        raise MvpOpsError("mvp_emd_configure: %s" % _ERR.get(rc, rc))


EMD_RECORD_INTS = 24     # csrc/emd_common.h: struct EmdHandover (96 bytes per cloud, right before the statistics)


def emd_records(scratch, nbytes, b):
    """The per-cloud hand-over records and statistics mvp_emd_forward left at the end of its scratch buffer
    (`scratch`: uint8 tensor of `nbytes` bytes, as passed to the call) -> dict of numpy arrays:
      rounds, bids          the statistics words
      first_handover        round at which the first kernel handed the cloud over (0: it never did)
      next_round            0 once the cloud is finished
      unassigned            persons unassigned at the LAST hand-over
      final_width           cluster width of the launch that finished the cloud (0: the first kernel did)
      final_launch          1 = the launch after the first kernel, 2 = the tiered one, 3 = LDS-resident (in its own launch or fused into launch 1)
      gathered_rounds       rounds that ran with gathered bids (csrc/emd_lean.hip; as of the last hand-over / the end of the clustered rounds)"""
    import torch
    rb = EMD_RECORD_INTS * 4
    stats = scratch[nbytes - b * 16:].view(torch.int64).view(b, 2).cpu().numpy()
    rec = scratch[nbytes - b * 16 - b * rb: nbytes - b * 16].view(torch.int32).view(b, EMD_RECORD_INTS).cpu().numpy()
    return {"rounds": stats[:, 0], "bids": stats[:, 1], "next_round": rec[:, 0], "unassigned": rec[:, 1],
            "first_handover": rec[:, 18], "final_width": rec[:, 19] & 15, "final_launch": rec[:, 19] >> 4,
            "gathered_rounds": rec[:, 20]}


def fps_cluster_scratch_bytes(b):
    return int(load().mvp_fps_cluster_scratch_bytes(int(b)))


def fps_scratch_bytes(b, n):
    return int(load().mvp_fps_scratch_bytes(int(b), int(n)))


def knn_scratch_bytes(b, n, m):
    """Scratch of mvp_knn_sorted (n candidates, m queries per cloud)."""
    return int(load().mvp_knn_scratch_bytes(int(b), int(n), int(m)))


def chamfer_scratch_bytes(b, n, m):
    return int(load().mvp_chamfer_scratch_bytes(int(b), int(n), int(m)))


def scatter_scratch_bytes(b, n_dst, m_src, r):
    """Scratch of the *_grad_ws entry points (0: shape not covered, they run the plain kernels)."""
    return int(load().mvp_scatter_scratch_bytes(int(b), int(n_dst), int(m_src), int(r)))


def pointwise_wgrad_scratch_bytes(b, cin, cout, length):
    """Scratch of mvp_pointwise_wgrad (0: shape not covered)."""
    return int(load().mvp_pointwise_wgrad_scratch_bytes(int(b), int(cin), int(cout), int(length)))


def pointwise_wgrad_mfma_scratch_bytes(b, cin, cout, length, with_bias):
    """Scratch of mvp_pointwise_wgrad_mfma (0: shape not covered)."""
    return int(load().mvp_pointwise_wgrad_mfma_scratch_bytes(int(b), int(cin), int(cout), int(length), int(with_bias)))


def pointwise_max_backward_scratch_bytes(b, cin, cout, length):
    """Scratch of mvp_pointwise_max_backward's staged weight gradient (0: not covered, it gathers directly)."""
    return int(load().mvp_pointwise_max_backward_scratch_bytes(int(b), int(cin), int(cout), int(length)))


def exported_symbols():
    """All entry points include/mvpops.h declares."""
    return ["mvp_abi_version", "mvp_last_hip_error", "mvp_emd_scratch_bytes", "mvp_emd_configure", "mvp_chamfer_scratch_bytes", "mvp_knn_scratch_bytes", "mvp_fps_scratch_bytes", "mvp_fps_cluster_scratch_bytes",
            "mvp_scatter_scratch_bytes", "mvp_pointwise_wgrad_scratch_bytes", "mvp_pointwise_wgrad_mfma_scratch_bytes",
            "mvp_pointwise_max_backward_scratch_bytes", "mvp_share_gather_sum_lds_bytes"] \
        + list(SIGNATURES)
