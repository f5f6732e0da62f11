"""Shared by tests/golden/make_model_golden.py (which runs the REFERENCE's
completion/model_utils.py and models/*.py in the build container) and by the
tests that run THIS repo's counterparts against the fixtures it wrote --
TEST INFRASTRUCTURE.

* `fill_named`: parameters from a generator seeded by the parameter's NAME
  (both trees use the reference's names), so no state dict is stored.
* `grad_probes`: two float64 numbers per parameter gradient (its norm and its
  dot product with a name-seeded direction): a 17 M-parameter gradient pinned
  in a few KB.
* `Recorder` / `Replayer`: the index-producing steps of a forward pass (FPS,
  kNN graphs, pooling neighbours, three_nn, ball_query) are recorded from the
  reference run; the replayer lets the code under test compute each of them
  itself, COUNTS the rows that differ from the recorded ones, checks that each
  of them is a tie up to float32 rounding (`unexplained` stays empty), and
  hands the recorded ones on -- so one neighbour flipped by a last-bit
  difference in a distance shows up as a counted, explained flip, not as a
  different network downstream.
"""
import math
import zlib

import numpy as np
import torch


class Args(dict):
    """yaml -> attribute dict (the reference uses munch, completion/train.py:200)."""
    __getattr__ = dict.get


def _seed(text):
    return zlib.crc32(text.encode())


def seeded_uniform(tag, shape, lo=0.0, hi=1.0, dtype=torch.float32):
    g = torch.Generator().manual_seed(_seed(tag))
    return (torch.rand(*shape, generator=g, dtype=torch.float64) * (hi - lo) + lo).to(dtype)


def fill_named(module, tag):
    """Every tensor of module.state_dict() <- uniform values from a generator seeded by tag + its name:
    weights in +-sqrt(3 / fan_in) (unit gain), everything one-dimensional in +-0.1."""
    with torch.no_grad():
        for name, t in module.state_dict().items():
            if t.dim() > 1:
                bound = math.sqrt(3.0 / t[0].numel())
            else:
                bound = 0.1
            t.copy_(seeded_uniform(tag + ":" + name, tuple(t.shape), -bound, bound).to(t.dtype))
    return module


def grad_probes(module, tag):
    """{name: (norm, dot with a name-seeded direction)} of every parameter gradient, float64; parameters
    without a gradient give (0, 0)."""
    names, norms, dots = [], [], []
    for name, p in module.named_parameters():
        g = p.grad
        names.append(name)
        if g is None:
            norms.append(0.0)
            dots.append(0.0)
            continue
        g = g.detach().double().cpu()
        r = seeded_uniform(tag + ":probe:" + name, tuple(g.shape), -1.0, 1.0, torch.float64)
        norms.append(float(g.norm()))
        dots.append(float((g * r).sum()))
    return names, np.asarray(norms, np.float64), np.asarray(dots, np.float64)


# ---------------------------------------------------------------------------
# keys of the recorded steps
# ---------------------------------------------------------------------------
def key_fps(xyz, m):
    return "fps|%d|%d|%d" % (xyz.size(0), xyz.size(1), int(m))


def key_knn(x, k):
    return "knn|%d|%d|%d|%d" % (x.size(0), x.size(1), x.size(2), int(k))


def key_knn_point(pk, point_input, point_output):
    return "knnpt|%d|%d|%d|%d|%d" % (point_input.size(0), point_input.size(1), point_output.size(1),
                                     point_input.size(2), int(pk))


def key_three_nn(target, source):
    return "tnn|%d|%d|%d" % (target.size(0), target.size(1), source.size(1))


def key_ball_query(sample_num, xyz, center):
    return "bq|%d|%d|%d|%d" % (xyz.size(0), xyz.size(1), center.size(1), int(sample_num))


def _small_int(a):
    a = np.asarray(a)
    return a.astype(np.int16) if a.size and a.max() < 32768 and a.min() >= -32768 else a.astype(np.int32)


class Recorder:
    """Wraps the reference's index-producing callables; `arrays()` -> {"<key>#<i>": value} in call order per key."""

    def __init__(self):
        self.calls = {}

    def _put(self, key, value):
        lst = self.calls.setdefault(key, [])
        # the reference repeats some calls on identical inputs (get_uniform_loss runs the same FPS five times,
        # model_utils.py:205-210): keep one copy
        if lst and np.array_equal(lst[-1], value):
            return
        lst.append(value)

    def fps(self, fn):
        def wrapped(xyz, m):
            out = fn(xyz, m)
            self._put(key_fps(xyz, m), _small_int(out.cpu().numpy()))
            return out
        return wrapped

    def knn(self, fn):
        def wrapped(x, k):
            out = fn(x, k)
            self._put(key_knn(x, k), _small_int(out.cpu().numpy()))
            return out
        return wrapped

    def knn_point(self, fn):
        def wrapped(pk, point_input, point_output):
            dist, idx = fn(pk, point_input, point_output)
            if point_input is not point_output:          # (the uniform loss's self-queries are not index steps)
                self._put(key_knn_point(pk, point_input, point_output), _small_int(idx.cpu().numpy()))
            return dist, idx
        return wrapped

    def three_nn(self, fn):
        def wrapped(target, source):
            dist, idx = fn(target, source)
            self._put(key_three_nn(target, source) + "|idx", _small_int(idx.cpu().numpy()))
            self._put(key_three_nn(target, source) + "|dist", dist.cpu().numpy())
            return dist, idx
        return wrapped

    def ball_query(self, fn):
        def wrapped(min_radius, max_radius, sample_num, xyz, center):
            out = fn(min_radius, max_radius, sample_num, xyz, center)
            self._put(key_ball_query(sample_num, xyz, center), _small_int(out.cpu().numpy()))
            return out
        return wrapped

    def arrays(self):
        return {"%s#%d" % (k, i): v for k, lst in self.calls.items() for i, v in enumerate(lst)}


class Replayer:
    """Counterpart of Recorder for the code under test.  `install(monkeypatch, modules)` wraps, in every module of
    `modules`, the names furthest_point_sample / knn / knn_point_idx / three_nn / ball_query.  A wrapped call runs
    the real callable, compares its rows with the recorded ones (`flips[key] = [rows that differ, rows]`) and
    returns the RECORDED value in the real result's dtype / device."""

    def __init__(self, arrays, prefix=""):
        self.rec = {}
        for name, v in arrays.items():
            if not name.startswith(prefix):
                continue
            key, i = name[len(prefix):].rsplit("#", 1)
            self.rec.setdefault(key, {})[int(i)] = v
        self.cursor = {}
        self.flips = {}
        self.unexplained = []
        self.max_dist_err = 0.0

    def _next(self, key):
        have = self.rec.get(key)
        assert have is not None, "no recorded step %r (recorded: %s)" % (key, sorted(self.rec))
        i = min(self.cursor.get(key, 0), len(have) - 1)      # repeated identical calls were stored once
        self.cursor[key] = i + 1
        return have[i]

    def _idx(self, key, got, explain=None):
        want = self._next(key)
        g = got.detach().cpu().numpy()
        assert g.shape == want.shape, (key, g.shape, want.shape)
        rows = (g.reshape(-1, g.shape[-1]) != want.reshape(-1, want.shape[-1])).any(axis=1)
        f = self.flips.setdefault(key, [0, 0])
        f[0] += int(rows.sum())
        f[1] += int(rows.size)
        if rows.any() and explain is not None:
            for r in np.nonzero(rows)[0]:
                if not explain(int(r), g.reshape(-1, g.shape[-1])[r], want.reshape(-1, want.shape[-1])[r].astype(np.int64)):
                    self.unexplained.append((key, int(r)))
        return torch.from_numpy(want.astype(np.int64)).to(device=got.device, dtype=got.dtype)

    # A flip is EXPLAINED when the two answers are equally good up to float32 rounding of the inputs' distances:
    # recomputed in float64 from the inputs the code under test actually saw.
    TIE = 1e-5

    @staticmethod
    def _sq(a, b):
        d = a.astype(np.float64) - b.astype(np.float64)
        return (d * d).sum(axis=-1)

    def fps(self, fn):
        def wrapped(xyz, m):
            pts = xyz.detach().cpu().numpy()

            def explain(b, got, want):
                j = int(np.nonzero(got != want)[0][0])          # the first pick that differs; later ones follow from it
                mind = np.min(self._sq(pts[b][:, None, :], pts[b][want[:j]][None, :, :]), axis=1)
                return abs(mind[got[j]] - mind[want[j]]) <= self.TIE * mind.max()
            return self._idx(key_fps(xyz, m), fn(xyz, m), explain)
        return wrapped

    def _ranked(self, key, got, queries, cands):
        """queries (B, M, C), cands (B, N, C) numpy; got (B, M, k): a differing row is explained when the two lists'
        sorted distances agree within TIE of the row's largest."""
        m = queries.shape[1]

        def explain(r, g_row, w_row):
            b, q = divmod(r, m)
            dg = np.sort(self._sq(cands[b][g_row], queries[b, q][None, :]))
            dw = np.sort(self._sq(cands[b][w_row], queries[b, q][None, :]))
            return np.abs(dg - dw).max() <= self.TIE * max(dw.max(), 1e-30)
        return self._idx(key, got, explain)

    def knn(self, fn):
        def wrapped(x, k):
            pts = x.detach().transpose(1, 2).cpu().numpy()
            return self._ranked(key_knn(x, k), fn(x, k), pts, pts)
        return wrapped

    def knn_point_idx(self, fn):
        def wrapped(pk, pi, po):
            return self._ranked(key_knn_point(pk, pi, po), fn(pk, pi, po), po.detach().cpu().numpy(), pi.detach().cpu().numpy())
        return wrapped

    def three_nn(self, fn):
        def wrapped(target, source):
            dist, idx = fn(target, source)
            key = key_three_nn(target, source)
            idx = self._ranked(key + "|idx", idx, target.detach().cpu().numpy(), source.detach().cpu().numpy())
            want = torch.from_numpy(self._next(key + "|dist")).to(dist.device)
            self.max_dist_err = max(self.max_dist_err, float((dist - want).abs().max()))
            return want, idx
        return wrapped

    def ball_query(self, fn):
        return lambda lo, hi, s, xyz, c: self._idx(key_ball_query(s, xyz, c), fn(lo, hi, s, xyz, c))

    def install(self, monkeypatch, modules):
        table = {"furthest_point_sample": self.fps, "knn": self.knn, "knn_point_idx": self.knn_point_idx,
                 "three_nn": self.three_nn, "ball_query": self.ball_query}
        wrapped = {}
        for mod in modules:
            for name, wrap in table.items():
                fn = getattr(mod, name, None)
                if fn is None:
                    continue
                if fn not in wrapped:
                    wrapped[fn] = wrap(fn)
                monkeypatch.setattr(mod, name, wrapped[fn])
        return self

    def total_flips(self, kind=None):
        items = [(k, v) for k, v in self.flips.items() if kind is None or k.startswith(kind)]
        return sum(v[0] for _, v in items), sum(v[1] for _, v in items)
