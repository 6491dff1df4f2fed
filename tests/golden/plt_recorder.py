"""A stand-in for matplotlib.pyplot that RECORDS what is drawn (imshow / plot / bar / barh / text and axis limits) instead
of rendering it: lets the reference's own plotting function and this project's be compared on the CONTENT of the figure
(the arrays handed to the artists), independent of the rasteriser.  Test infrastructure only."""
import numpy as np


class _Spine:
    def set_visible(self, v):
        pass


class _Ax:
    def __init__(self, rec, key):
        self.rec, self.key = rec, key
        self.spines = {k: _Spine() for k in ("left", "right", "top", "bottom")}

    def _put(self, what, *args, **kw):
        self.rec.calls.append((self.key, what, [np.asarray(a) if isinstance(a, (list, tuple, np.ndarray)) else a for a in args], kw))

    def imshow(self, m, **kw): self._put("imshow", np.asarray(m, dtype=np.float64), **kw)
    def plot(self, x, y, *a, **kw): self._put("plot", np.asarray(x, dtype=np.float64), np.asarray(y, dtype=np.float64))
    def bar(self, x, h, **kw): self._put("bar", np.asarray(x, dtype=np.float64), np.asarray(h, dtype=np.float64), **kw)
    def barh(self, y, w, **kw): self._put("barh", np.asarray(y, dtype=np.float64), np.asarray(w, dtype=np.float64), **kw)
    def text(self, x, y, s, **kw): self._put("text", x, y, s)
    def set_xlim(self, *a): self._put("xlim", *a)
    def set_ylim(self, *a): self._put("ylim", *a)
    def set_xticks(self, *a): pass
    def set_yticks(self, *a): pass
    def axis(self, *a): pass
    # save_distributed_info_plane (visualization.py:83-113): a twin axis, z-order juggling
    def twinx(self): return _Ax(self.rec, (self.key, "twin"))
    def set_zorder(self, z): pass
    def get_zorder(self): return 0
    @property
    def patch(self): return _Spine()


class _GridSpec:
    def __getitem__(self, key):
        return key


class _Fig:
    def __init__(self, rec):
        self.rec = rec

    def add_gridspec(self, *a, **kw):
        return _GridSpec()

    def add_subplot(self, key):
        return _Ax(self.rec, key)

    def gca(self):
        return _Ax(self.rec, "main")

    def savefig(self, fname, **kw):
        self.rec.saved.append(fname)


class Recorder:
    """rec = Recorder(); use `rec` wherever `plt` is expected; rec.calls = [(subplot key, artist, arrays, kwargs)]."""

    def __init__(self):
        self.calls, self.saved = [], []

    def figure(self, **kw):
        return _Fig(self)

    def axis(self, *a): pass
    def gca(self): return _Ax(self, "main")
    def savefig(self, fname, **kw): self.saved.append(fname)
    def clf(self): pass
    def close(self, *a): pass

    def plots(self, key):
        """every (x, y) handed to ax.plot on the subplot `key`, in call order"""
        return [(a[0], a[1]) for k, what, a, kw in self.calls if k == key and what == "plot"]

    def content(self):
        """{(subplot key, artist): [arrays]} with the first call per key kept."""
        out = {}
        for key, what, args, kw in self.calls:
            out.setdefault((key, what), args)
        return out
