"""Record / batch layouts in HBM and the key -> column-range maps the Python views are cut from.

Reference data model being replaced: ReplayBuffer.buffers = {key: float64[size, T or T+1, dim]}
(replay_buffer.py:23-24, shapes from config.configure_buffer config.py:184-208) and the dict of
[batch, dim] arrays a sampler returns (her.py:110-183).  Here one episode is ONE contiguous record of
(T+1) rows; a row is [o | ag | g | u | task_descr | extra] (float32) where `extra` holds every other key
(change, info_*) in sorted order, and a staged batch row is
[o | task_descr | u | g | o_2 | g_2 | r | ag | ag_2 | extra].
"""
from collections import OrderedDict

from curious_amd import _lib

CORE_KEYS = ('o', 'ag', 'g', 'u', 'task_descr')


def _pad4(n):
    return (n + 3) // 4 * 4


class RecordLayout:
    """Built from the reference's buffer_shapes dict {key: (T or T+1, dim)} (config.py:200-208)."""

    def __init__(self, buffer_shapes, T):
        self.T = int(T)
        shapes = {k: tuple(v) for k, v in buffer_shapes.items()}
        for k in ('o', 'ag', 'g', 'u'):
            if k not in shapes:
                raise KeyError('buffer_shapes lacks %r' % k)
        assert shapes['o'][0] == T + 1 and shapes['ag'][0] == T + 1, "'o' and 'ag' hold T+1 steps"
        self.dims = OrderedDict()
        self.dims['o'] = shapes['o'][1]
        self.dims['ag'] = shapes['ag'][1]
        self.dims['g'] = shapes['g'][1]
        self.dims['u'] = shapes['u'][1]
        self.dims['task_descr'] = shapes['task_descr'][1] if 'task_descr' in shapes else 0
        self.extra_keys = sorted(k for k in shapes if k not in CORE_KEYS)
        for k in self.extra_keys:
            assert shapes[k][0] == T, 'extra key %s must hold T steps' % k
            self.dims[k] = shapes[k][1] if len(shapes[k]) > 1 else 1
        self.keys = list(shapes.keys())            # reference order (for dict outputs)
        self.steps = {k: shapes[k][0] for k in shapes}
        # record row
        off = 0
        self.off = {}
        for k in CORE_KEYS:
            self.off[k] = off
            off += self.dims[k]
        self.off_extra = off
        for k in self.extra_keys:
            self.off[k] = off
            off += self.dims[k]
        self.dimextra = off - self.off_extra
        self.row_stride = _pad4(off)
        self.rec_floats = (self.T + 1) * self.row_stride
        # staged batch row
        d = self.dims
        b = OrderedDict()
        boff = 0
        for name, dim in (('o', d['o']), ('task_descr', d['task_descr']), ('u', d['u']), ('g', d['g']),
                          ('o_2', d['o']), ('g_2', d['g']), ('r', 1), ('ag', d['ag']), ('ag_2', d['ag'])):
            b[name] = (boff, dim)
            boff += dim
        self.boff_extra = boff
        for k in self.extra_keys:
            b[k] = (boff, d[k])
            boff += d[k]
        self.batch_cols = b
        self.batch_stride = _pad4(boff)

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ('_c_layout', '_c_batch_layout')}   # ctypes structs

    # ---------------------------------------------------------------- C structs
    def c_layout(self):
        """curious_layout_t of this layout (built once: the layout is immutable and the library only reads it)."""
        if getattr(self, '_c_layout', None) is not None:
            return self._c_layout
        L = _lib.Layout()
        L.T = self.T
        L.dimo, L.dimag, L.dimg, L.dimu = self.dims['o'], self.dims['ag'], self.dims['g'], self.dims['u']
        L.dimtd, L.dimextra = self.dims['task_descr'], self.dimextra
        L.off_o, L.off_ag, L.off_g, L.off_u = self.off['o'], self.off['ag'], self.off['g'], self.off['u']
        L.off_td, L.off_extra, L.row_stride = self.off['task_descr'], self.off_extra, self.row_stride
        self._c_layout = L
        return L

    def c_batch_layout(self):
        if getattr(self, '_c_batch_layout', None) is not None:
            return self._c_batch_layout
        B = _lib.BatchLayout()
        c = self.batch_cols
        B.off_o, B.off_td, B.off_u, B.off_g = c['o'][0], c['task_descr'][0], c['u'][0], c['g'][0]
        B.off_o2, B.off_g2, B.off_r = c['o_2'][0], c['g_2'][0], c['r'][0]
        B.off_ag, B.off_ag2, B.off_extra, B.stride = c['ag'][0], c['ag_2'][0], self.boff_extra, self.batch_stride
        self._c_batch_layout = B
        return B

    # ---------------------------------------------------------------- views
    def record_views(self, storage):
        """storage: tensor [..., T+1, row_stride] -> {key: view [..., T or T+1, dim]} (reference key set)."""
        out = OrderedDict()
        for k in self.keys:
            o, d = self.off[k], self.dims[k]
            out[k] = storage[..., :self.steps[k], o:o + d]
        return out

    def batch_views(self, batch, keys=None):
        """batch: tensor [n, batch_stride] -> {key: view [n, dim]}."""
        out = OrderedDict()
        for k in (keys if keys is not None else self.batch_cols.keys()):
            o, d = self.batch_cols[k]
            out[k] = batch[:, o:o + d]
        return out

    def same_as(self, other):
        return (self.T == other.T and dict(self.dims) == dict(other.dims)
                and self.extra_keys == other.extra_keys)


def pack_episodes(layout, episode_batch):
    """Host-side packing of a reference-style episode dict {key: [E, T or T+1, dim]} (NumPy arrays or CPU/GPU
    tensors) into a float32 record block [E, T+1, row_stride] (NumPy).  Rows t = T carry o and ag only."""
    import numpy as np
    first = np.asarray(_to_numpy(episode_batch['u']))
    E = first.shape[0]
    rec = np.zeros([E, layout.T + 1, layout.row_stride], np.float32)
    for k in layout.keys:
        if k not in episode_batch:
            raise KeyError('episode batch lacks key %r' % k)
        v = _to_numpy(episode_batch[k])
        v = v.reshape(E, layout.steps[k], -1)
        o, d = layout.off[k], layout.dims[k]
        assert v.shape[2] == d, 'key %s has dim %d, expected %d' % (k, v.shape[2], d)
        rec[:, :layout.steps[k], o:o + d] = v
    return rec


def _to_numpy(v):
    import numpy as np
    if hasattr(v, 'detach'):
        v = v.detach().cpu().numpy()
    return np.asarray(v)
