"""Batch engine: device buffers (torch tensors), contexts, and the batched planner over libfcpp.so.

torch is used for device memory and streams only; all arithmetic happens in the HIP library.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib as L

_contexts = {}


def _torch():
    import torch
    return torch


class Context:
    """One fcpp_ctx per device, bound to torch's current stream at every call."""

    def __init__(self, device=0):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError('no GPU visible: the fcpp operators have no CPU fallback')
        self.device = int(device)
        self.lib = L.load()
        h = C.c_void_p()
        L.check(self.lib.fcpp_ctx_create(self.device, C.byref(h)))
        self.handle = h
        self._bound = None          # the stream handle the library was last given
        self._arena = None          # (lane, pitch) of the output arena, asked once

    def bind_stream(self):
        torch = _torch()
        s = torch.cuda.current_stream(self.device).cuda_stream
        if s != self._bound:        # (only this method sets the library's stream)
            L.check(self.lib.fcpp_ctx_set_stream(self.handle, C.c_void_p(s)))
            self._bound = s

    def arena(self):
        """-> (lane bytes, pitch bytes) of the output arena, (0, 0) without one; asked once (reserve_outputs asks again)"""
        if self._arena is None:
            self._arena = self.outputs_info()[:2]
        return self._arena

    def reserve_outputs(self, lane_gib=24.0, pitch_gib=24.0):
        """Give the context its output ARENA (fcpp_ctx_reserve_outputs): one device allocation of 4 x pitch + lane, made once -- it takes
        the driver seconds, so it belongs to start-up, not to a plan call -- in which Batch.alloc() then places the five output arrays of
        every batch a pitch apart (far apart they are written a class faster than back to back).  Any number of live
        batches share it.  Raises when the device has not that much room."""
        L.check(self.lib.fcpp_ctx_reserve_outputs(self.handle, int(lane_gib * 2**30), int(pitch_gib * 2**30)))
        self._arena = None

    def outputs_info(self):
        """-> (lane bytes, pitch bytes, live bytes per lane) of the output arena; zeros without one"""
        a, b, c = C.c_int64(), C.c_int64(), C.c_int64()
        L.check(self.lib.fcpp_ctx_outputs_info(self.handle, C.byref(a), C.byref(b), C.byref(c)))
        return a.value, b.value, c.value

    def set_setup(self, mode):
        """Where batches are set up from now on: 'auto' (on the device where the device planner takes the batch: the reference's sampling,
        no obstacle-aware swaths), 'host', 'device' (fcpp_ctx_set_setup)."""
        m = {'auto': L.SETUP_AUTO, 'host': L.SETUP_HOST, 'device': L.SETUP_DEVICE}[mode] if isinstance(mode, str) else int(mode)
        L.check(self.lib.fcpp_ctx_set_setup(self.handle, m))

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.lib.fcpp_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


def get_context(device=None):
    torch = _torch()
    if device is None:
        device = torch.cuda.current_device() if torch.cuda.is_available() else 0
    device = int(device)
    if device not in _contexts:
        _contexts[device] = Context(device)
    return _contexts[device]


def _ptr(t):
    return C.c_void_p(t.data_ptr() if t is not None else 0)


def make_vehicle(vp=None, **kw):
    """fcpp_vehicle from a VehicleParams-like object (attributes named as MLP:31-38) or keywords."""
    v = L.default_vehicle()
    if vp is not None:
        for n, _ in L.Vehicle._fields_:
            setattr(v, n, float(getattr(vp, n)))
    for k, val in kw.items():
        setattr(v, k, float(val))
    return v


def make_options(turn_model=L.TURN_ARC, sample_spacing=0.0, clothoid_frac=0.5, clothoid_fit=1, geofence_tol=1e-6,
                 avoid_obstacles=False, ring_order=L.RING_AS_VERTICES):
    """fcpp_options.  avoid_obstacles: clip the swaths of layer 1 at the obstacles and drive around them (build-defined,
    include/fcpp.h); False = the reference's behaviour (obstacles only flag the points inside them)."""
    o = L.default_options()
    o.obstacle_mode = L.OBSTACLES_AVOID if avoid_obstacles else L.OBSTACLES_FLAG
    o.turn_model = int(turn_model)
    o.sample_spacing = float(sample_spacing)
    o.clothoid_frac = float(clothoid_frac)
    o.clothoid_fit = int(clothoid_fit)
    o.geofence_tol = float(geofence_tol)
    o.ring_order = int(ring_order)      # order of the inset corners in Shapely's buffer(-d).exterior.coords (include/fcpp.h)
    return o


@dataclass
class FieldSpec:
    """Constructor arguments of one planner (MLP:63-72)."""
    field_length: float = None
    field_width: float = None
    field_vertices: list = None
    obstacles: list = None
    start_point: tuple = None
    end_point: tuple = None


_FIELD_DT = None


def _field_dtype():
    global _FIELD_DT
    if _FIELD_DT is None:
        _FIELD_DT = np.dtype(L.Field)          # fcpp_field as a numpy record (same layout as the ctypes structure)
    return _FIELD_DT


class FieldTable:
    """The constructor arguments of n planners as ONE array of fcpp_field records (+ the batch's obstacle polygons in CSR form): what
    fcpp_batch_create / fcpp_plan_count take, built with array operations instead of a Python loop over FieldSpec objects (65 536
    specs cost 0.45 s to build and pack one by one; the table of the same fields 3 ms).

        FieldTable.from_rectangles(LH)            (n, 2) array of (field_length, field_width)                  MLP:127-132
        FieldTable.from_vertices(V)               (n, 4, 2) array of field_vertices                              MLP:116-122
        FieldTable.from_specs([FieldSpec, ...])   the general case (obstacles, start / end points per field)

    start_points / end_points: (n, 2) arrays, NaN rows = not given.  A table can be sliced (`table[lo:hi]`: the shard of a rank); the
    slice shares the polygon table."""

    def __init__(self, rec, poly_offsets=None, poly_x=None, poly_y=None):
        self._cargs = None
        self._dev = None            # the records' copy in device memory after to_device() (a torch uint8 tensor)
        self.rec = rec
        self._pinned = None         # the pinned buffer the records live in after pin()
        self.poly_offsets = np.zeros(1, dtype=np.int64) if poly_offsets is None else np.ascontiguousarray(poly_offsets, dtype=np.int64)
        self.poly_x = np.zeros(0, dtype=np.float64) if poly_x is None else np.ascontiguousarray(poly_x, dtype=np.float64)
        self.poly_y = np.zeros(0, dtype=np.float64) if poly_y is None else np.ascontiguousarray(poly_y, dtype=np.float64)

    @property
    def rec(self):
        """the fcpp_field records (a numpy record array; written in place they stay what the library reads)"""
        return self._rec

    @rec.setter
    def rec(self, value):
        self._rec = value
        self._cargs = None          # (the pointers c_args() made are another array's)
        self._dev = None            # (and so is the copy on the device)

    def __len__(self):
        return int(self.rec.shape[0])

    def __getitem__(self, sl):
        if not isinstance(sl, slice):
            raise TypeError('a FieldTable is sliced, not indexed')
        t = FieldTable(self.rec[sl], self.poly_offsets, self.poly_x, self.poly_y)
        t._pinned = self._pinned    # (a slice of pinned records is pinned, and keeps the buffer alive)
        if self._dev is not None:   # (a contiguous slice of a table on the device is on the device)
            lo, hi, step = sl.indices(len(self))
            if step == 1:
                sz = _field_dtype().itemsize
                t._dev = self._dev[lo * sz:max(lo, hi) * sz]
        return t

    def pin(self):
        """Moves the records into pinned host memory (once; needs a GPU): fcpp_batch_create / fcpp_plan_points then let the device read
        them where they lie instead of copying them first -- for a table that is planned more than once, or built in place.  -> self"""
        if self._pinned is None:
            torch = _torch()
            dt = _field_dtype()
            n = len(self)
            buf = torch.empty(max(n, 1) * dt.itemsize, dtype=torch.uint8, pin_memory=True)
            rec = buf.numpy().view(dt)[:n]
            rec[...] = self.rec
            self.rec, self._pinned, self._cargs = rec, buf, None
        return self

    def to_device(self, device=None):
        """Copies the records into DEVICE memory (once; needs a GPU): fcpp_batch_plan / fcpp_batch_create / fcpp_plan_points then read them
        where they lie -- nothing crosses PCIe in front of the plan call's first kernel (the headline's 512 KB of records were ~10 us of
        it), as for a table whose fields are made on the GPU.  `rec` stays the host's copy (slicing, sharding): records written in place
        afterwards are NOT seen by the library until to_device() is called again on a table with a new `rec`.  The host paths of the library
        (a handful of fields, AVOID mode, FCPP_SETUP=host) copy the records back first.  -> self"""
        if self._dev is None:
            torch = _torch()
            rec = np.ascontiguousarray(self.rec)
            dev = torch.device('cuda', torch.cuda.current_device() if device is None else device)
            self._dev = torch.from_numpy(rec.view(np.uint8).reshape(-1).copy()).to(dev)
            self._cargs = None
        return self

    @staticmethod
    def _points(rec, name, pts):
        if pts is None:
            return
        pts = np.asarray(pts, dtype=np.float64).reshape(len(rec), 2)
        has = ~np.isnan(pts).any(axis=1)
        rec['has_' + name] = has
        rec[name + '_x'] = np.where(has, pts[:, 0], 0.0)
        rec[name + '_y'] = np.where(has, pts[:, 1], 0.0)

    @classmethod
    def from_vertices(cls, V, start_points=None, end_points=None):
        V = np.asarray(V, dtype=np.float64)
        if V.ndim != 3 or V.shape[1:] != (4, 2):
            raise ValueError('only quadrilateral fields are supported (4 vertices)')
        rec = np.zeros(V.shape[0], dtype=_field_dtype())
        rec['vx'], rec['vy'] = V[:, :, 0], V[:, :, 1]
        rec['from_vertices'] = 1
        cls._points(rec, 'start', start_points)
        cls._points(rec, 'end', end_points)
        return cls(rec)

    @classmethod
    def from_rectangles(cls, LH, start_points=None, end_points=None):
        LH = np.asarray(LH, dtype=np.float64).reshape(-1, 2)
        rec = np.zeros(LH.shape[0], dtype=_field_dtype())
        rec['vx'][:, 1] = rec['vx'][:, 2] = LH[:, 0]          # (0,0), (L,0), (L,H), (0,H)  (MLP:127-132)
        rec['vy'][:, 2] = rec['vy'][:, 3] = LH[:, 1]
        cls._points(rec, 'start', start_points)
        cls._points(rec, 'end', end_points)
        return cls(rec)

    @classmethod
    def from_specs(cls, specs):
        """Raises ValueError like MLP:135 when a spec names no field."""
        n = len(specs)
        rec = np.zeros(n, dtype=_field_dtype())
        V = np.zeros((n, 4, 2), dtype=np.float64)
        offs, px, py = [0], [], []
        nan2 = (float('nan'), float('nan'))
        starts, ends = [], []
        for i, s in enumerate(specs):
            if s.field_vertices is not None:
                vs = list(s.field_vertices)
                if len(vs) != 4:
                    raise ValueError('only quadrilateral fields are supported (4 vertices)')
                V[i] = vs
                rec['from_vertices'][i] = 1
            elif s.field_length is not None and s.field_width is not None:
                V[i] = ((0.0, 0.0), (s.field_length, 0.0), (s.field_length, s.field_width), (0.0, s.field_width))
            else:
                raise ValueError('必须提供 field_vertices 或 (field_length, field_width)')
            starts.append(nan2 if s.start_point is None else (float(s.start_point[0]), float(s.start_point[1])))
            ends.append(nan2 if s.end_point is None else (float(s.end_point[0]), float(s.end_point[1])))
            obs = s.obstacles or []
            rec['obstacle_first'][i] = len(offs) - 1
            rec['n_obstacles'][i] = len(obs)
            for poly in obs:
                for (x, y) in poly:
                    px.append(float(x))
                    py.append(float(y))
                offs.append(len(px))
        rec['vx'], rec['vy'] = V[:, :, 0], V[:, :, 1]
        cls._points(rec, 'start', np.asarray(starts, dtype=np.float64).reshape(n, 2))
        cls._points(rec, 'end', np.asarray(ends, dtype=np.float64).reshape(n, 2))
        return cls(rec, offs, px, py)

    def c_args(self):
        """-> (fcpp_field pointer, fcpp_polys, objects to keep alive during the call)"""
        if self._cargs is not None:
            return self._cargs
        # (the pointers are kept only when they point INTO self.rec: in-place writes to the records then stay what the library reads.  A
        # non-contiguous view -- table[::2] -- is copied here, at every call, so that writes made since the last one are seen)
        contiguous = self.rec.flags.c_contiguous
        rec = self.rec if contiguous else np.ascontiguousarray(self.rec)
        polys = L.Polys(len(self.poly_offsets) - 1, self.poly_offsets.ctypes.data_as(L.c_i64_p), self.poly_x.ctypes.data_as(L.c_double_p),
                        self.poly_y.ctypes.data_as(L.c_double_p))
        if self._dev is not None:
            cargs = (C.cast(C.c_void_p(self._dev.data_ptr()), C.POINTER(L.Field)), polys, [self._dev, self.poly_offsets, self.poly_x, self.poly_y])
            self._cargs = cargs
            return cargs
        cargs = (C.cast(C.c_void_p(rec.ctypes.data), C.POINTER(L.Field)), polys, [rec, self.poly_offsets, self.poly_x, self.poly_y])
        if contiguous:
            self._cargs = cargs
        return cargs


def as_table(specs):
    return specs if isinstance(specs, FieldTable) else FieldTable.from_specs(specs)


def pack_fields(specs):
    """-> (Field pointer, Polys, keep-alive list).  Raises ValueError like MLP:135 when no field is given."""
    return as_table(specs).c_args()


class InfoTable:
    """fcpp_field_info of n fields: `infos[i]` is the ctypes record (attribute access as before), `infos.array` a numpy record view of
    all of them (`infos.array['n_main']`), `infos.counts()` the points per field -- no Python loop over 65 536 records."""

    def __init__(self, n):
        self.n = int(n)
        self._c = (L.FieldInfo * max(self.n, 1))()
        self.array = np.frombuffer(self._c, dtype=np.dtype(L.FieldInfo))[:self.n]

    def __len__(self):
        return self.n

    def __getitem__(self, i):
        if isinstance(i, slice):
            return [self._c[k] for k in range(*i.indices(self.n))]
        if i < 0:
            i += self.n
        if not 0 <= i < self.n:
            raise IndexError(i)
        return self._c[i]

    def __iter__(self):
        return (self._c[k] for k in range(self.n))

    def __eq__(self, other):            # (an empty table equals an empty list, as the list this used to be)
        try:
            return len(other) == self.n and all(a is b or bytes(a) == bytes(b) for a, b in zip(self, other))
        except TypeError:
            return NotImplemented

    def counts(self):
        return (self.array['n_main'] + self.array['n_head']).astype(np.int64)


def plan_count(specs, vehicle, options):
    """Host-only sizing/decisions (fcpp_plan_count); needs no GPU.  -> InfoTable"""
    lib = L.load()
    arr, polys, _keep = pack_fields(specs)
    info = InfoTable(len(specs))
    L.check(lib.fcpp_plan_count(C.byref(vehicle), C.byref(options), len(specs), arr, C.byref(polys), info._c))
    return info


def plan_points(specs, vehicle, options, device=None):
    """Points per field (n_main + n_head; 0 for a field that raises) as a numpy int64 array: the sizing a sharded job cuts its blocks on
    (fcpp_plan_points).  Computed on the GPU where the device-side setup takes the batch, else on the host."""
    ctx = get_context(device)
    arr, polys, _keep = pack_fields(specs)
    out = np.zeros(len(specs), dtype=np.int64)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_plan_points(ctx.handle, C.byref(vehicle), C.byref(options), len(specs), arr, C.byref(polys), out.ctypes.data_as(L.c_i64_p)))
    return out


class _ArenaArrays:
    """Five output arrays from the context's arena (fcpp_outputs_alloc), handed to torch as views of the library's memory through the CUDA
    array interface; the allocation goes back to the arena when the last view is gone."""

    class _View:
        def __init__(self, owner, ptr, n, typestr):
            self._owner = owner
            self.__cuda_array_interface__ = {'shape': (n,), 'typestr': typestr, 'data': (ptr, False), 'version': 2, 'strides': None}

    def __init__(self, ctx, n_points, ptrs=None):
        self.ctx = ctx
        if ptrs is None:
            ptrs = [C.c_void_p() for _ in range(5)]
            L.check(ctx.lib.fcpp_outputs_alloc(ctx.handle, int(n_points), 0, *[C.byref(q) for q in ptrs]))
            ptrs = [q.value for q in ptrs]
        self.ptrs = list(ptrs)          # (given: arrays the library has already allocated -- fcpp_batch_plan -- which this object now owns)

    def tensors(self, n_points):
        torch = _torch()
        dev = torch.device('cuda', self.ctx.device)
        out = []
        for k, p in enumerate(self.ptrs):
            v = self._View(self, p, int(n_points), '<f8' if k < 4 else '<i4')
            t = torch.as_tensor(v, device=dev)
            t._fcpp_owner = v               # (the view object -- and through it this allocation -- lives as long as the tensor object)
            out.append(t)
        return out

    def __del__(self):
        try:
            if getattr(self, 'ptrs', None) and self.ctx.handle:
                self.ctx.lib.fcpp_outputs_free(self.ctx.handle, C.c_void_p(self.ptrs[0]))
                self.ptrs = None
        except Exception:
            pass


class _StatsView:
    """The statistics records a batch keeps in its own device allocation (fcpp_batch_own_stats), handed to torch through the CUDA array
    interface; the batch's close() waits for the last tensor made from this view."""

    def __init__(self, batch, ptr, n_words):
        self._batch = batch
        self.__cuda_array_interface__ = {'shape': (n_words,), 'typestr': '<i8', 'data': (ptr, False), 'version': 2, 'strides': None}

    def __del__(self):
        try:
            b = self._batch
            b._stats_views = 0
            if getattr(b, '_close_pending', False):
                b.close()
        except Exception:
            pass


class BatchResult:
    """Device-resident result of one batch: SoA tensors + per-field stats."""

    def __init__(self, batch, x, y, kappa, v, flagseg, stats_raw):
        self.batch, self.x, self.y, self.kappa, self.v, self.flagseg = batch, x, y, kappa, v, flagseg
        self.stats_raw = stats_raw   # (n_fields, 13) int64 view of fcpp_field_stats

    def stats(self):
        """-> dict of numpy arrays, one entry per field."""
        raw = self.stats_raw.cpu().numpy()
        names = [n for n, _ in L.FieldStats._fields_]
        out = {}
        for k, n in enumerate(names):
            col = raw[:, k]
            out[n] = col.view(np.float64).copy() if k < L.STATS_DOUBLES else col.copy()
        return out

    def field_slice(self, i):
        info = self.batch.info[i]
        return slice(info.point_offset, info.point_offset + info.n_main + info.n_head)


class Batch:
    """n independent fields planned together on one GPU (fcpp_batch_*)."""

    def __init__(self, specs, vehicle, options=None, device=None):
        self.ctx = get_context(device)
        self.lib = self.ctx.lib
        self.vehicle = vehicle
        self.options = options or make_options()
        self.n_fields = len(specs)
        import time
        t0 = time.perf_counter()
        arr, polys, _keep = pack_fields(specs)
        self.pack_ms = (time.perf_counter() - t0) * 1e3          # (0 for a FieldTable the caller already holds)
        self.ctx.bind_stream()
        h = C.c_void_p()
        L.check(self.lib.fcpp_batch_create(self.ctx.handle, C.byref(self.vehicle), C.byref(self.options),
                                           self.n_fields, arr, C.byref(polys), C.byref(h)))
        self.handle = h
        self._info = None
        self._token = object()      # marks the buffers alloc() makes for this batch
        tot = C.c_int64()
        L.check(self.lib.fcpp_batch_info(self.handle, None, C.byref(tot)))
        self.total_points = tot.value
        self._last_mode = 1

    @classmethod
    def plan(cls, specs, vehicle, options=None, device=None):
        """The reference's plan call for a whole batch in ONE library call (fcpp_batch_plan: plan_complete_coverage, MLP:387-465, sets a new
        field up and generates its path): batch creation, its output arrays (from the context's arena when it has one) and one step, with
        no Python between them.  -> (Batch, BatchResult), asynchronous like run(); the buffers of the result serve further run() calls."""
        torch = _torch()
        self = cls.__new__(cls)
        self.ctx = get_context(device)
        self.lib = self.ctx.lib
        self.vehicle = vehicle
        self.options = options or make_options()
        self.n_fields = len(specs)
        arr, polys, _keep = pack_fields(specs)
        self.pack_ms = 0.0
        self._info = None
        self._token = object()
        self._last_mode = 1
        self.ctx.bind_stream()
        h = C.c_void_p()
        ptrs = [C.c_void_p() for _ in range(5)]
        tot = C.c_int64()
        # (no statistics records of ours: the batch's own, inside its device allocation -- nothing is allocated in front of the call, and
        # whatever Python can do behind it -- the device object, the tensors over the arrays -- is done while the device works)
        L.check(self.lib.fcpp_batch_plan(self.ctx.handle, C.byref(self.vehicle), C.byref(self.options), self.n_fields, arr, C.byref(polys),
                                         None, C.byref(h), *[C.byref(q) for q in ptrs], C.byref(tot)))
        dev = torch.device('cuda', self.ctx.device)
        self.handle = h
        self.total_points = n = tot.value
        sp = C.c_void_p()
        L.check(self.lib.fcpp_batch_own_stats(h, C.byref(sp)))
        if self.n_fields > 0 and sp.value:
            # (a view of the batch's own memory: while a tensor made from it is alive, close() only marks the batch -- the tables are
            # released when the last such tensor is gone)
            sv = _StatsView(self, sp.value, self.n_fields * L.STATS_WORDS)
            stats = torch.as_tensor(sv, device=dev).view(self.n_fields, L.STATS_WORDS)
            stats._fcpp_owner = sv
            self._stats_views = 1
        else:
            stats = torch.empty((self.n_fields, L.STATS_WORDS), dtype=torch.int64, device=dev)
        owner = _ArenaArrays(self.ctx, n, [q.value or 0 for q in ptrs])
        if n > 0:
            x, y, kappa, v, fs = owner.tensors(n)
        else:
            x, y, kappa, v = (torch.empty(0, dtype=torch.float64, device=dev) for _ in range(4))
            fs = torch.empty(0, dtype=torch.int32, device=dev)
            x._fcpp_owner = owner
        lane, pitch = self.ctx.arena()
        self.layout = {'layout': 'arena' if (n > 0 and pitch and owner.ptrs[1] - owner.ptrs[0] == pitch) else 'plain', 'one_call': True}
        bufs = self._trusted((x, y, kappa, v, fs, stats))
        return self, BatchResult(self, *bufs)

    @property
    def info(self):
        """fcpp_field_info of every field (InfoTable).  A batch set up on the device keeps the records there until they are asked for:
        the first access copies them back (and waits for the batch's setup)."""
        if self._info is None:
            t = InfoTable(self.n_fields)
            L.check(self.lib.fcpp_batch_info(self.handle, t._c, None))
            self._info = t
        return self._info

    def setup_times(self):
        """Where the time of this batch's creation went, in ms: {'pack', 'host_plan', 'templates', 'tiler', 'image', 'h2d', 'create'
        (the fcpp_batch_create call), 'threads', 'image_bytes'}.  The plan call the reference times (plan_complete_coverage,
        MLP:387-465) = this + one run()."""
        t = L.SetupTimes()
        L.check(self.lib.fcpp_batch_setup_times(self.handle, C.byref(t)))
        return {'pack': self.pack_ms, 'host_plan': t.host_plan_ms, 'templates': t.templates_ms, 'tiler': t.tiler_ms, 'image': t.image_ms,
                'h2d': t.h2d_ms, 'create': t.total_ms, 'threads': int(t.threads), 'image_bytes': int(t.image_bytes),
                'device_setup': int(t.device_setup)}

    def setup_path(self):
        """'device' if this batch was set up on the GPU (fcpp_devplan), else 'host'"""
        t = L.SetupTimes()
        L.check(self.lib.fcpp_batch_setup_times(self.handle, C.byref(t)))
        return 'device' if t.device_setup else 'host'

    def debug_table(self, table):
        """One of the batch's device tables as bytes (fcpp_batch_debug_table; tests)."""
        nb = C.c_int64()
        L.check(self.lib.fcpp_batch_debug_table(self.handle, int(table), None, 0, C.byref(nb)))
        buf = np.zeros(max(nb.value, 1), dtype=np.uint8)
        L.check(self.lib.fcpp_batch_debug_table(self.handle, int(table), C.c_void_p(buf.ctypes.data), nb.value, C.byref(nb)))
        return buf[:nb.value]

    SPREAD_MIN_BYTES = 512 << 20        # batches with less output than this live in the caches: placement does not matter

    def alloc(self, layout='auto', best_of=1, include=()):
        """Output buffers (x, y, kappa, v, flagseg, stats) for run().

        layout: where the five arrays lie in device memory.  The hot kernels write them side by side, and on MI355X five write streams
        that lie within a few GiB of each other reach 4.6 TB/s where the same streams 12-24 GiB or more apart reach 6.3-6.6 TB/s
        (HISTORY.md; tools/placement_pitch.py: the speed class follows the pitch between the arrays, nothing else).
          'spread': ONE allocation, the arrays L.OUTPUT_PITCH (24 GiB) + their own size apart (less if the device has less room); the gaps
                    belong to the allocation -- a caller that needs them sub-allocates its own slab with the same rule.
          'plain' : five separate tensors, wherever the allocator puts them (usually back to back: the slow class).
          'arena' : from the context's output arena (Context.reserve_outputs(): one allocation made once, five lanes a pitch apart, shared
                    by all live batches) -- the spread placement without an allocation per batch.
          'auto'  : 'arena' when the context has one and the arrays are large enough to matter (>= 512 MiB of output), else 'plain'.
                    (Never 'spread': an allocation of ~100 GiB is not something a default should make.)
        self.layout tells which one was used and the pitch.

        best_of > 1 (opt-in, round 2's remedy): `best_of` further candidate sets are allocated ('plain'), the batch's own step is timed on
        each (and on the sets in `include`), the fastest is kept (`self.placement`).  Setup work; never inside a timed region; not what
        bench.py reports as its primary figures."""
        if best_of > 1 and self.total_points > 0:
            torch = _torch()
            import time
            dev = torch.device('cuda', self.ctx.device)
            sets, ms = list(include), []
            for _ in range(int(best_of)):
                try:
                    sets.append(self._alloc_once())
                except RuntimeError:          # out of device memory: choose among what we have
                    break
            if not sets:
                raise RuntimeError(f'out of device memory: no candidate set of output arrays ({36 * self.total_points / 2**30:.1f} GiB each) could be allocated')
            for s in sets:
                self.run(s)
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                for _ in range(3):
                    self.run(s)
                torch.cuda.synchronize(dev)
                ms.append((time.perf_counter() - t0) / 3 * 1e3)
            k = min(range(len(sets)), key=lambda i: ms[i])
            self.placement = {'probe': 'step', 'step_ms': [round(v, 4) for v in ms], 'chosen': k}
            keep = sets[k]
            del sets
            torch.cuda.empty_cache()
            return keep
        if layout not in ('auto', 'plain', 'spread', 'arena'):
            raise ValueError("layout: 'auto', 'plain', 'spread' or 'arena'")
        n = self.total_points
        lane, pitch = self.ctx.arena()
        if layout == 'arena' or (layout == 'auto' and lane >= 8 * n and 36 * n >= self.SPREAD_MIN_BYTES):
            # the context's arena (Context.reserve_outputs): array k in lane k, a pitch apart, shared with every other live batch
            if lane < 8 * n:
                raise RuntimeError('the context has no output arena of that size: Context.reserve_outputs()')
            torch = _torch()
            arr = _ArenaArrays(self.ctx, n)
            if arr.ptrs[1] - arr.ptrs[0] != pitch:        # (the arena was full: the library fell back to an allocation of its own)
                self.layout = {'layout': 'plain', 'note': 'the output arena is full'}
            else:
                self.layout = {'layout': 'arena', 'pitch_GiB': round(pitch / 2**30, 3), 'lane_GiB': round(lane / 2**30, 3)}
            x, y, kappa, v, fs = arr.tensors(n)
            stats = torch.empty((self.n_fields, L.STATS_WORDS), dtype=torch.int64, device=torch.device('cuda', self.ctx.device))
            return self._trusted((x, y, kappa, v, fs, stats))
        if layout in ('plain', 'auto'):
            # ('auto' never takes device memory the arrays do not need: the spread placement is the arena's, or an explicit layout='spread')
            self.layout = {'layout': 'plain'}
            return self._trusted(self._alloc_once())
        return self._trusted(self._alloc_spread(strict=True))

    def _trusted(self, buffers):
        """buffers alloc() has just made carry this batch's token: run() does not check them again (no reference to them is kept here:
        they go back to the allocator / the arena when the caller drops them)"""
        tok = self._token
        for t in buffers:
            t._fcpp_ok = tok
        return buffers

    def _alloc_spread(self, strict=False):
        torch = _torch()
        dev = torch.device('cuda', self.ctx.device)
        n = self.total_points
        S = (8 * n + 4095) // 4096 * 4096
        free, _total = torch.cuda.mem_get_info(dev)
        free += torch.cuda.memory_reserved(dev) - torch.cuda.memory_allocated(dev)       # (what the caching allocator could hand back)
        reserve = 8 << 30
        P = max(L.OUTPUT_PITCH, S + (1 << 30))
        if free < 4 * P + S + reserve:
            P = (free - reserve - S) // 4 if free > 5 * S + reserve else 0
        P = P // 4096 * 4096
        if P < S + (1 << 30) and not strict:       # no room to spread: the plain layout
            self.layout = {'layout': 'plain', 'note': 'no room for the spread layout'}
            return self._alloc_once()
        P = max(P, S)
        try:
            slab = torch.empty(4 * P + S, dtype=torch.uint8, device=dev)
        except RuntimeError:
            if strict:
                raise
            self.layout = {'layout': 'plain', 'note': 'the spread allocation failed'}
            return self._alloc_once()
        x, y, kappa, v = (slab[k * P: k * P + 8 * n].view(torch.float64) for k in range(4))
        fs = slab[4 * P: 4 * P + 4 * n].view(torch.int32)
        stats = torch.empty((self.n_fields, L.STATS_WORDS), dtype=torch.int64, device=dev)
        self.layout = {'layout': 'spread', 'pitch_GiB': round(P / 2**30, 3), 'allocation_GiB': round((4 * P + S) / 2**30, 3)}
        return x, y, kappa, v, fs, stats

    def _alloc_once(self):
        torch = _torch()
        dev = torch.device('cuda', self.ctx.device)
        n = self.total_points
        # (the four float64 arrays: rows of one allocation, each on a 512-byte boundary -- one call to the allocator instead of four)
        npad = (n + 63) // 64 * 64
        x, y, kappa, v = torch.empty((4, npad), dtype=torch.float64, device=dev)[:, :n].unbind(0)
        fs = torch.empty(n, dtype=torch.int32, device=dev)
        # (not cleared: fcpp_batch_run writes every field's row, zeros for a field that raised -- a fill kernel here would sit in the
        # stream in front of the step)
        stats = torch.empty((self.n_fields, L.STATS_WORDS), dtype=torch.int64, device=dev)
        return x, y, kappa, v, fs, stats

    def _check_buffers(self, buffers):
        """The library writes total_points elements into each array without looking at it again: sizes, types, device and layout
        are checked here (a short buffer would be an out-of-bounds device write)."""
        torch = _torch()
        x, y, kappa, v, fs, stats = buffers
        want = [('x', x, torch.float64, self.total_points), ('y', y, torch.float64, self.total_points),
                ('kappa', kappa, torch.float64, self.total_points), ('v', v, torch.float64, self.total_points),
                ('flagseg', fs, None, self.total_points), ('stats', stats, torch.int64, self.n_fields * L.STATS_WORDS)]
        for name, t, dtype, numel in want:
            if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device.index != self.ctx.device:
                raise ValueError(f'{name}: expected a tensor on cuda:{self.ctx.device}')
            if dtype is None:
                if t.dtype not in (torch.int32, torch.uint32):
                    raise ValueError(f'{name}: expected int32 / uint32, got {t.dtype}')
            elif t.dtype != dtype:
                raise ValueError(f'{name}: expected {dtype}, got {t.dtype}')
            if t.numel() < numel or not t.is_contiguous():
                raise ValueError(f'{name}: needs {numel} contiguous elements, got {t.numel()}')

    def run(self, buffers=None, mode=1):
        """Enqueue the hot path on torch's current stream; returns a BatchResult (asynchronous)."""
        if buffers is None:
            buffers = self.alloc()
        tok = self._token
        for t in buffers:
            if getattr(t, '_fcpp_ok', None) is not tok:
                self._check_buffers(buffers)
                break
        x, y, kappa, v, fs, stats = buffers
        self._last_mode = 1 if int(mode) >= 1 else 0
        self.ctx.bind_stream()
        L.check(self.lib.fcpp_batch_run(self.handle, _ptr(x), _ptr(y), _ptr(kappa), _ptr(v), _ptr(fs), _ptr(stats),
                                        int(mode)))
        return BatchResult(self, x, y, kappa, v, fs, stats)

    def connectors(self):
        """-> (approach, departure) tensors of shape (n_fields, 50, 2); rows of fields without a kept
        start/end point are NaN."""
        torch = _torch()
        dev = torch.device('cuda', self.ctx.device)
        ap = torch.full((self.n_fields, 50, 2), float('nan'), dtype=torch.float64, device=dev)
        dp = torch.full((self.n_fields, 50, 2), float('nan'), dtype=torch.float64, device=dev)
        self.ctx.bind_stream()
        L.check(self.lib.fcpp_batch_connectors(self.handle, _ptr(ap), _ptr(dp)))
        return ap, dp

    def set_profiling(self, on=True, every=1):
        """Per-kernel HIP-event timing of every `every`-th run() from now on (on=False: off)."""
        L.check(self.lib.fcpp_batch_set_profiling(self.handle, int(every) if on else 0))

    def stage_times(self):
        """-> ({kernel name: mean ms per run}, runs) from the HIP events recorded since the last call."""
        ms = (C.c_double * 16)()
        ns, nr = C.c_int(), C.c_int()
        L.check(self.lib.fcpp_batch_stage_times(self.handle, 16, ms, C.byref(ns), C.byref(nr)))
        runs = max(nr.value, 1)
        return {self.lib.fcpp_batch_stage_name(self._last_mode, k).decode(): ms[k] / runs for k in range(ns.value)}, nr.value

    def stage_points(self):
        """-> {kernel name: points one launch of it processes} for the pipeline of the last run()."""
        out = {}
        k = 0
        while True:
            name = self.lib.fcpp_batch_stage_name(self._last_mode, k).decode()
            if not name:
                break
            n = C.c_int64()
            L.check(self.lib.fcpp_batch_stage_points(self.handle, self._last_mode, k, C.byref(n)))
            out[name] = n.value
            k += 1
        return out

    def reduce_classes(self):
        """-> [paths whose statistics 8 lanes / a wavefront / a workgroup / 64 workgroups reduce] (fused pipeline)"""
        out = (C.c_int64 * 4)()
        L.check(self.lib.fcpp_batch_reduce_classes(self.handle, out))
        return list(out)

    def point_split(self):
        """-> (points handled by k_plan_quiet, points handled by k_plan_fused) in the fused pipeline."""
        q, g = C.c_int64(), C.c_int64()
        L.check(self.lib.fcpp_batch_point_split(self.handle, C.byref(q), C.byref(g)))
        return q.value, g.value

    def close(self):
        if getattr(self, 'handle', None):
            if getattr(self, '_stats_views', 0):        # (a result of plan() still reads the batch's own statistics records)
                self._close_pending = True
                return
            self.lib.fcpp_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


# ---- standalone operators on device tensors -------------------------------------------------------
def _dev_f64(a, device):
    torch = _torch()
    if isinstance(a, torch.Tensor):
        return a.to(device=device, dtype=torch.float64).contiguous()
    return torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64), device=device)


def _offsets(offsets, n, device):
    """-> (device tensor, host numpy copy or None): CSR offsets of the paths.  Host-side offsets (lists, numpy) are handed to the
    library as they are, so it need not read the device copy back."""
    torch = _torch()
    if offsets is None:
        offsets = [0, n]
    if isinstance(offsets, torch.Tensor):
        if offsets.is_cuda:
            return offsets.to(device=device, dtype=torch.int64).contiguous(), None
        offsets = offsets.numpy()
    host = np.ascontiguousarray(offsets, dtype=np.int64)
    return torch.as_tensor(host, device=device), host


def _host_ptr(a):
    return C.c_void_p(a.ctypes.data if a is not None else 0)


def curvature(x, y, offsets=None, device=None):
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    x, y = _dev_f64(x, dev), _dev_f64(y, dev)
    off, off_h = _offsets(offsets, x.numel(), dev)
    k = torch.empty_like(x)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_curvature(ctx.handle, off.numel() - 1, _ptr(off), x.numel(), _ptr(x), _ptr(y), _ptr(k), _host_ptr(off_h)))
    return k


def speed_plan(x, y, v, vehicle, clamp=True, offsets=None, device=None, want_kappa=False):
    """_apply_curvature_based_speed_limit (clamp=True) or _smooth_speed_profile only (clamp=False)."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    x, y, v = _dev_f64(x, dev), _dev_f64(y, dev), _dev_f64(v, dev)
    off, off_h = _offsets(offsets, x.numel(), dev)
    out = torch.empty_like(v)
    kap = torch.empty_like(v) if want_kappa else None
    nadj = torch.zeros(off.numel() - 1, dtype=torch.int64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_speed_plan(ctx.handle, C.byref(vehicle), int(bool(clamp)), off.numel() - 1, _ptr(off),
                                    x.numel(), _ptr(x), _ptr(y), _ptr(v), _ptr(out), _ptr(kap), _ptr(nadj), _host_ptr(off_h)))
    return (out, nadj, kap) if want_kappa else (out, nadj)


def verify(x, y, v, vehicle, offsets=None, device=None):
    """-> dict of numpy arrays per path (fcpp_verify)."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    x, y, v = _dev_f64(x, dev), _dev_f64(y, dev), _dev_f64(v, dev)
    off, off_h = _offsets(offsets, x.numel(), dev)
    n_paths = off.numel() - 1
    stats = torch.zeros((n_paths, L.STATS_WORDS), dtype=torch.int64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_verify(ctx.handle, C.byref(vehicle), n_paths, _ptr(off), x.numel(), _ptr(x), _ptr(y),
                                _ptr(v), _ptr(stats), _host_ptr(off_h)))
    raw = stats.cpu().numpy()
    out = {}
    for k, (n, _) in enumerate(L.FieldStats._fields_):
        col = raw[:, k]
        out[n] = col.view(np.float64).copy() if k < L.STATS_DOUBLES else col.copy()
    return out


def _polys(polygons):
    """list of vertex lists -> (L.Polys, keep-alive arrays)"""
    offs, px, py = [0], [], []
    for poly in polygons:
        for (a, b) in poly:
            px.append(float(a))
            py.append(float(b))
        offs.append(len(px))
    o, ax, ay = np.asarray(offs, dtype=np.int64), np.asarray(px, dtype=np.float64), np.asarray(py, dtype=np.float64)
    return L.Polys(len(offs) - 1, o.ctypes.data_as(L.c_i64_p), ax.ctypes.data_as(L.c_double_p), ay.ctypes.data_as(L.c_double_p)), [o, ax, ay]


def validate(x, y, v, vehicle, field_polygons=None, obstacles=None, obstacle_offsets=None, geofence_tol=1e-6, offsets=None, device=None):
    """fcpp_validate: lateral-acceleration / geofence / obstacle flags and per-path statistics of caller-supplied paths.
    field_polygons: one vertex list per path (or None); obstacles: vertex lists; obstacle_offsets: n_paths + 1 indices into them (None: every
    path against all).  -> (flags uint32 tensor, dict of numpy arrays per path)"""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    x, y, v = _dev_f64(x, dev), _dev_f64(y, dev), _dev_f64(v, dev)
    off, off_h = _offsets(offsets, x.numel(), dev)
    n_paths = off.numel() - 1
    opt = make_options(geofence_tol=geofence_tol)
    fp, keep1 = _polys(field_polygons) if field_polygons is not None else (None, None)
    ob, keep2 = _polys(obstacles) if obstacles else (None, None)
    oo = np.ascontiguousarray(obstacle_offsets, dtype=np.int64) if obstacle_offsets is not None else None
    flags = torch.empty(x.numel(), dtype=torch.int32, device=dev)
    stats = torch.zeros((n_paths, L.STATS_WORDS), dtype=torch.int64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_validate(ctx.handle, C.byref(vehicle), C.byref(opt), n_paths, _ptr(off), x.numel(), _ptr(x), _ptr(y), _ptr(v),
                                  C.byref(fp) if fp is not None else None, C.byref(ob) if ob is not None else None, _host_ptr(oo), _ptr(flags),
                                  _ptr(stats), _host_ptr(off_h)))
    raw = stats.cpu().numpy()
    out = {}
    for k, (n, _) in enumerate(L.FieldStats._fields_):
        col = raw[:, k]
        out[n] = col.view(np.float64).copy() if k < L.STATS_DOUBLES else col.copy()
    return flags, out


def straight_segments(segs, n_points, device=None):
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    segs = _dev_f64(segs, dev).reshape(-1, 4)
    out = torch.empty((segs.shape[0], int(n_points), 2), dtype=torch.float64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_straight_segments(ctx.handle, segs.shape[0], _ptr(segs), int(n_points), _ptr(out)))
    return out


def corner_turns(corners, corner_index, with_reverse, vehicle, field_length, field_width, device=None):
    """Corner turns as a batch (fcpp_corner_turns; MLP:1580-1608 and 1024-1084 / 1154-1288): for every corner the 15-point
    quarter arc and, where with_reverse is set, the tangent reverse fill toward the box [0, L] x [0, H].
    -> list of (turn (15, 2), reverse (n, 2) or None) numpy arrays."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    c = _dev_f64(corners, dev).reshape(-1, 2)
    n = int(c.shape[0])
    ci = torch.as_tensor(np.ascontiguousarray(corner_index, dtype=np.int32), device=dev)
    rv = torch.as_tensor(np.ascontiguousarray(with_reverse, dtype=np.int32), device=dev)
    stride = 15 + max(10, int(3.0 * vehicle.min_turn_radius / 0.5))
    out = torch.empty((n, stride, 2), dtype=torch.float64, device=dev)
    counts = torch.zeros((n, 2), dtype=torch.int32, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_corner_turns(ctx.handle, C.byref(vehicle), n, _ptr(c), _ptr(ci), _ptr(rv), float(field_length),
                                      float(field_width), stride, _ptr(out), _ptr(counts)))
    out, counts = out.cpu().numpy(), counts.cpu().numpy()
    res = []
    for k in range(n):
        nt, nr = int(counts[k, 0]), int(counts[k, 1])
        res.append((out[k, :nt].copy(), out[k, nt:nt + nr].copy() if nr else None))
    return res


def fresnel(t, device=None):
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    t = _dev_f64(t, dev)
    c, s = torch.empty_like(t), torch.empty_like(t)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_fresnel(ctx.handle, t.numel(), _ptr(t), _ptr(c), _ptr(s)))
    return c, s


def ga_fitness(routes, D, order_mode=0, device=None):
    """-> (distance, fitness) tensors for a population of tours (GA:168-181)."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    D = _dev_f64(D, dev)
    n = D.shape[0]
    if isinstance(routes, torch.Tensor):
        r = routes.to(device=dev, dtype=torch.int32).contiguous()
    else:
        r = torch.as_tensor(np.ascontiguousarray(routes, dtype=np.int32), device=dev)
    r = r.reshape(-1, n)
    dist = torch.empty(r.shape[0], dtype=torch.float64, device=dev)
    fit = torch.empty_like(dist)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_ga_fitness(ctx.handle, n, r.shape[0], _ptr(D), _ptr(r), _ptr(dist), _ptr(fit),
                                    int(order_mode)))
    return dist, fit


def distance_matrix(xy, device=None):
    """Centroid distance matrix (MVP:229-259, MFP:263-288) -> (n, n) float64 device tensor; row 0 = the depot by convention."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    p = _dev_f64(xy, dev).reshape(-1, 2)
    x, y = p[:, 0].contiguous(), p[:, 1].contiguous()
    n = int(x.shape[0])
    D = torch.empty((n, n), dtype=torch.float64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_distance_matrix(ctx.handle, n, _ptr(x), _ptr(y), _ptr(D)))
    return D


def best_connections(from_lists, to_lists, device=None):
    """Shortest exit -> entry connection for a batch of node pairs (MFP:290-320).  from_lists[p] / to_lists[p]: (k, 2) candidate
    points of pair p.  -> (index into from_lists[p], index into to_lists[p], distance) as numpy arrays (-1, -1, inf for empty lists)."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    n = len(from_lists)
    f = [np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in from_lists]
    t = [np.asarray(a, dtype=np.float64).reshape(-1, 2) for a in to_lists]
    fo = np.concatenate([[0], np.cumsum([len(a) for a in f])]).astype(np.int64)
    to = np.concatenate([[0], np.cumsum([len(a) for a in t])]).astype(np.int64)
    fxy = np.vstack(f) if n else np.zeros((0, 2))
    txy = np.vstack(t) if n else np.zeros((0, 2))
    d = lambda a: torch.as_tensor(np.ascontiguousarray(a), device=dev)
    fo_d, to_d, fx, fy, tx, ty = d(fo), d(to), d(fxy[:, 0]), d(fxy[:, 1]), d(txy[:, 0]), d(txy[:, 1])
    bf = torch.empty(n, dtype=torch.int32, device=dev)
    bt = torch.empty(n, dtype=torch.int32, device=dev)
    bd = torch.empty(n, dtype=torch.float64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_best_connections(ctx.handle, n, _ptr(fo_d), _ptr(to_d), _ptr(fx), _ptr(fy), _ptr(tx), _ptr(ty), _ptr(bf), _ptr(bt),
                                          _ptr(bd)))
    bf, bt = bf.cpu().numpy().astype(np.int64), bt.cpu().numpy().astype(np.int64)
    ok = bf >= 0
    return np.where(ok, bf - fo[:-1], -1), np.where(ok, bt - to[:-1], -1), bd.cpu().numpy()


def ga_evolve(D, routes, cfg, seed=0, device=None):
    """The GA's evolution loop on the device (fcpp_ga_evolve; GA:64-115, 183-268).  cfg: an object with GAConfig's attributes.
    -> (final population tensor (pop, n) int32, best_route tensor, best_fitness_history, avg_fitness_history (numpy), L.GaResult)"""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    D = _dev_f64(D, dev)
    n = D.shape[0]
    if isinstance(routes, torch.Tensor):
        r = routes.to(device=dev, dtype=torch.int32).contiguous().clone()
    else:
        r = torch.as_tensor(np.ascontiguousarray(routes, dtype=np.int32), device=dev)
    r = r.reshape(-1, n)
    c = L.GaConfig(int(r.shape[0]), int(cfg.max_generations), float(cfg.crossover_rate), float(cfg.mutation_rate), int(cfg.elite_size),
                   int(cfg.tournament_size), int(cfg.convergence_threshold), 0, int(seed) & 0xffffffffffffffff)
    best = torch.empty(n, dtype=torch.int32, device=dev)
    hist = torch.zeros(2 * max(c.max_generations, 1), dtype=torch.float64, device=dev)
    res = L.GaResult()
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_ga_evolve(ctx.handle, n, C.byref(c), _ptr(D), _ptr(r), _ptr(best), _ptr(hist), C.byref(res)))
    h = hist.cpu().numpy()
    g = res.generations
    return r, best, h[:g].copy(), h[c.max_generations:c.max_generations + g].copy(), res


# ---- coverage rasterisation (include/fcpp.h: fcpp_cover_grid; MLP:1357-1371, 1426-1509) ------------------------------
def half_planes(vertices):
    """12 doubles (a, b, c) x 4 for a convex quadrilateral: inside <=> a*x + b*y + c >= 0 for all four edges."""
    v = [(float(x), float(y)) for x, y in vertices]
    area2 = sum(v[i][0] * v[(i + 1) % 4][1] - v[(i + 1) % 4][0] * v[i][1] for i in range(4))
    sgn = 1.0 if area2 > 0 else -1.0
    out = []
    for i in range(4):
        (x0, y0), (x1, y1) = v[i], v[(i + 1) % 4]
        a, b = -(y1 - y0) * sgn, (x1 - x0) * sgn            # inward normal (not normalised: only the sign is used)
        out += [a, b, -(a * x0 + b * y0)]
    return out


NOWHERE = [0.0, 0.0, -1.0] * 4      # half-planes nobody is inside of (an empty inner polygon)


def make_cover_job(ox, oy, res, nx, ny, radius, n_a, n_b=0, pts_first=0, grid_first=-1, shift=0.0, strict=True, outer=None,
                   inner=None):
    j = L.CoverJob()
    j.ox, j.oy, j.res, j.shift, j.radius = float(ox), float(oy), float(res), float(shift), float(radius)
    j.nx, j.ny, j.n_a, j.n_b = int(nx), int(ny), int(n_a), int(n_b)
    j.pts_first, j.grid_first = int(pts_first), int(grid_first)
    j.strict, j.region = int(bool(strict)), int(outer is not None)
    j.outer[:] = list(outer) if outer is not None else [0.0] * 12
    j.inner[:] = list(inner) if inner is not None else NOWHERE
    return j


def cover_grid(jobs, px, py, want_grid=False, device=None):
    """Run a list of L.CoverJob over the device (or host) point arrays px, py.
    -> (counts int64 tensor (n_jobs, 3), grid uint8 tensor or None); a job's grid is grid[j.grid_first : + nx*ny].view(ny, nx)."""
    ctx = get_context(device)
    torch = _torch()
    dev = torch.device('cuda', ctx.device)
    px, py = _dev_f64(px, dev), _dev_f64(py, dev)
    n = len(jobs)
    total = 0
    for j in jobs:                      # (updated in place: grid_first tells the caller where each job's grid starts)
        j.grid_first = total if want_grid else -1
        total += j.nx * j.ny if want_grid else 0
    arr = (L.CoverJob * max(n, 1))(*jobs)
    grid = torch.zeros(total, dtype=torch.uint8, device=dev) if want_grid else None
    counts = torch.zeros((n, 3), dtype=torch.int64, device=dev)
    ctx.bind_stream()
    L.check(ctx.lib.fcpp_cover_grid(ctx.handle, n, arr, px.numel(), _ptr(px), _ptr(py), _ptr(grid) if want_grid and total else None,
                                    _ptr(counts)))
    return counts, grid
