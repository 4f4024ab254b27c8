"""Host-side mirror of the reference's hot-path interface over the C ABI.

Reference (tayloraswift/jpeg @ 2024_08_07, sources/jpeg/):
    JPEG.Data.Spectral.idct()                 decode.swift:4154
    JPEG.Data.Planar.interleaved(cosite:)     decode.swift:4182
    JPEG.Data.Rectangular.unpack(as:)         decode.swift:4294
    JPEG.Data.Rectangular.pack(size:layout:metadata:pixels:)   encode.swift:456
    JPEG.Data.Rectangular.decomposed()        encode.swift:389
    JPEG.Data.Planar.fdct(quanta:)            encode.swift:353
Same names, argument meaning and error behaviour (the reference's precondition failures
surface as JpegAmdError(EINVAL)).  PyTorch is used only to own device memory and the
stream; every computation is a call into libjpeg_amd.so.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np

from . import _lib

MAX_PLANES = _lib.MAX_PLANES


class YCbCr:
    """JPEG.YCbCr colour target (jpeg.swift:160-209, 481-540)."""
    code = _lib.COLOR_YCC8


class RGB:
    """JPEG.RGB colour target (jpeg.swift:210-269, 542-600)."""
    code = _lib.COLOR_RGB8


def _torch():
    import torch
    return torch


class Context:
    """One jpeg_amd_ctx bound to a device and to torch's current stream on it."""

    def __init__(self, device: int = 0, stream: Optional[int] = None, own_stream: bool = False):
        torch = _torch()
        if not torch.cuda.is_available():
            raise RuntimeError("jpeg_amd needs a GPU: torch.cuda.is_available() is False "
                               "(there is no CPU fallback)")
        self.device = int(device)
        self.torch_device = torch.device("cuda", self.device)
        if stream is None:
            stream = torch.cuda.current_stream(self.torch_device).cuda_stream
        self._h = C.c_void_p()
        flags = _lib.CTX_OWN_STREAM if own_stream else 0
        _lib.check(_lib.lib().jpeg_amd_ctx_create(self.device, C.c_void_p(stream), flags,
                                                 C.byref(self._h)), "jpeg_amd_ctx_create")

    @property
    def handle(self):
        return self._h

    def synchronize(self):
        _lib.check(_lib.lib().jpeg_amd_ctx_synchronize(self._h), "synchronize", self._h)

    def timer_begin(self):
        _lib.check(_lib.lib().jpeg_amd_timer_begin(self._h), "timer_begin", self._h)

    def timer_end(self) -> float:
        ms = C.c_float()
        _lib.check(_lib.lib().jpeg_amd_timer_end(self._h, C.byref(ms)), "timer_end", self._h)
        return ms.value

    def empty(self, n: int, dtype):
        torch = _torch()
        return torch.empty(max(int(n), 0), dtype=dtype, device=self.torch_device)

    def upload(self, a: np.ndarray):
        """numpy (int16 / uint16 / uint8) -> device tensor (uint16 travels as int16)."""
        torch = _torch()
        a = np.ascontiguousarray(a)
        if a.dtype == np.uint16:
            a = a.view(np.int16)
        return torch.from_numpy(a).to(self.torch_device)

    def close(self):
        if self._h:
            _lib.lib().jpeg_amd_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DEFAULT: Dict[int, Context] = {}


def default_context(device: int = 0) -> Context:
    if device not in _DEFAULT:
        _DEFAULT[device] = Context(device)
    return _DEFAULT[device]


@dataclass(frozen=True)
class Component:
    """JPEG.Component + its quanta key (jpeg.swift:1107-1160)."""
    factor: Tuple[int, int]
    qi: int


class Layout:
    """The part of JPEG.Layout<Format> the spectral pipeline reads (jpeg.swift:1084-1635).

    format: 'y8' | 'ycc8' | ('custom', precision, n_recognized)
    components: {component key: Component(factor, qi)} in plane order.  Components beyond
    the format's recognised count are non-recognised: they take part in `scale` only.
    """

    def __init__(self, format, components: Dict[int, Component | tuple]):
        self.format = format
        comps = {}
        for key, c in components.items():
            if not isinstance(c, Component):
                factor, qi = c
                c = Component(tuple(factor), int(qi))
            comps[key] = c
        self.components = comps
        if format == "y8":
            self.precision, nrec = 8, 1
        elif format == "ycc8":
            self.precision, nrec = 8, 3
        elif isinstance(format, tuple) and format[0] == "custom":
            self.precision, nrec = int(format[1]), int(format[2])
        else:
            raise ValueError(f"unknown format {format!r}")
        if nrec > len(comps) or nrec > MAX_PLANES:
            raise ValueError("format recognises more components than the layout holds")
        self.recognized = list(comps.keys())[:nrec]
        self.planes = [comps[k] for k in self.recognized]

    @property
    def scale(self) -> Tuple[int, int]:
        """decode.swift:2181-2190: max factor over ALL components."""
        return (max(c.factor[0] for c in self.components.values()),
                max(c.factor[1] for c in self.components.values()))

    @property
    def count(self) -> int:
        return len(self.planes)

    def units(self, size) -> List[Tuple[int, int]]:
        """decode.swift:2606-2616"""
        sx, sy = self.scale

        def u(n, s):
            return n // s + (1 if n % s else 0)
        return [(u(size[0] * c.factor[0], 8 * sx), u(size[1] * c.factor[1], 8 * sy))
                for c in self.planes]

    def c_layout(self, size, units=None, q: Optional[Sequence[int]] = None) -> _lib.Layout:
        L = _lib.Layout()
        L.width, L.height = int(size[0]), int(size[1])
        L.precision, L.nplanes = self.precision, self.count
        L.scale_x, L.scale_y = self.scale
        units = units if units is not None else self.units(size)
        for p, c in enumerate(self.planes):
            L.factor_x[p], L.factor_y[p] = c.factor
            L.units_x[p], L.units_y[p] = units[p]
            L.qi[p] = q[p] if q is not None else 0
        return L


def _ptrs(tensors):
    return _lib.ptr_array([t.data_ptr() if t is not None else None for t in tensors])


def _quanta_array(tables: Sequence[np.ndarray]):
    q = np.ascontiguousarray(np.stack([np.asarray(t, np.uint16).reshape(64) for t in tables]))
    return q, q.ctypes.data_as(C.c_void_p)


class Spectral:
    """JPEG.Data.Spectral<Format> (decode.swift:1370-1479): quantised coefficients,
    one int16 tensor [units_y, units_x, 64] (zigzag) per plane, resident in HBM."""

    def __init__(self, ctx: Context, size, layout: Layout, planes, quanta: Sequence[np.ndarray],
                 q: Sequence[int]):
        self.ctx, self.size, self.layout = ctx, (int(size[0]), int(size[1])), layout
        self.planes = list(planes)
        self.quanta = [np.asarray(t, np.uint16).reshape(64).copy() for t in quanta]
        self.q = list(q)                     # Plane.q: table index per plane
        if len(self.planes) != layout.count or len(self.q) != layout.count:
            raise ValueError("plane count does not match layout")

    @classmethod
    def from_host(cls, ctx, size, layout, planes: Sequence[np.ndarray], quanta, q=None):
        q = list(q) if q is not None else _dedupe_q(layout)
        dev = [ctx.upload(np.asarray(p, np.int16)) for p in planes]
        return cls(ctx, size, layout, dev, quanta, q)

    @property
    def units(self):
        return [(int(p.shape[1]), int(p.shape[0])) for p in self.planes]

    def _layout(self):
        return self.layout.c_layout(self.size, self.units, self.q)

    def idct(self) -> "Planar":
        """Spectral.idct() -- decode.swift:4154-4165."""
        torch = _torch()
        L = self._layout()
        out = [self.ctx.empty(64 * ux * uy, torch.int16).view(8 * uy, 8 * ux)
               for ux, uy in self.units]
        qarr, qptr = _quanta_array(self.quanta)
        _lib.check(_lib.lib().jpeg_amd_spectral_idct(
            self.ctx.handle, C.byref(L), _ptrs(self.planes), qptr, len(self.quanta),
            _ptrs(out)), "jpeg_amd_spectral_idct", self.ctx.handle)
        return Planar(self.ctx, self.size, self.layout, out)

    def rectangular(self, cosite: bool = False) -> "Rectangular":
        """Fused idct().interleaved(cosite:) (decode.swift:4154-4165, 4182-4276) -> Rectangular: any format (precision 1 .. 16,
        1 .. 4 planes); one launch with no Planar in HBM where the planes lie at the image's scale or at half of it, the staged
        kernels otherwise -- the same samples either way."""
        torch = _torch()
        L = self._layout()
        W, H = self.size
        out = self.ctx.empty(W * H * self.layout.count, torch.int16)
        qarr, qptr = _quanta_array(self.quanta)
        _lib.check(_lib.lib().jpeg_amd_spectral_rectangular(
            self.ctx.handle, C.byref(L), _ptrs(self.planes), qptr, len(self.quanta), 1 if cosite else 0, out.data_ptr()),
            "jpeg_amd_spectral_rectangular", self.ctx.handle)
        return Rectangular(self.ctx, self.size, self.layout, out.view(H, W, self.layout.count))

    def decode(self, color=RGB, cosite: bool = False):
        """Fused idct().interleaved(cosite:).unpack(as:) -> uint8 tensor [H*W, 3]."""
        torch = _torch()
        L = self._layout()
        out = self.ctx.empty(self.size[0] * self.size[1] * 3, torch.uint8)
        qarr, qptr = _quanta_array(self.quanta)
        _lib.check(_lib.lib().jpeg_amd_decode(
            self.ctx.handle, C.byref(L), _ptrs(self.planes), qptr, len(self.quanta),
            1 if cosite else 0, color.code, out.data_ptr()), "jpeg_amd_decode", self.ctx.handle)
        return out.view(-1, 3)

    def host_planes(self) -> List[np.ndarray]:
        return [p.cpu().numpy() for p in self.planes]

    def compress(self, scans, process: str = "baseline", metadata=None, path=None, restart_interval: int = 0) -> bytes:
        """Spectral.compress(stream:) / compress(path:) -- encode.swift:1918-1972, os.swift:330.
        scans: the layout's scan progression -- Scan objects, or plain lists of (plane index, dc
        selector, ac selector) for sequential scans; metadata: see _metadata_array.  The coefficient planes come back to the host and are entropy-coded
        there (csrc/entropy_encode.cpp).  Returns the file's bytes (and writes `path`)."""
        info = _lib.FrameInfo()
        info.width, info.height = self.size
        info.precision, info.ncomponents = self.layout.precision, self.layout.count
        info.process = {"baseline": 0, "extended": 1, "progressive": 2}[process]
        info.scale_x, info.scale_y = self.layout.scale
        info.restart_interval = int(restart_interval)   # extension: DRI + RSTm (0 = like the reference)
        keys = []
        for p, (key, comp) in enumerate(zip(self.layout.recognized, self.layout.planes)):
            info.id[p] = int(key)
            info.factor_x[p], info.factor_y[p] = comp.factor
            info.units_x[p], info.units_y[p] = self.units[p]
            keys.append(comp.qi)
        host = [np.ascontiguousarray(p.cpu().numpy()) for p in self.planes]
        # self.quanta[self.q[p]] is plane p's table; give every distinct quanta key one table
        tkeys = sorted(set(keys))
        tables = np.stack([self.quanta[self.q[keys.index(k)]] for k in tkeys]).astype(np.uint16)
        qkey = (C.c_int32 * len(keys))(*keys)
        tk = (C.c_int32 * len(tkeys))(*tkeys)
        sarr = _scan_array(scans)
        marr, nmeta, _keep = _metadata_array(metadata)
        n = C.c_size_t()
        args = [C.byref(info), qkey, _lib.ptr_array([h.ctypes.data for h in host]), tables.ctypes.data, tk, len(tkeys),
                sarr, len(scans), marr, nmeta]
        _lib.check(_lib.lib().jpeg_amd_jpeg_encode_spectral(*args, None, 0, C.byref(n)), "jpeg_amd_jpeg_encode_spectral")
        out = np.empty(n.value, np.uint8)
        _lib.check(_lib.lib().jpeg_amd_jpeg_encode_spectral(*args, out.ctypes.data, out.size, C.byref(n)),
                   "jpeg_amd_jpeg_encode_spectral")
        data = out.tobytes()
        if path is not None:
            with open(path, "wb") as f:
                f.write(data)
        return data

    @classmethod
    def decompress(cls, ctx: Context, source, scans: int = 0) -> "Spectral":
        """Spectral.decompress(stream:) / decompress(path:) -- decode.swift:3728, os.swift:309.
        source: a path or the file's bytes.  The entropy-coded segments are decoded on the host
        by the library (csrc/entropy.cpp); the coefficient planes land in HBM."""
        data = _file_bytes(source)
        info, planes, quanta = _decode_spectral(data, scans)   # scans > 0: the image after that many scans
        n = info.ncomponents
        if info.precision == 8 and n == 1:
            fmt = "y8"
        elif info.precision == 8 and n == 3:
            fmt = "ycc8"
        else:
            fmt = ("custom", info.precision, n)
        layout = Layout(fmt, {info.id[c]: Component((info.factor_x[c], info.factor_y[c]), c) for c in range(n)})
        return cls.from_host(ctx, (info.width, info.height), layout, planes, [quanta[c] for c in range(n)],
                             q=list(range(n)))


class Scan:
    """JPEG.Header.Scan constructors (jpeg.swift:1640-1760).  Components are PLANE INDICES
    (frame order); table selectors are 0..3."""

    def __init__(self, components, band=(0, 0), bit=0, refine=0):
        self.components, self.band, self.bit, self.refine = sorted(components), band, bit, refine

    @classmethod
    def sequential(cls, *components):
        """.sequential((c, dc, ac), ...)"""
        return cls(components)

    @classmethod
    def progressive_dc(cls, *components, bits):
        """.progressive((c, dc), ..., bits: bits...)"""
        return cls([(c, dc, 0) for c, dc in components], (0, 1), bits, 0)

    @classmethod
    def progressive_dc_refine(cls, *components, bit):
        """.progressive(c, ..., bit: bit)"""
        return cls([(c, 0, 0) for c in components], (0, 1), bit, 1)

    @classmethod
    def progressive_ac(cls, component, band, bits):
        """.progressive((c, ac), band: lo ..< hi, bits: bits...)"""
        return cls([(component[0], 0, component[1])], (max(band[0], 1), min(band[1], 64)), bits, 0)

    @classmethod
    def progressive_ac_refine(cls, component, band, bit):
        """.progressive((c, ac), band: lo ..< hi, bit: bit)"""
        return cls([(component[0], 0, component[1])], (max(band[0], 1), min(band[1], 64)), bit, 1)


def _scan_array(scans):
    """Scan objects, or plain [(plane index, dc selector, ac selector), ...] lists for
    sequential scans -> jpeg_amd_scan[]."""
    arr = (_lib.Scan * len(scans))()
    for i, sc in enumerate(scans):
        if not isinstance(sc, Scan):
            sc = Scan.sequential(*sc)
        arr[i].ncomponents = len(sc.components)
        for j, (c, dc, ac) in enumerate(sc.components):
            arr[i].component[j], arr[i].dc[j], arr[i].ac[j] = c, dc, ac
        arr[i].band_lo, arr[i].band_hi = sc.band
        arr[i].bit, arr[i].refine = sc.bit, sc.refine
    return arr


def _metadata_array(metadata):
    """[("jfif", (version_minor, unit, density_x, density_y)) | ("comment", bytes) |
    ("application", n, bytes), ...] -> (jpeg_amd_metadata[], keep-alive buffers) -- JPEG.Metadata."""
    metadata = list(metadata or [])
    arr = (_lib.Metadata * max(len(metadata), 1))()
    keep = []
    for i, m in enumerate(metadata):
        if m[0] == "jfif":
            arr[i].kind = 0
            arr[i].jfif.version_minor, arr[i].jfif.unit, arr[i].jfif.density_x, arr[i].jfif.density_y = m[1]
        else:
            data = np.frombuffer(bytes(m[-1]), np.uint8).copy()
            keep.append(data)
            arr[i].kind, arr[i].app = (2, 0) if m[0] == "comment" else (1, int(m[1]))
            arr[i].data, arr[i].size = data.ctypes.data, data.size
    return arr, len(metadata), keep


def _file_bytes(source) -> np.ndarray:
    if isinstance(source, np.ndarray):
        return np.ascontiguousarray(source, np.uint8)
    if isinstance(source, (bytes, bytearray, memoryview)):
        return np.frombuffer(bytes(source), np.uint8)
    return np.fromfile(source, np.uint8)       # a path


def inspect(source) -> _lib.FrameInfo:
    """Frame geometry of a JPEG file without decoding its scans (jpeg_amd_jpeg_inspect)."""
    data = _file_bytes(source)
    info = _lib.FrameInfo()
    _lib.check(_lib.lib().jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)), "jpeg_amd_jpeg_inspect")
    return info


def _decode_spectral(data: np.ndarray, scans: int = 0):
    info = inspect(data)
    planes = [np.empty((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((MAX_PLANES, 64), np.uint16)
    _lib.check(_lib.lib().jpeg_amd_jpeg_decode_spectral_partial(
        data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]), quanta.ctypes.data, None, 0, scans),
        "jpeg_amd_jpeg_decode_spectral_partial")
    return info, planes, quanta


def _dedupe_q(layout: Layout) -> List[int]:
    """Spectral.set(quanta:) (decode.swift:2510-2543): one table per distinct quanta key,
    in plane order."""
    keys: List[int] = []
    q = []
    for c in layout.planes:
        if c.qi not in keys:
            keys.append(c.qi)
        q.append(keys.index(c.qi))
    return q


class Planar:
    """JPEG.Data.Planar<Format> (decode.swift:1480-1598): one uint16 plane
    [8*units_y, 8*units_x] per component (stored as int16 bit patterns)."""

    def __init__(self, ctx, size, layout, planes):
        self.ctx, self.size, self.layout, self.planes = ctx, (int(size[0]), int(size[1])), layout, list(planes)

    @classmethod
    def from_host(cls, ctx, size, layout, planes: Sequence[np.ndarray]):
        return cls(ctx, size, layout, [ctx.upload(np.asarray(p, np.uint16)) for p in planes])

    @property
    def units(self):
        return [(int(p.shape[1]) // 8, int(p.shape[0]) // 8) for p in self.planes]

    def interleaved(self, cosite: bool = False) -> "Rectangular":
        """Planar.interleaved(cosite:) -- decode.swift:4182-4276."""
        torch = _torch()
        L = self.layout.c_layout(self.size, self.units)
        W, H = self.size
        out = self.ctx.empty(W * H * self.layout.count, torch.int16)
        _lib.check(_lib.lib().jpeg_amd_planar_interleaved(
            self.ctx.handle, C.byref(L), _ptrs(self.planes), 1 if cosite else 0,
            out.data_ptr()), "jpeg_amd_planar_interleaved", self.ctx.handle)
        return Rectangular(self.ctx, self.size, self.layout, out.view(H, W, self.layout.count))

    def fdct(self, quanta: Dict[int, Sequence[int]]) -> Spectral:
        """Planar.fdct(quanta:) -- encode.swift:353-370.  quanta: {quanta key: 64 zigzag values}."""
        torch = _torch()
        keys: List[int] = []
        q = []
        for c in self.layout.planes:
            if c.qi not in quanta:
                # decode.swift:2527-2530 preconditionFailure("missing quantization table ...")
                raise _lib.JpegAmdError(_lib.EINVAL, f"missing quantization table for quanta key {c.qi}")
            if c.qi not in keys:
                keys.append(c.qi)
            q.append(keys.index(c.qi))
        tables = [np.asarray(quanta[k], np.uint16).reshape(64) for k in keys]
        L = self.layout.c_layout(self.size, self.units, q)
        out = [self.ctx.empty(64 * ux * uy, torch.int16).view(uy, ux, 64) for ux, uy in self.units]
        qarr, qptr = _quanta_array(tables)
        _lib.check(_lib.lib().jpeg_amd_planar_fdct(
            self.ctx.handle, C.byref(L), _ptrs(self.planes), qptr, len(tables), _ptrs(out)),
            "jpeg_amd_planar_fdct", self.ctx.handle)
        return Spectral(self.ctx, self.size, self.layout, out, tables, q)

    def host_planes(self) -> List[np.ndarray]:
        return [p.cpu().numpy().view(np.uint16) for p in self.planes]


class Rectangular:
    """JPEG.Data.Rectangular<Format> (decode.swift:1650-1718): interleaved uint16 samples
    [H, W, count] (stored as int16 bit patterns)."""

    def __init__(self, ctx, size, layout, values):
        self.ctx, self.size, self.layout, self.values = ctx, (int(size[0]), int(size[1])), layout, values
        # decode.swift:1710-1712
        if values.numel() != layout.count * self.size[0] * self.size[1]:
            raise _lib.JpegAmdError(_lib.EINVAL, "array count does not match size and layout")
        if self.size[0] <= 0 or self.size[1] <= 0:
            raise _lib.JpegAmdError(_lib.EINVAL, "size must be positive")

    @property
    def stride(self) -> int:
        return self.layout.count

    @classmethod
    def from_host(cls, ctx, size, layout, values: np.ndarray):
        return cls(ctx, size, layout, ctx.upload(np.asarray(values, np.uint16)))

    def unpack(self, color=RGB):
        """Rectangular.unpack(as:) -- decode.swift:4291-4298 -> uint8 tensor [H*W, 3]."""
        torch = _torch()
        n = self.size[0] * self.size[1]
        out = self.ctx.empty(3 * n, torch.uint8)
        _lib.check(_lib.lib().jpeg_amd_rectangular_unpack(
            self.ctx.handle, self.values.data_ptr(), n, self.layout.count, color.code,
            out.data_ptr()), "jpeg_amd_rectangular_unpack", self.ctx.handle)
        return out.view(-1, 3)

    @classmethod
    def decompress(cls, ctx: Context, source, cosite: bool = False) -> "Rectangular":
        """Rectangular.decompress(stream:cosite:) -- decode.swift:4367-4374:
        Spectral.decompress(...).idct().interleaved(cosite:)."""
        return Spectral.decompress(ctx, source).rectangular(cosite=cosite)

    @classmethod
    def pack(cls, ctx, size, layout, pixels, color=RGB) -> "Rectangular":
        """Rectangular.pack(size:layout:metadata:pixels:) -- encode.swift:453-464.
        pixels: uint8 [H*W, 3] (numpy or device tensor)."""
        torch = _torch()
        if isinstance(pixels, np.ndarray):
            pixels = ctx.upload(np.asarray(pixels, np.uint8))
        n = int(size[0]) * int(size[1])
        if pixels.numel() != 3 * n:
            raise _lib.JpegAmdError(_lib.EINVAL, "array count does not match size and layout")
        out = ctx.empty(n * layout.count, torch.int16)
        _lib.check(_lib.lib().jpeg_amd_rectangular_pack(
            ctx.handle, pixels.data_ptr(), n, layout.count, color.code, out.data_ptr()),
            "jpeg_amd_rectangular_pack", ctx.handle)
        return cls(ctx, size, layout, out.view(int(size[1]), int(size[0]), layout.count))

    def decomposed(self) -> Planar:
        """Rectangular.decomposed() -- encode.swift:389-425."""
        torch = _torch()
        units = self.layout.units(self.size)
        L = self.layout.c_layout(self.size, units)
        out = [self.ctx.empty(64 * ux * uy, torch.int16).view(8 * uy, 8 * ux) for ux, uy in units]
        _lib.check(_lib.lib().jpeg_amd_rectangular_decomposed(
            self.ctx.handle, C.byref(L), self.values.data_ptr(), _ptrs(out)),
            "jpeg_amd_rectangular_decomposed", self.ctx.handle)
        return Planar(self.ctx, self.size, self.layout, out)

    def spectral(self, quanta: Dict[int, Sequence[int]]) -> Spectral:
        """decomposed().fdct(quanta:) in one call (encode.swift:389-425, 353-370): one launch for formats whose planes lie at the
        image's scale or at half of it (jpeg_amd_rectangular_spectral), the staged kernels otherwise -- same coefficients."""
        torch = _torch()
        keys: List[int] = []
        q = []
        for c in self.layout.planes:
            if c.qi not in quanta:
                raise _lib.JpegAmdError(_lib.EINVAL, f"missing quantization table for quanta key {c.qi}")
            if c.qi not in keys:
                keys.append(c.qi)
            q.append(keys.index(c.qi))
        tables = [np.asarray(quanta[k], np.uint16).reshape(64) for k in keys]
        units = self.layout.units(self.size)
        L = self.layout.c_layout(self.size, units, q)
        out = [self.ctx.empty(64 * ux * uy, torch.int16).view(uy, ux, 64) for ux, uy in units]
        qarr, qptr = _quanta_array(tables)
        _lib.check(_lib.lib().jpeg_amd_rectangular_spectral(
            self.ctx.handle, C.byref(L), self.values.data_ptr(), qptr, len(tables), _ptrs(out)),
            "jpeg_amd_rectangular_spectral", self.ctx.handle)
        return Spectral(self.ctx, self.size, self.layout, out, tables, q)

    @classmethod
    def encode(cls, ctx, size, layout, pixels, quanta: Dict[int, Sequence[int]], color=RGB) -> Spectral:
        """Fused pack(...).decomposed().fdct(quanta:) -> Spectral."""
        torch = _torch()
        if isinstance(pixels, np.ndarray):
            pixels = ctx.upload(np.asarray(pixels, np.uint8))
        keys: List[int] = []
        q = []
        for c in layout.planes:
            if c.qi not in quanta:
                raise _lib.JpegAmdError(_lib.EINVAL, f"missing quantization table for quanta key {c.qi}")
            if c.qi not in keys:
                keys.append(c.qi)
            q.append(keys.index(c.qi))
        tables = [np.asarray(quanta[k], np.uint16).reshape(64) for k in keys]
        units = layout.units(size)
        L = layout.c_layout(size, units, q)
        out = [ctx.empty(64 * ux * uy, torch.int16).view(uy, ux, 64) for ux, uy in units]
        qarr, qptr = _quanta_array(tables)
        _lib.check(_lib.lib().jpeg_amd_encode(
            ctx.handle, C.byref(L), pixels.data_ptr(), color.code, qptr, len(tables), _ptrs(out)),
            "jpeg_amd_encode", ctx.handle)
        return Spectral(ctx, size, layout, out, tables, q)

    def compress(self, quanta: Dict[int, Sequence[int]], scans, process: str = "baseline", metadata=None,
                 path=None, restart_interval: int = 0) -> bytes:
        """Rectangular.compress(stream:quanta:) / compress(path:quanta:) -- encode.swift:2031,
        os.swift:412: decomposed().fdct(quanta:).compress(...)."""
        return self.spectral(quanta).compress(scans, process=process, metadata=metadata, path=path,
                                              restart_interval=restart_interval)

    def host_values(self) -> np.ndarray:
        return self.values.cpu().numpy().view(np.uint16)

    # ---- the one-call forms of the C ABI for custom formats (host memory in, host memory out) --------------------------------
    @staticmethod
    def decompress_to_host(ctx: Context, source, cosite: bool = False, recognized: int = 0, threads: int = 0):
        """Rectangular<Format>.decompress(stream:cosite:) for any JPEG.Format in ONE call of the C ABI
        (jpeg_amd_decompress_rectangular; decode.swift:4367-4374): -> (frame info, uint16 array [H, W, n]).
        recognized: how many of the frame's components the format recognises (0: all)."""
        data = np.frombuffer(_file_bytes(source), np.uint8)
        info = _lib.FrameInfo()
        _lib.check(_lib.lib().jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)), "jpeg_amd_jpeg_inspect")
        n = recognized or info.ncomponents
        out = np.empty((info.height, info.width, n), np.uint16)
        _lib.check(_lib.lib().jpeg_amd_decompress_rectangular(
            ctx.handle, data.ctypes.data, data.size, 1 if cosite else 0, recognized, threads, out.ctypes.data, out.size,
            C.byref(info)), "jpeg_amd_decompress_rectangular", ctx.handle)
        return info, out

    @staticmethod
    def compress_from_host(ctx: Context, size, layout: "Layout", values, quanta: Dict[int, Sequence[int]], scans,
                           process: str = "baseline", metadata=None, path=None, restart_interval: int = 0) -> bytes:
        """Rectangular<Format>.compress(stream:quanta:) for any JPEG.Format in ONE call of the C ABI
        (jpeg_amd_compress_rectangular; encode.swift:2031).  values: uint16 [H, W, count] in host memory."""
        values = np.ascontiguousarray(np.asarray(values, np.uint16).reshape(int(size[1]), int(size[0]), layout.count))
        info = _lib.FrameInfo()
        info.width, info.height = int(size[0]), int(size[1])
        info.precision, info.ncomponents = layout.precision, layout.count
        info.process = {"baseline": 0, "extended": 1, "progressive": 2}[process]
        info.restart_interval = int(restart_interval)
        keys = []
        for p, (key, comp) in enumerate(zip(layout.recognized, layout.planes)):
            info.id[p] = int(key)
            info.factor_x[p], info.factor_y[p] = comp.factor
            keys.append(comp.qi)
        tkeys = sorted(set(keys))
        for k in tkeys:
            if k not in quanta:
                raise _lib.JpegAmdError(_lib.EINVAL, f"missing quantization table for quanta key {k}")
        tables = np.stack([np.asarray(quanta[k], np.uint16).reshape(64) for k in tkeys])
        qkey = (C.c_int32 * len(keys))(*keys)
        tk = (C.c_int32 * len(tkeys))(*tkeys)
        sarr = _scan_array(scans)
        marr, nmeta, _keep = _metadata_array(metadata)
        n = C.c_size_t()
        # the entropy coder sizes its output from the coefficients: a first call without a buffer would run the kernels twice,
        # so the buffer is sized generously instead (raw samples + headers) and the call repeated only if that was too small
        cap = values.size * 2 + (1 << 16)
        for _ in range(2):
            out = np.empty(cap, np.uint8)
            st = _lib.lib().jpeg_amd_compress_rectangular(ctx.handle, C.byref(info), values.ctypes.data, qkey, tables.ctypes.data, tk,
                                                          len(tkeys), sarr, len(scans), marr, nmeta, out.ctypes.data, out.size, C.byref(n))
            if st == _lib.EINVAL and n.value > cap:
                cap = n.value
                continue
            _lib.check(st, "jpeg_amd_compress_rectangular", ctx.handle)
            break
        data = out[:n.value].tobytes()
        if path is not None:
            with open(path, "wb") as f:
                f.write(data)
        return data


def compression_quanta(kind: str, level: float) -> np.ndarray:
    """JPEG.CompressionLevel.quanta (encode.swift:260-333): host-side constant tables,
    64 values in zigzag order.  kind: 'luminance' | 'chrominance'."""
    lum = [16, 11, 10, 16, 124, 140, 151, 161, 12, 12, 14, 19, 126, 158, 160, 155,
           14, 13, 16, 24, 140, 157, 169, 156, 14, 17, 22, 29, 151, 187, 180, 162,
           18, 22, 37, 56, 168, 109, 103, 177, 24, 35, 55, 64, 181, 104, 113, 192,
           49, 64, 78, 87, 103, 121, 120, 101, 72, 92, 95, 98, 112, 100, 103, 199]
    chr_ = [17, 18, 24, 47, 99, 99, 99, 99, 18, 21, 26, 66, 99, 99, 99, 99,
            24, 26, 56, 99, 99, 99, 99, 99, 47, 66, 99, 99, 99, 99, 99, 99] + [99] * 32
    key = lum if kind == "luminance" else chr_
    from .zigzag import ZIGZAG
    out = np.empty(64, np.uint16)
    for h in range(8):
        for k in range(8):
            v = 1.0 * (1 - level) + key[8 * h + k] * level
            v = float(np.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)  # .rounded(), Double
            out[ZIGZAG[h][k]] = int(max(1.0, min(v, 255.0)))
    return out
