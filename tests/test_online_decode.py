"""examples/decode-online: the reference dumps the picture after EVERY scan of a progressive file
(JPEG.Context hands the spectral image out between scans).  jpeg_amd_jpeg_decode_spectral_partial
stops after the first k scans; decoding those planes must give the reference's k-th dump."""
import ctypes as C
import hashlib

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib

ONLINE = G.manifest()["online"]


def _partial(k):
    lib = _lib.lib()
    data = np.fromfile(G.path(ONLINE["file"]), np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.full((info.units_y[c], info.units_x[c], 64), 9, np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral_partial(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                                     quanta.ctypes.data, C.byref(info), 1, k) == 0
    return info, planes, quanta


@pytest.mark.parametrize("snap", ONLINE["snapshots"], ids=lambda s: f"after-{s['after_scans']}-scans")
def test_partial_planes_decode_to_the_references_snapshot_on_the_cpu_path(snap):
    from oracle import oracle as O
    info, planes, quanta = _partial(snap["after_scans"])
    assert info.nscans == snap["after_scans"]
    factors = [(info.factor_x[c], info.factor_y[c]) for c in range(3)]
    _, rect = O.decode(planes, [quanta[c] for c in range(3)], factors, (info.width, info.height))
    rgb = O.unpack_rgb8(rect, 3)
    assert rgb.size == snap["rgb_nbytes"]
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == snap["rgb_sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("snap", ONLINE["snapshots"], ids=lambda s: f"after-{s['after_scans']}-scans")
def test_partial_decode_on_the_device(snap):
    import jpeg_amd as J
    ctx = J.default_context()
    rgb = J.Spectral.decompress(ctx, G.path(ONLINE["file"]), scans=snap["after_scans"]).decode(J.RGB).cpu().numpy()
    assert hashlib.sha256(rgb.tobytes()).hexdigest() == snap["rgb_sha256"]
