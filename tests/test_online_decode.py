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


def test_stream_fed_in_4096_byte_pieces_yields_every_snapshot():
    """jpeg_amd_stream_*: the file arrives 4096 bytes at a time like in examples/decode-online
    (main.swift:28-31); whenever a push completes one more scan the snapshot must decode to the
    reference's dump for that scan -- here on the CPU path."""
    from oracle import oracle as O
    lib = _lib.lib()
    data = np.fromfile(G.path(ONLINE["file"]), np.uint8)
    s = C.c_void_p(lib.jpeg_amd_stream_create())
    try:
        done, fin, seen = C.c_int(0), C.c_int(0), 0
        info = _lib.FrameInfo()
        assert lib.jpeg_amd_stream_info(s, C.byref(info)) == _lib.EINVAL        # no frame header yet
        for lo in range(0, data.size, 4096):
            piece = np.ascontiguousarray(data[lo:lo + 4096])
            assert lib.jpeg_amd_stream_push(s, piece.ctypes.data, piece.size, C.byref(done), C.byref(fin)) == 0
            while seen < done.value:                 # (a push may complete more than one scan)
                seen += 1
                if seen != done.value:
                    continue                          # only the latest state can be snapshotted
                assert lib.jpeg_amd_stream_info(s, C.byref(info)) == 0
                planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(3)]
                quanta = np.zeros((4, 64), np.uint16)
                assert lib.jpeg_amd_stream_snapshot(s, _lib.ptr_array([p.ctypes.data for p in planes]), quanta.ctypes.data) == 0
                factors = [(info.factor_x[c], info.factor_y[c]) for c in range(3)]
                _, rect = O.decode(planes, [quanta[c] for c in range(3)], factors, (info.width, info.height))
                snap = ONLINE["snapshots"][seen - 1]
                assert hashlib.sha256(O.unpack_rgb8(rect, 3).tobytes()).hexdigest() == snap["rgb_sha256"], seen
        assert fin.value == 1 and done.value == len(ONLINE["snapshots"])
    finally:
        lib.jpeg_amd_stream_destroy(s)


@pytest.mark.parametrize("name", ["color-sequential-restart.jpg", "grayscale-progressive-2.jpg", "karlie-kwk-2019.jpg"])
@pytest.mark.parametrize("piece", [1, 7, 1000])
def test_stream_in_tiny_pieces_equals_the_one_shot_decoder(name, piece):
    lib = _lib.lib()
    e = G.entry(name)
    data = np.fromfile(G.path(e["file"]), np.uint8)
    s = C.c_void_p(lib.jpeg_amd_stream_create())
    try:
        done, fin = C.c_int(0), C.c_int(0)
        step = piece if piece > 7 or data.size < 40000 else 97 * piece    # byte-by-byte only for the small files
        for lo in range(0, data.size, step):
            part = np.ascontiguousarray(data[lo:lo + step])
            assert lib.jpeg_amd_stream_push(s, part.ctypes.data, part.size, C.byref(done), C.byref(fin)) == 0
        assert fin.value == 1 and done.value == e["scans"]
        info = _lib.FrameInfo()
        assert lib.jpeg_amd_stream_info(s, C.byref(info)) == 0
        planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(info.ncomponents)]
        quanta = np.zeros((4, 64), np.uint16)
        assert lib.jpeg_amd_stream_snapshot(s, _lib.ptr_array([p.ctypes.data for p in planes]), quanta.ctypes.data) == 0
        assert [G.sha(p) for p in planes] == e["coef_sha256"]
        assert [quanta[c].tolist() for c in range(info.ncomponents)] == e["quanta_zigzag"]
    finally:
        lib.jpeg_amd_stream_destroy(s)


def _in_memory():
    return G.manifest()["in_memory"]


def test_in_memory_example_gold_on_the_cpu_path():
    """examples/in-memory: decompress -> unpack of a progressive 4:4:4 file, dumped as .rgb."""
    from oracle import oracle as O
    m = _in_memory()
    lib = _lib.lib()
    data = np.fromfile(G.path(m["file"]), np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.zeros((info.units_y[c], info.units_x[c], 64), np.int16) for c in range(3)]
    quanta = np.zeros((4, 64), np.uint16)
    assert lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                             quanta.ctypes.data, None) == 0
    factors = [(info.factor_x[c], info.factor_y[c]) for c in range(3)]
    _, rect = O.decode(planes, [quanta[c] for c in range(3)], factors, (info.width, info.height))
    assert hashlib.sha256(O.unpack_rgb8(rect, 3).tobytes()).hexdigest() == m["rgb_sha256"]


@pytest.mark.gpu
def test_in_memory_example_gold_on_the_device():
    import jpeg_amd as J
    m = _in_memory()
    ctx = J.default_context()
    rgb = J.Rectangular.decompress(ctx, G.path(m["file"])).unpack(J.RGB).cpu().numpy()
    assert rgb.size == m["rgb_nbytes"] and hashlib.sha256(rgb.tobytes()).hexdigest() == m["rgb_sha256"]
    fused = J.Spectral.decompress(ctx, G.path(m["file"])).decode(J.RGB).cpu().numpy()
    assert (fused == rgb).all()
