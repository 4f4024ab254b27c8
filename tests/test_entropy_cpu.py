"""Host entropy decoder of libjpeg_amd.so -- CPU only, no GPU needed.

Every reference fixture (baseline, progressive, restart intervals; 4:2:0, 4:4:4, grey) must
decode to exactly the coefficient planes and quantisation tables recorded in
tests/golden/MANIFEST.json (which the gold-pinned oracle decode starts from)."""
import ctypes as C

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib


def _decode(path):
    lib = _lib.lib()
    data = np.fromfile(path, np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    planes = [np.full((info.units_y[c], info.units_x[c], 64), 77, np.int16) for c in range(info.ncomponents)]
    quanta = np.zeros((4, 64), np.uint16)
    info2 = _lib.FrameInfo()
    st = lib.jpeg_amd_jpeg_decode_spectral(data.ctypes.data, data.size, _lib.ptr_array([p.ctypes.data for p in planes]),
                                           quanta.ctypes.data, C.byref(info2))
    assert st == 0
    return info, planes, quanta


@pytest.mark.parametrize("name", G.decode_names())
def test_entropy_decoder_matches_manifest(name):
    e = G.entry(name)
    info, planes, quanta = _decode(G.path(e["file"]))
    assert (info.width, info.height, info.precision) == (e["width"], e["height"], e["precision"])
    assert ["baseline", "extended", "progressive"][info.process] == e["process"]
    assert info.nscans == e["scans"] and info.restart_interval == e["restart_interval"]
    n = info.ncomponents
    assert [info.id[c] for c in range(n)] == e["component_ids"]
    assert [[info.factor_x[c], info.factor_y[c]] for c in range(n)] == e["factors"]
    assert [[info.units_x[c], info.units_y[c]] for c in range(n)] == e["units"]
    assert [G.sha(p) for p in planes] == e["coef_sha256"]
    assert [quanta[c].tolist() for c in range(n)] == e["quanta_zigzag"]


@pytest.mark.parametrize("case", [c for c in G.encode_cases() if "file" in c], ids=lambda c: f"{c['mode']}-{c['level']}")
def test_entropy_decoder_on_encode_basic_files(case):
    """These files re-define DQT slot 0 between their two scans: the table must be the one in
    force at each component's first scan."""
    info, planes, quanta = _decode(G.path(case["file"]))
    assert [G.sha(p) for p in planes] == case["coef_sha256"]
    assert [quanta[c].tolist() for c in range(3)] == case["quanta_zigzag"]


def test_garbage_and_truncation_are_errors_not_crashes():
    lib = _lib.lib()
    info = _lib.FrameInfo()
    junk = np.arange(256, dtype=np.uint8)
    assert lib.jpeg_amd_jpeg_inspect(junk.ctypes.data, junk.size, C.byref(info)) != 0
    data = np.fromfile(G.path(G.entry("color-sequential-1.jpg")["file"]), np.uint8)
    for cut in (3, 100, 600, data.size // 2):
        part = np.ascontiguousarray(data[:cut])
        st = lib.jpeg_amd_jpeg_inspect(part.ctypes.data, part.size, C.byref(info))
        assert st in (0, _lib.EINVAL, _lib.ENOSUP)
