"""jpeg_amd_decompress: file bytes -> pixels in one call (host entropy decoder
feeding the device path).  Must reproduce the reference's regression golds
(tests/regression/gold, examples/decode-*) byte for byte."""
import ctypes as C

import numpy as np
import pytest

import _golden as G
from jpeg_amd import _lib

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import jpeg_amd as J
    return J.Context()


def _decompress(ctx, path, color):
    lib = _lib.lib()
    data = np.fromfile(path, np.uint8)
    info = _lib.FrameInfo()
    assert lib.jpeg_amd_jpeg_inspect(data.ctypes.data, data.size, C.byref(info)) == 0
    out = np.zeros(info.width * info.height * 3, np.uint8)
    st = lib.jpeg_amd_decompress(ctx.handle, data.ctypes.data, data.size, 0, color.code, out.ctypes.data, out.size, C.byref(info))
    assert st == 0, st
    return out


@pytest.mark.parametrize("name", G.decode_names(gold_only=True))
def test_decompress_matches_reference_gold(ctx, name):
    import jpeg_amd as J
    e = G.entry(name)
    for key, color in (("rgb_sha256", J.RGB), ("ycc_sha256", J.YCbCr)):
        if key in e["gold"]:
            assert G.sha(_decompress(ctx, G.path(e["file"]), color)) == e["gold"][key]


@pytest.mark.parametrize("name", [n for n in G.decode_names() if "restart" in n])
def test_decompress_restart_files_match_oracle(ctx, name):
    import jpeg_amd as J
    from oracle import oracle as O
    img = G.image(name)
    _, rect = O.decode(img.planes, img.quanta, img.factors, (img.width, img.height))
    want = O.unpack_rgb8(rect, len(img.components))
    got = _decompress(ctx, G.path(G.entry(name)["file"]), J.RGB)
    assert (got.reshape(want.shape) == want).all()


def test_decompress_rejects_small_output_buffer(ctx):
    import jpeg_amd as J
    lib = _lib.lib()
    data = np.fromfile(G.path(G.entry("color-sequential-1.jpg")["file"]), np.uint8)
    out = np.zeros(16, np.uint8)
    st = lib.jpeg_amd_decompress(ctx.handle, data.ctypes.data, data.size, 0, J.RGB.code, out.ctypes.data, out.size, None)
    assert st == _lib.EINVAL


@pytest.mark.parametrize("name", ["color-progressive-2.jpg", "grayscale-sequential-1.jpg", "karlie-kwk-2019.jpg"])
def test_python_mirror_decompress(ctx, name):
    """Spectral.decompress / Rectangular.decompress (decode.swift:3728, 4367) from a path and from bytes."""
    import jpeg_amd as J
    e = G.entry(name)
    path = G.path(e["file"])
    rgb = J.Rectangular.decompress(ctx, path).unpack(J.RGB).cpu().numpy()
    assert G.sha(rgb) == e["gold"]["rgb_sha256"]
    fused = J.Spectral.decompress(ctx, open(path, "rb").read()).decode(J.RGB).cpu().numpy()
    assert (fused == rgb).all()


def test_decompress_batch_matches_single_image_calls(ctx):
    """jpeg_amd_decompress_batch: several files of one geometry, host threads + one fused launch."""
    import jpeg_amd as J
    lib = _lib.lib()
    names = ["color-sequential-1.jpg", "color-progressive-1.jpg"]          # same 319x480 4:2:0 frame
    files = [np.fromfile(G.path(G.entry(n)["file"]), np.uint8) for n in names] * 5
    n = len(files)
    info = _lib.FrameInfo()
    w, h = G.entry(names[0])["width"], G.entry(names[0])["height"]
    out = np.zeros((n, w * h * 3), np.uint8)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in files])
    sizes = (C.c_size_t * n)(*[f.size for f in files])
    st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, 3, 0, J.RGB.code, out.ctypes.data, 0, C.byref(info))
    assert st == 0, st
    for i in range(n):
        assert G.sha(out[i]) == G.entry(names[i % 2])["gold"]["rgb_sha256"]
    # a file of another geometry in the batch is a precondition failure
    other = np.fromfile(G.path(G.entry("karlie-2019.jpg")["file"]), np.uint8)
    ptrs2 = (C.c_void_p * 2)(files[0].ctypes.data, other.ctypes.data)
    sizes2 = (C.c_size_t * 2)(files[0].size, other.size)
    assert lib.jpeg_amd_decompress_batch(ctx.handle, ptrs2, sizes2, 2, 2, 0, J.RGB.code, out.ctypes.data, 0, None) == _lib.EINVAL


def test_two_contexts_on_two_threads():
    """A jpeg_amd_ctx is single-threaded, different contexts may run concurrently (SURVEY 8b,
    "Threading"): two host threads, each with its own context and stream, decode different files
    over and over; every result must be the gold."""
    import threading
    import jpeg_amd as J
    names = ["color-sequential-2.jpg", "grayscale-progressive-1.jpg"]
    errors = []

    def worker(name):
        try:
            c = J.Context(0, own_stream=True)
            e = G.entry(name)
            for _ in range(25):
                got = _decompress(c, G.path(e["file"]), J.RGB)
                if G.sha(got) != e["gold"]["rgb_sha256"]:
                    errors.append(name)
                    return
        except Exception as ex:   # noqa: BLE001
            errors.append(repr(ex))

    threads = [threading.Thread(target=worker, args=(n,)) for n in names]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors


def test_library_loaded_before_the_context_still_finds_the_device():
    """Host-only calls (jpeg_amd.inspect) load libjpeg_amd.so before any context exists; PyTorch
    brings its own HIP runtime and the two must agree -- in a fresh interpreter."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = ("import sys; sys.path.insert(0, %r); import jpeg_amd as J; "
            "i = J.inspect(%r); c = J.Context(0); print('ok', i.width)" % (root, G.path(G.entry("karlie-2019.jpg")["file"])))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "ok 640" in r.stdout, r.stderr[-400:]


def test_twelve_bit_four_component_file_through_the_staged_path(ctx):
    """examples/custom-color/output.jpg (written by the reference: 12-bit, 4 components, 16-bit
    quantisation tables, progressive): host entropy decode -> Spectral.idct() -> interleaved()
    on the device against the oracle (the reference keeps no gold for custom formats)."""
    import jpeg_amd as J
    from oracle import oracle as O
    path = G.path("encode/custom-color-output.jpg")
    spectral = J.Spectral.decompress(ctx, path)
    info = J.inspect(path)
    assert (info.precision, info.ncomponents, info.process) == (12, 4, 2)
    planes = spectral.host_planes()
    factors = [c.factor for c in spectral.layout.planes]
    planar = spectral.idct()
    want = [O.idct_plane(p, q, 12) for p, q in zip(planes, spectral.quanta)]
    for got, w in zip(planar.host_planes(), want):
        assert (got == w).all()
    rect = planar.interleaved(cosite=False).host_values()
    assert (rect == O.interleave(want, factors, spectral.layout.scale, spectral.size)).all()
    assert rect.max() > 255          # really more than 8 bits


@pytest.mark.parametrize("pinned", [False, True])
def test_decompress_batch_over_several_chunks_pageable_and_pinned_output(ctx, pinned):
    """70 files = three chunks of the pipeline (both pinned / device slots are reused); the output either in pageable memory
    (downloaded into the library's pinned slot and copied out by the host threads) or page-locked (downloaded straight into
    the caller's buffer), with a row pitch between the images."""
    import torch
    import jpeg_amd as J
    lib = _lib.lib()
    names = ["color-sequential-1.jpg", "color-progressive-1.jpg", "color-sequential-restart.jpg"]
    names = [n for n in names if (G.entry(n)["width"], G.entry(n)["height"]) == (G.entry(names[0])["width"], G.entry(names[0])["height"])]
    files = [np.fromfile(G.path(G.entry(n)["file"]), np.uint8) for n in names]
    n = 70
    batch = [files[i % len(files)] for i in range(n)]
    w, h = G.entry(names[0])["width"], G.entry(names[0])["height"]
    stride = w * h * 3 + 64
    holder = torch.zeros(n * stride, dtype=torch.uint8, pin_memory=pinned)
    out = holder.numpy()
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in batch])
    sizes = (C.c_size_t * n)(*[f.size for f in batch])
    for threads in (1, 5):
        out[:] = 0
        st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out.ctypes.data, stride, None)
        assert st == 0, st
        for i in range(n):
            assert G.sha(out[i * stride:i * stride + w * h * 3]) == G.entry(names[i % len(names)])["gold"]["rgb_sha256"], (i, threads)
            assert not out[i * stride + w * h * 3:(i + 1) * stride].any()


def test_sparse_coefficients_expand_to_the_planes_on_the_device(ctx):
    """jpeg_amd_jpeg_decode_sparse (host) + jpeg_amd_spectral_expand_batch (device) = the planes of
    jpeg_amd_jpeg_decode_spectral, for a batch of sequential fixtures of one geometry; a skipped image keeps its planes."""
    import torch
    from _sparse import sparse_decode
    lib = _lib.lib()
    names = ["grayscale-sequential-1.jpg", "grayscale-sequential-2.jpg", "grayscale-sequential-1.jpg"]   # one geometry (640 x 359)
    datas = [open(G.path(G.entry(n)["file"]), "rb").read() for n in names]
    datas.append(open(G.path(G.entry("color-sequential-1.jpg")["file"]), "rb").read())   # ... and a three-plane one, alone
    for group in (datas[:3], datas[3:]):
        _expand_on_device(ctx, lib, group)


def _expand_on_device(ctx, lib, datas):
    import torch
    from _sparse import sparse_decode
    info = _lib.FrameInfo()
    buf = (C.c_uint8 * len(datas[0])).from_buffer_copy(datas[0])
    assert lib.jpeg_amd_jpeg_inspect(buf, len(datas[0]), C.byref(info)) == 0
    nc = info.ncomponents
    units = [(info.units_x[c], info.units_y[c]) for c in range(nc)]
    blocks = sum(a * b for a, b in units)
    n = len(datas)
    cap = 64 * blocks
    desc = np.zeros((n, blocks), np.uint32)
    ent = np.zeros((n, cap), np.uint32)
    want = []
    for i, d in enumerate(datas):
        st, de, en, _q = sparse_decode(lib, d, info)
        assert st == 0
        desc[i] = de
        ent[i, :en.size] = en
        planes = [np.zeros((b, a, 64), np.int16) for a, b in units]
        b2 = (C.c_uint8 * len(d)).from_buffer_copy(d)
        q2 = np.zeros((4, 64), np.uint16)
        assert lib.jpeg_amd_jpeg_decode_spectral(b2, len(d), _lib.ptr_array([p.ctypes.data for p in planes]), q2.ctypes.data, None) == 0
        want.append(planes)
    dev = ctx.torch_device
    d_desc, d_ent = torch.from_numpy(desc.view(np.int32)).to(dev), torch.from_numpy(ent.view(np.int32)).to(dev)
    skip = torch.tensor([0] * (n - 1) + [1 if n > 1 else 0], dtype=torch.uint8, device=dev)
    coefs = [torch.full((n, 64 * a * b), 77, dtype=torch.int16, device=dev) for a, b in units]
    L = _lib.Layout()
    L.width, L.height, L.precision, L.nplanes = info.width, info.height, 8, nc
    L.scale_x, L.scale_y = info.scale_x, info.scale_y
    for c in range(nc):
        L.factor_x[c], L.factor_y[c], L.units_x[c], L.units_y[c], L.qi[c] = info.factor_x[c], info.factor_y[c], units[c][0], units[c][1], 0
    st = lib.jpeg_amd_spectral_expand_batch(ctx.handle, C.byref(L), n, d_desc.data_ptr(), blocks, d_ent.data_ptr(), cap, skip.data_ptr(),
                                            _lib.ptr_array([c.data_ptr() for c in coefs]), _lib.size_array([64 * a * b for a, b in units]))
    assert st == 0, st
    torch.cuda.synchronize()
    for i in range(n):
        for c in range(nc):
            got = coefs[c][i].cpu().numpy().reshape(want[i][c].shape)
            if i == n - 1 and n > 1:
                assert (got == 77).all()
            else:
                assert (got == want[i][c]).all(), (i, c)


def test_decompress_batch_device_leaves_the_pixels_on_the_device(ctx):
    """jpeg_amd_decompress_batch_device: 70 files (three chunks; sequential ones travel as sparse coefficients, the
    progressive one as planes -- both kinds in every chunk), the pixels stay in device memory, with a pitch."""
    import torch
    import jpeg_amd as J
    lib = _lib.lib()
    names = ["color-sequential-1.jpg", "color-progressive-1.jpg", "color-sequential-restart.jpg"]
    names = [n for n in names if (G.entry(n)["width"], G.entry(n)["height"]) == (G.entry(names[0])["width"], G.entry(names[0])["height"])]
    files = [np.fromfile(G.path(G.entry(n)["file"]), np.uint8) for n in names]
    n = 70
    batch = [files[i % len(files)] for i in range(n)]
    w, h = G.entry(names[0])["width"], G.entry(names[0])["height"]
    stride = w * h * 3 + 48
    out = torch.zeros(n * stride, dtype=torch.uint8, device=ctx.torch_device)
    ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in batch])
    sizes = (C.c_size_t * n)(*[f.size for f in batch])
    for threads in (1, 4):
        out.zero_()
        st = lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out.data_ptr(), stride, None)
        assert st == 0, st
        got = out.cpu().numpy()
        for i in range(n):
            assert G.sha(got[i * stride:i * stride + w * h * 3]) == G.entry(names[i % len(names)])["gold"]["rgb_sha256"], (i, threads)
            assert not got[i * stride + w * h * 3:(i + 1) * stride].any()


@pytest.mark.timeout(120)
@pytest.mark.parametrize("device_out", [False, True])
def test_a_bad_file_in_the_middle_of_a_batch_fails_the_call_and_leaves_the_context_usable(ctx, device_out):
    """70 files, the 41st cut off inside its frame header: the call reports the error (and returns: the threads waiting for
    later chunks are told to stop), and the next call on the same context -- same staging, same threads -- is right."""
    import torch
    import jpeg_amd as J
    lib = _lib.lib()
    name = "color-sequential-1.jpg"
    good = np.fromfile(G.path(G.entry(name)["file"]), np.uint8)
    bad = good[:40].copy()
    w, h = G.entry(name)["width"], G.entry(name)["height"]
    n = 70
    def call(files, threads):
        ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in files])
        sizes = (C.c_size_t * n)(*[f.size for f in files])
        if device_out:
            out = torch.zeros(n * w * h * 3, dtype=torch.uint8, device=ctx.torch_device)
            st = lib.jpeg_amd_decompress_batch_device(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out.data_ptr(), 0, None)
            return st, out.cpu().numpy()
        out = np.zeros(n * w * h * 3, np.uint8)
        st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, threads, 0, J.RGB.code, out.ctypes.data, 0, None)
        return st, out
    for threads in (1, 6):
        st, _ = call([bad if i == 40 else good for i in range(n)], threads)
        assert st == _lib.EINVAL
        st, out = call([good] * n, threads)
        assert st == 0
        for i in (0, 39, 40, 69):
            assert G.sha(out[i * w * h * 3:(i + 1) * w * h * 3]) == G.entry(name)["gold"]["rgb_sha256"]


def test_decompress_batch_without_a_single_helper_thread():
    """a process that cannot start one more thread (RLIMIT_NPROC reached) used to get ENOMEM for batches of more than
    two chunks; now the calling thread decodes every chunk itself in front of its submission.  In a fresh interpreter: context
    first (the runtime's own threads exist by then), then the limit, then 70 files = three chunks, pageable output (so the copy
    pool cannot start a thread either).  Root is exempt from the limit: skipped there."""
    import os, subprocess, sys
    if os.geteuid() == 0:
        pytest.skip("RLIMIT_NPROC does not bind root")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, resource, ctypes as C
sys.path.insert(0, %r); sys.path.insert(0, %r)
import numpy as np
import _golden as G
import jpeg_amd as J
from jpeg_amd import _lib
ctx = J.Context(0)
lib = _lib.lib()
names = ["color-sequential-1.jpg", "color-progressive-1.jpg", "color-sequential-restart.jpg"]
names = [n for n in names if (G.entry(n)["width"], G.entry(n)["height"]) == (G.entry(names[0])["width"], G.entry(names[0])["height"])]
files = [np.fromfile(G.path(G.entry(n)["file"]), np.uint8) for n in names]
n = 70
batch = [files[i %% len(files)] for i in range(n)]
w, h = G.entry(names[0])["width"], G.entry(names[0])["height"]
out = np.zeros(n * w * h * 3, np.uint8)
ptrs = (C.c_void_p * n)(*[f.ctypes.data for f in batch]); sizes = (C.c_size_t * n)(*[f.size for f in batch])
resource.setrlimit(resource.RLIMIT_NPROC, (1, resource.getrlimit(resource.RLIMIT_NPROC)[1]))   # no new thread from here on
import threading
try:
    t = threading.Thread(target=lambda: None); t.start(); t.join(); print("limit does not bind"); sys.exit(3)
except RuntimeError:
    pass
st = lib.jpeg_amd_decompress_batch(ctx.handle, ptrs, sizes, n, 8, 0, J.RGB.code, out.ctypes.data, 0, None)
assert st == 0, st
for i in range(n):
    assert G.sha(out[i * w * h * 3:(i + 1) * w * h * 3]) == G.entry(names[i %% len(names)])["gold"]["rgb_sha256"], i
print("ok")
''' % (root, os.path.join(root, "tests"))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=900)
    if r.returncode == 3:
        pytest.skip("RLIMIT_NPROC does not bind in this environment")
    assert r.returncode == 0 and "ok" in r.stdout, (r.stdout[-300:], r.stderr[-800:])
