"""The opt-in variants of the fused 4:2:0 decode stay bit-identical to the oracle: the band-walk kernel
(JPEG_AMD_BAND=1, kernels_band.hip), the register-prefetch luma kernel (JPEG_AMD_DIRECT=1), the part-pipelined launch
(JPEG_AMD_OVERLAP=1), the persistent chroma kernel (JPEG_AMD_K1_PERSIST=1), the per-XCD image partition of batches
(JPEG_AMD_XCD_IMAGES=1), the four-waves-per-SIMD luma kernel with the
chroma tile inside the coefficient buffer (JPEG_AMD_ALIAS=1) and the 16-row encode tiles (JPEG_AMD_ENC_TY=16).  They were built to answer VERDICT r01's questions, measured slower than the default path
 and are kept switchable; the switches are read once per process, hence the child processes."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, %r)
import jpeg_amd as J
from oracle import oracle as O
ctx = J.Context(0)
rng = np.random.default_rng(11)
quanta = [rng.integers(1, 30, 64).astype(np.uint16) for _ in range(2)]
import os
sizes = [(2048, 1540), (1000, 700), (520, 24), (17, 17), (4112, 520), (2048, 1024), (1920, 1080), (2064, 192), (640, 320), (128, 64)]   # QUAD-shaped: (2048, 1024); with JPEG_AMD_QUAD=2 also (1920, 1080) and (2064, 192); with =4 (1920, 1080), (640, 320), (128, 64) as stacks of two 16 x 4 strips
if os.environ.get("JA_TEST_BIG"):   # enough strips / blocks for the part pipeline and the persistent chroma kernel to engage
    sizes = [(8192, 6416), (1000, 700)]
for size in sizes:
    layout = J.Layout("ycc8", {1: J.Component((2, 2), 0), 2: J.Component((1, 1), 1), 3: J.Component((1, 1), 1)})
    units = layout.units(size)
    planes = [np.clip(rng.laplace(0, 30, (uy, ux, 64)), -1000, 1000).astype(np.int16) for ux, uy in units]
    spectral = J.Spectral.from_host(ctx, size, layout, planes, quanta, q=[0, 1, 1])
    _, rect = O.decode(planes, [quanta[0], quanta[1], quanta[1]], [(2, 2), (1, 1), (1, 1)], size, threads=8)
    assert (spectral.decode(J.RGB).cpu().numpy() == O.unpack_rgb8(rect, 3, threads=8)).all(), size
    assert (spectral.decode(J.YCbCr).cpu().numpy() == O.unpack_ycc8(rect, 3)).all(), size
    rgb = O.unpack_rgb8(rect, 3, threads=8)
    coef = J.Rectangular.encode(ctx, size, layout, rgb, {0: quanta[0], 1: quanta[1]}, J.RGB).host_planes()
    want = O.encode(rgb, size, [(2, 2), (1, 1), (1, 1)], [quanta[0], quanta[1], quanta[1]], threads=8)
    assert all((a == b).all() for a, b in zip(coef, want)), size
print("ok")
""" % ROOT


@pytest.mark.gpu
@pytest.mark.parametrize("switch", ["JPEG_AMD_BAND=1", "JPEG_AMD_DIRECT=1", "JPEG_AMD_OVERLAP=1", "JPEG_AMD_K1_PERSIST=1",
                                    "JPEG_AMD_ALIAS=1", "JPEG_AMD_QUAD=0", "JPEG_AMD_QUAD=2", "JPEG_AMD_QUAD=4", "JPEG_AMD_ENC_TY=16"])
def test_opt_in_path_matches_oracle(switch):
    k, v = switch.split("=")
    env = dict(os.environ)
    env[k] = v
    if k in ("JPEG_AMD_OVERLAP", "JPEG_AMD_K1_PERSIST"):
        env["JA_TEST_BIG"] = "1"
    p = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "ok" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]


@pytest.mark.gpu
def test_batch_with_every_image_on_one_xcd_matches_oracle():
    """JPEG_AMD_XCD_IMAGES=1 (k_luma_fused walks one strip list per residue of blockIdx.x mod 8) only engages for batches of
    32 images or more that fill the resident grid: run the 64 x 1080p slice of config 5 under it."""
    env = dict(os.environ)
    env["JPEG_AMD_XCD_IMAGES"] = "1"
    p = subprocess.run([sys.executable, "-m", "pytest", os.path.join(ROOT, "tests", "test_gpu_full_size.py"), "-m", "gpu", "-q", "-x",
                        "-k", "config5"], env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0 and " passed" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
