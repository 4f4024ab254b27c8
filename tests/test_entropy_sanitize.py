"""The host entropy coder under AddressSanitizer + UndefinedBehaviorSanitizer (CPU build only --
GPU sanitizers are not available on the pool): every fixture decoded (1 and 4 threads), written
again as sequential and as progressive scans, decoded again; then 200 corrupted variants each."""
import glob
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_entropy_coder_is_clean_under_asan_and_ubsan(tmp_path):
    exe = str(tmp_path / "entropy_sanitize")
    src = [os.path.join(ROOT, "tests", "cpp", "entropy_sanitize.cpp"),
           os.path.join(ROOT, "jpeg_amd", "csrc", "entropy.cpp"), os.path.join(ROOT, "jpeg_amd", "csrc", "entropy_encode.cpp")]
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined",
                           "-I", os.path.join(ROOT, "include"), *src, "-o", exe, "-lpthread"])
    files = sorted(glob.glob(os.path.join(ROOT, "tests", "golden", "decode", "*.jpg")) +
                   glob.glob(os.path.join(ROOT, "tests", "golden", "encode", "*.jpg")))
    assert len(files) >= 25
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=0", UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([exe, *files], capture_output=True, text=True, timeout=1200, env=env)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-800:], r.stderr[-2000:])
