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


def test_worker_pool_is_clean_under_tsan(tmp_path):
    """The host thread pool of the batch file paths (csrc/worker_pool.hpp) under ThreadSanitizer: regions of every size and
    thread limit on pools of 1, 2, 5 and 16 threads -- every item exactly once, never more threads than the region was given,
    begin() / finish() apart -- and the directing-thread queue pattern of jpeg_amd_decompress_batch to its end."""
    exe = str(tmp_path / "worker_pool_test")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-g", "-fsanitize=thread", "-I", os.path.join(ROOT, "jpeg_amd", "csrc"),
                           os.path.join(ROOT, "tests", "cpp", "worker_pool_test.cpp"), "-o", exe, "-lpthread"])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().endswith("ok"), (r.stdout[-800:], r.stderr[-2000:])
