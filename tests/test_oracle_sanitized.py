"""The CPU oracle under AddressSanitizer + UBSan (GPU sanitizers are not available on the pool): the
oracle tests that do not need the compiled reference run once more against oracle/_san in a
subprocess with libasan preloaded; any out-of-bounds access or undefined behaviour fails the run."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_oracle_is_clean_under_asan_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "-s", "san"])
    asan = subprocess.check_output(["gcc", "-print-file-name=libasan.so"], text=True).strip()
    env = dict(os.environ)
    env.update(RNA_ORACLE_SO=os.path.join(ROOT, "oracle", "_san", "librna_oracle_san.so"), LD_PRELOAD=asan,
               ASAN_OPTIONS="detect_leaks=0:halt_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    tests = ["test_oracle_gridmap.py", "test_oracle_misc.py", "test_oracle_msgs.py", "test_oracle_scan.py"]
    out = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider"] +
                         [os.path.join(ROOT, "tests", t) for t in tests], env=env, capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "passed" in out.stdout and "runtime error" not in out.stderr
