"""Sanitizer leg of the CPU oracle: every parity claim leans on oracle/*.c, so the
oracle-vs-fixture tests are run once more against `make -C oracle asan` builds (AddressSanitizer + UndefinedBehaviorSanitizer,
-O1 -g) in a child process with the ASan runtime preloaded and output capture off; any report fails the test."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
FILES = ["tests/test_oracle_pong_golden.py", "tests/test_pong_states_golden.py", "tests/test_oracle_car_golden.py",
         "tests/test_oracle_car_obs_golden.py", "tests/test_oracle_car_physics.py", "tests/test_oracle_float32_ref.py::test_float_gray_of_an_achromatic_pixel",
         "tests/test_car_wrappers_golden.py"]


def test_oracle_fixture_tests_are_clean_under_asan_and_ubsan():
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "asan"], stdout=subprocess.DEVNULL)
    asan = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True, check=True).stdout.strip()
    assert os.path.isabs(asan) and os.path.exists(asan), asan
    env = dict(os.environ, LD_PRELOAD=asan, CRL_ORACLE_SUFFIX="_asan",
               ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1",  # (CPython itself "leaks" at exit)
               UBSAN_OPTIONS="print_stacktrace=1")
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-s", "-m", "not gpu", "-p", "no:cacheprovider", *FILES],
                       cwd=ROOT, env=env, capture_output=True, text=True, timeout=1500)
    out = r.stdout + r.stderr
    for mark in ("runtime error", "AddressSanitizer", "UndefinedBehaviorSanitizer", "SUMMARY:"):
        assert mark not in out, out[-6000:]
    assert r.returncode == 0 and " passed" in r.stdout, out[-4000:]
    n = int(r.stdout.strip().splitlines()[-1].split(" passed")[0].split()[-1])
    assert n >= 28, r.stdout[-500:]
