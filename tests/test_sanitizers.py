"""AddressSanitizer + UndefinedBehaviorSanitizer over the host C++ that parses files and handles raw buffers: the CPU oracle and the
host mirror's readers / codecs, compiled together with -fsanitize=address,undefined (tests/sanitize/).  No GPU involved."""
import os
import subprocess

from product import ensure_built, ROOT

SAN = os.path.join(ROOT, "tests", "sanitize")


def test_oracle_and_host_mirror_under_asan_ubsan(tmp_path):
    ensure_built()
    subprocess.check_call(["make", "-C", SAN], stdout=subprocess.DEVNULL)
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:exitcode=66", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
    p = subprocess.run([os.path.join(SAN, "_build", "san_main"), os.path.join(ROOT, "tests", "golden"), str(tmp_path)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stdout[-3000:] + p.stderr[-6000:]
    assert "0 failure(s)" in p.stdout
