"""CPU-side code under AddressSanitizer and UBSan (no GPU sanitizer exists on this pool): the C oracle driven through
every entry point, and the host-only table builders of the GPU library (schedule compiler, critical sets, CRC maps)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ENV = dict(os.environ, ASAN_OPTIONS="verify_asan_link_order=0:detect_leaks=1", UBSAN_OPTIONS="print_stacktrace=1:halt_on_error=1")
ENV.pop("LD_PRELOAD", None)


def _run(make_dir, target, binary):
    out = subprocess.run(["make", "-C", os.path.join(ROOT, make_dir), target], capture_output=True, text=True)
    assert out.returncode == 0, out.stderr[-3000:]
    run = subprocess.run([os.path.join(ROOT, make_dir, binary)], capture_output=True, text=True, env=ENV, timeout=600)
    assert run.returncode == 0, (run.stdout[-2000:], run.stderr[-4000:])
    return run.stdout


def test_oracle_under_address_and_ub_sanitizers():
    assert "oracle sanitizer run ok" in _run("oracle", "sanitize", "_build/sanitize_oracle")


def test_host_table_builders_under_address_and_ub_sanitizers():
    assert "host tables sanitizer run ok" in _run(os.path.join("polar-codes_amd", "csrc"), "sanitize", "build/sanitize_host_tables")
