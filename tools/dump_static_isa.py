"""Compiles the specialised SC decoder for one code here (no GPU needed) and disassembles it.
Usage: python tools/dump_static_isa.py [N K design_snr_db] > isa.s
"""
import os
os.environ["POLAR_EXPERIMENT"] = "1"  # the library reads its switches only with this set
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "polar-codes_amd"), ROOT]
import functions as pcf  # noqa: E402
import _native  # noqa: E402

N, K, snr = (int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (1024, 512, 2.0)
with tempfile.TemporaryDirectory() as d:
    co = os.path.join(d, "static.co")
    os.environ["POLAR_STATIC_DUMP"] = co
    size = _native.static_compile_check(pcf.degaBuild(N, K, snr), "gfx950")
    sys.stderr.write("code object: %d bytes\n" % size)
    sys.stdout.write(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout)
    notes = subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-readelf", "--notes", co], capture_output=True, text=True).stdout
    for line in notes.splitlines():
        if any(k in line for k in (".vgpr_count", ".sgpr_count", ".vgpr_spill_count", ".group_segment_fixed_size", ".private_segment_fixed_size")):
            sys.stderr.write(line.strip() + "\n")
