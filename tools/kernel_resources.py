"""Per-kernel register / spill / LDS figures of one HIP source, from the compiler's resource-usage remarks.

usage: python tools/kernel_resources.py polar-codes_amd/csrc/kernels_scl_reg.hip [substring filter] [extra compiler flags]
"""
import re, subprocess, sys, os

def main():
    src = sys.argv[1]
    filt = sys.argv[2] if len(sys.argv) > 2 else ""
    extra = sys.argv[3:]
    cmd = ["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", "-ffp-contract=off", "-D__HIP_PLATFORM_AMD__",
           "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"] + extra
    out = subprocess.run(cmd[:-4] + ["-c", os.path.basename(src), "-o", "/dev/null"] + extra, capture_output=True, text=True,
                         cwd=os.path.dirname(os.path.abspath(src))).stderr
    cur = None
    rows = {}
    for line in out.splitlines():
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[\w/]+\])?: (\S+) \[-Rpass", line)
        if not m:
            continue
        k, v = m.group(1).strip(), m.group(2)
        if k == "Function Name":
            cur = subprocess.run(["c++filt", v], capture_output=True, text=True).stdout.strip().replace("(anonymous namespace)::", "").split("(")[0]
            rows[cur] = {}
        elif cur:
            rows[cur][k] = v
    for name, r in rows.items():
        if filt in name:
            print(f"{name:60s} vgpr {r.get('VGPRs','?'):>4} agpr {r.get('AGPRs','?'):>3} spill {r.get('VGPRs Spill','?'):>4} scratch {r.get('ScratchSize','?'):>5} "
                  f"occ {r.get('Occupancy','?'):>2} lds {r.get('LDS Size','?'):>6}")

if __name__ == "__main__":
    main()
