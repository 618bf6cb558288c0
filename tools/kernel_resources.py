"""Registers, spills, scratch, LDS and occupancy of every kernel of the library, from the
compiler's resource-usage remarks (compiles csrc/diffstereo.hip once more into /tmp).
    python3 tools/kernel_resources.py [substring of the kernel's name]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "stereo-vision_amd", "csrc")


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else ""
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off",
           "-fhip-fp32-correctly-rounded-divide-sqrt", "-Xarch_device", "-fno-honor-nans",
           "-I" + os.path.join(ROOT, "include"), "-c", "-o", "/tmp/ds_resources.o",
           os.path.join(CSRC, "diffstereo.hip"), "-Rpass-analysis=kernel-resource-usage"]
    err = subprocess.run(cmd, stderr=subprocess.PIPE, text=True).stderr
    rows, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(Function Name|VGPRs|AGPRs|VGPRs Spill|SGPRs Spill|ScratchSize \[bytes/lane\]|"
                      r"Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]): (\S+)", line)
        if not m:
            continue
        if m.group(1) == "Function Name":
            cur = {"name": m.group(2)}
            rows.append(cur)
        elif cur is not None:
            cur[m.group(1).split(" [")[0]] = m.group(2)
    for r in rows:
        name = subprocess.run(["c++filt", r["name"]], stdout=subprocess.PIPE, text=True).stdout.strip()
        name = name.split("(")[0].replace("void ds::", "")
        if want in name:
            print("%-62s vgpr %3s  spill %3s  sgpr-spill %3s  scratch %4s  lds %6s  waves/simd %s" % (
                name, r.get("VGPRs"), r.get("VGPRs Spill"), r.get("SGPRs Spill"), r.get("ScratchSize"),
                r.get("LDS Size"), r.get("Occupancy")))


if __name__ == "__main__":
    main()
