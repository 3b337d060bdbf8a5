"""Register / LDS / scratch / occupancy table of every kernel of one translation unit, from hipcc's resource-usage remarks
(no GPU needed):  python tools/kernel_resources.py canonswap_amd/csrc/conv_halo.hip -DHALO_GROUP=2 [more flags]"""
import re
import subprocess
import sys

src, flags = sys.argv[1], sys.argv[2:]
cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", *flags,
       "-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
txt = subprocess.run(cmd, capture_output=True, text=True).stderr
blocks = re.split(r"remark: [^\n]*Function Name: ", txt)[1:]


def field(b, key):
    m = re.search(re.escape(key) + r": (\d+)", b)
    return m.group(1) if m else "?"


for b in blocks:
    name = b.split("\n")[0].strip().split()[0]
    name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    name = name.replace("void conv_halo_kernel", "halo").replace("(ConvParams)", "")
    print("%-52s VGPR %4s AGPR %4s SGPR %4s scratch %5s occ %2s LDS %s" % (
        name, field(b, "VGPRs"), field(b, "AGPRs"), field(b, "SGPRs"), field(b, "ScratchSize [bytes/lane]"),
        field(b, "Occupancy [waves/SIMD]"), field(b, "LDS Size [bytes/block]")))
