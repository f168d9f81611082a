"""VGPRs / spills / scratch / LDS of every kernel in a hipcc -S listing (amdhsa.kernels metadata).
usage: python tools/kernel_regs.py file.s [name filter]"""
import re
import subprocess
import sys

s = open(sys.argv[1]).read()
flt = sys.argv[2] if len(sys.argv) > 2 else ""
for b in s.split("  - .agpr_count")[1:]:
    name = re.search(r"\.name:\s+(\S+)", b).group(1)
    try:
        name = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip() or name
    except OSError:
        pass
    if flt and flt not in name:
        continue
    g = lambda k: re.search(r"\.%s:\s+(\d+)" % k, b).group(1)
    print("%-100s vgpr %3s spill %3s scratch %4s lds %6s" % (name[:100], g("vgpr_count"), g("vgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
