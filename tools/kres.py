"""Summarise `hipcc -Rpass-analysis=kernel-resource-usage` remarks (file) per kernel."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for b in re.split(r"remark: Function Name: ", txt)[1:]:
    name = b.split()[0]
    if pat and not re.search(pat, name):
        continue

    def g(k):
        return re.search(re.escape(k) + r": (\d+)", b).group(1)

    print("%-64s VGPR %3s AGPR %3s SGPR %3s scratch %4s occ %s spill %s" % (
        name[-64:], g("VGPRs"), g("AGPRs"), g("TotalSGPRs"), g("ScratchSize [bytes/lane]"),
        g("Occupancy [waves/SIMD]"), g("VGPRs Spill")))
