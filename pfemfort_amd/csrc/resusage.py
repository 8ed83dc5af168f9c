"""Summarise `make asm`'s resource_usage.txt (hipcc -Rpass-analysis=kernel-resource-usage)."""
import re, sys
t = open(sys.argv[1] if len(sys.argv) > 1 else 'resource_usage.txt').read()
for b in re.split(r'remark: [^\n]*Function Name: ', t)[1:]:
    name = b.split()[0]
    if 'rocprim' in name or 'hipcub' in name:
        continue
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    print("%-62s VGPR %4s AGPR %3s SGPR %3s scratch %4s occ %2s LDS %s" % (
        name[:62], g('VGPRs'), g('AGPRs'), g('TotalSGPRs'), g(r'ScratchSize \[bytes/lane\]'),
        g(r'Occupancy \[waves/SIMD\]'), g(r'LDS Size \[bytes/block\]')))
