"""Print registers / spills / occupancy per kernel from a `hipcc -Rpass-analysis=kernel-resource-usage` log (stdin or file)."""
import re, subprocess, sys
t = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
for b in t.split('Function Name: ')[1:]:
    name = b.split()[0]
    def g(k):
        m = re.search(k + r': (\d+)', b)
        return m.group(1) if m else '?'
    dn = subprocess.run(['c++filt', name], capture_output=True, text=True).stdout.strip()
    dn = dn.replace('(anonymous namespace)::', '').replace('void ', '')[:80]
    print("%-82s V%4s A%4s spill %3s scratch %4s occ %s" % (dn, g('VGPRs'), g('AGPRs'), g('VGPR Spill'), g(r'ScratchSize \[bytes/lane\]'), g(r'Occupancy \[waves/SIMD\]')))
