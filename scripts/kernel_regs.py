"""Dev container: register / occupancy report of the search kernels of one dimension (hipcc -Rpass-analysis=kernel-resource-usage).
usage: hipcc ... -Rpass-analysis=kernel-resource-usage -c search_d<D>.hip 2> rpt; python scripts/kernel_regs.py rpt"""
import re
import sys
txt = open(sys.argv[1]).read()
for b in re.split(r'remark: [^\n]*Function Name: ', txt)[1:]:
    name = b.split('\n')[0]
    g = lambda k: (re.search(k + r': (\d+)', b) or [None, '?'])[1]
    m = re.search(r'search_kernelILi(\d+)ELb(\d)ELi(\d+)ELi(\d+)ELi(\d+)ELb(\d)ELi(\d+)ELb(\d)ELb(\d)', name)
    tag = ('D%s F%s K%s NCHR%-2s NW%-2s CB%s RB%-2s U8%s QB%s' % m.groups()) if m else name[:50]
    occ, scr = g(r'Occupancy \[waves/SIMD\]'), g(r'ScratchSize \[bytes/lane\]')
    print(f"{tag:50s} VGPR {g('VGPRs'):>3s} AGPR {g('AGPRs'):>3s} spill {g('VGPRs Spill'):>3s} SGPR {g('SGPRs'):>3s} occ {occ} scratch {scr}")
