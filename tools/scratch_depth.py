"""Dev: where do the scan kernels spill?  Lists every scratch_ instruction of the coarse_scan kernels in the
assembly (hipcc -S of csrc/aura_knn.hip) with the loop depth of its basic block: spills inside a tile loop
(depth >= 2) cost a vmcnt(0) drain per tile."""
import re, sys
lines = open(sys.argv[1]).read().split('\n')
pat = sys.argv[2] if len(sys.argv) > 2 else 'coarse_scan_kernel'
cur, depth, out = None, 0, {}
for i, l in enumerate(lines):
    if l.startswith('_Z') and ':' in l.split(';')[0]:
        cur, depth = l.strip(), 0
    if l.startswith('.LBB'):
        m = re.search(r'Depth=(\d)', l)
        depth = int(m.group(1)) if m else 0
    if 'scratch_' in l and cur and pat in cur:
        out.setdefault(cur[30:78], []).append(depth)
for k, v in out.items():
    print(k, 'scratch ops by loop depth:', {d: v.count(d) for d in sorted(set(v))})
