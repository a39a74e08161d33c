#!/usr/bin/env python3
"""Reads tools/micro/wg_map's output: which workgroups share a SIMD, and how launch order maps to SIMDs."""
import collections
import sys
rows = [tuple(int(x) for x in ln.split()) for ln in open(sys.argv[1]) if ln.strip() and ln[0].isdigit()]
# HW_REG_HW_ID (gfx9): wave_id[3:0] simd_id[5:4] pipe_id[7:6] cu_id[11:8] sh_id[12] se_id[15:13] ...
def key(hw, xcc):
    return (xcc & 15, (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15, (hw >> 4) & 3)
simd = collections.defaultdict(list)
for i, hw, xcc, t0 in rows:
    simd[key(hw, xcc)].append(i)
print("workgroups", len(rows), "distinct SIMDs", len(simd), "waves per SIMD", collections.Counter(len(v) for v in simd.values()))
for blk in range(0, len(rows), 1024):
    ks = {key(hw, xcc) for i, hw, xcc, t0 in rows[blk:blk + 1024]}
    print("workgroups %d..%d land on %d distinct SIMDs" % (blk, blk + 1023, len(ks)))
ex = sorted(simd.items())[:6]
for k, v in ex:
    print("SIMD", k, "->", sorted(v))
# the stride structure: for the first SIMDs, differences between co-resident workgroup ids
d = collections.Counter()
for v in simd.values():
    v = sorted(v)
    for a, b in zip(v, v[1:]):
        d[b - a] += 1
print("id differences between workgroups sharing a SIMD:", d.most_common(8))
xs = collections.Counter(key(hw, xcc)[0] for i, hw, xcc, t0 in rows[:64])
print("first 64 workgroups by XCC:", dict(xs))
