"""Per-kernel breakdown of the LAST training step in a rocprofv3 kernel trace CSV (step boundary = the AdamW launches).
usage: python scripts/trace_step.py <kernel_trace.csv> [--timeline]"""
import collections, csv, re, sys
rows = list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
idx = [i for i, r in enumerate(rows) if 'adamw_kernel' in r['Kernel_Name']]
end = idx[-1]
prev = [i for i in idx if i < end - 1][-1]
seg = rows[prev + 1:end + 1]
agg = collections.defaultdict(lambda: [0, 0.0])
busy, last_end = 0.0, None
for r in seg:
    s, e = int(r['Start_Timestamp']), int(r['End_Timestamp'])
    nm = re.sub(r'\(.*', '', r['Kernel_Name'].replace('void ', '').replace('lamp::', ''))[:52]
    agg[nm][0] += 1; agg[nm][1] += (e - s) / 1e3
    if last_end is None or s >= last_end: busy += e - s
    elif e > last_end: busy += e - last_end
    last_end = e if last_end is None else max(last_end, e)
    if '--timeline' in sys.argv:
        geo = ""
        if '--geometry' in sys.argv:      # grid (workgroups), workgroup size, LDS bytes, registers: what a same-grid copy kernel has to reproduce
            gx, wx = int(r.get('Grid_Size_X', 0) or 0), int(r.get('Workgroup_Size_X', 1) or 1)
            geo = f"  wgs {gx // max(wx, 1):6d} x {wx:4d}  lds {r.get('LDS_Block_Size', '?'):>6}  vgpr {r.get('VGPR_Count', '?'):>4}"
        print(f"{(s - int(seg[0]['Start_Timestamp'])) / 1e3:9.1f} {(e - s) / 1e3:8.1f}  {nm}{geo}")
span = (int(seg[-1]['End_Timestamp']) - int(seg[0]['Start_Timestamp'])) / 1e3
tot = sum(v[1] for v in agg.values())
print(f"launches {sum(v[0] for v in agg.values())}  kernel us {tot:.1f}  busy us {busy / 1e3:.1f}  span us {span:.1f}  idle {100 * (1 - busy / 1e3 / span):.1f} %")
for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]): print(f"{k:54s} {v[0]:3d} {v[1]:8.1f}")
