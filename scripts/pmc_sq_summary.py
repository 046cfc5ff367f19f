"""Per-kernel means of a rocprofv3 --pmc + --kernel-trace run (one counter group): counters per dispatch, dispatch duration from the kernel
trace, and - where the group has them - MFMA busy and wait shares.
  python scripts/pmc_sq_summary.py <rocprof dir>            -> one block per kernel (those that took >= 1 % of the GPU time)
  python scripts/pmc_sq_summary.py --table <gpurun_out/sq>   -> a table over the *_group1.txt / *_group2.txt files of scripts/sq_counters.sh
SQ_VALU_MFMA_BUSY_CYCLES counts matrix-pipe cycles summed over the chip's 1024 SIMDs; SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count
quad-cycles per wave (MI355X_MICROARCH.md, cycle constants); GRBM_GUI_ACTIVE sums the 8 XCDs."""
import csv, glob, os, re, sys
from collections import defaultdict

def short(k):
    k = re.sub(r"^void ", "", k)
    k = re.sub(r"\(.*$", "", k)
    return k.replace("lamp::", "")[:58]

def summarize(root):
    cnt = defaultdict(lambda: defaultdict(lambda: [0.0, 0]))
    for f in glob.glob(os.path.join(root, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            a = cnt[short(r["Kernel_Name"])][r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
    dur = defaultdict(lambda: [0.0, 0])
    for f in glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d = dur[short(r["Kernel_Name"])]
            d[0] += (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3; d[1] += 1
    total = sum(v[0] for v in dur.values()) or 1.0
    for k in sorted(dur, key=lambda k: -dur[k][0]):
        if dur[k][0] < 0.01 * total or k not in cnt:
            continue
        n, us = dur[k][1], dur[k][0] / dur[k][1]
        c = {name: s / m for name, (s, m) in cnt[k].items()}
        print(f"{k}\n  dispatches {n}  avg_us {us:.2f}  share_of_gpu_time {dur[k][0] / total:.3f}")
        for name in sorted(c):
            print(f"  {name:30s} {c[name]:16.0f}")
        if "SQ_WAVE_CYCLES" in c and c["SQ_WAVE_CYCLES"] > 0:
            w = c["SQ_WAVE_CYCLES"]
            print("  shares of wave cycles: " + ", ".join(f"{n2[3:].lower()} {c[n2] / w:.3f}" for n2 in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_WAIT_INST_LDS") if n2 in c))
        if "SQ_VALU_MFMA_BUSY_CYCLES" in c and "SQ_BUSY_CYCLES" in c and c["SQ_BUSY_CYCLES"] > 0:
            print(f"  mfma_busy_cycles / sq_busy_cycles {c['SQ_VALU_MFMA_BUSY_CYCLES'] / c['SQ_BUSY_CYCLES']:.3f}")
        if "GRBM_GUI_ACTIVE" in c:
            print(f"  clock_MHz_from_GRBM_GUI_ACTIVE {c['GRBM_GUI_ACTIVE'] / 8.0 / us:.0f}")

def parse_blocks(path):
    out, name = {}, None
    for line in open(path).read().split("\n"):
        if line and not line.startswith(" "):
            name = line; out[name] = {}
        elif name and line.strip():
            parts = line.split()
            try: out[name][parts[0]] = float(parts[-1])
            except ValueError: pass
    return out

NSQ = 32            # SQ_BUSY_CYCLES is summed over the chip's 8 XCDs x 4 shader engines (a 112 us GEMM at the 1.9 - 2.0 GHz its in-kernel clock shows: 30.2 kernel lengths)
NSIMD = 1024
FMAX_MHZ = 2400.0   # the part's maximum engine clock: an implied clock above it means the counter window was wider than the kernel

if sys.argv[1] == "--table":
    # the elapsed-cycle denominator comes from the SAME pass as the numerators - SQ_BUSY_CYCLES / 32 - instead of
    # GRBM_GUI_ACTIVE / 8 of another run (whose window is wider than a 5 - 40 us kernel: implied clocks of 3 - 5 GHz in round 4's table);
    # the clock it implies against the kernel-trace duration is printed and rows above the part's 2.4 GHz are refused; a lower bound of the
    # matrix-pipe share that needs no cycle counter at all (kernel-trace duration x 2.4 GHz) stands beside it.
    d = sys.argv[2]
    for f in sorted(glob.glob(os.path.join(d, "*_group1.txt"))):
        print("==", os.path.basename(f)[:-11])
        g2 = parse_blocks(f.replace("_group1.txt", "_group2.txt")) if os.path.exists(f.replace("_group1.txt", "_group2.txt")) else {}
        g1 = parse_blocks(f)
        text = open(f).read().split("\n")
        us_of = {}
        name = None
        for line in text:
            if line and not line.startswith(" "):
                name = line
            elif name and "avg_us" in line:
                us_of[name] = float(line.split("avg_us")[1].split()[0])
        name = None
        for line in text:
            if line and not line.startswith(" "):
                name = line
                c, us = g1.get(name, {}), us_of.get(name)
                if us and c.get("SQ_BUSY_CYCLES", 0) > 0:
                    elapsed = c["SQ_BUSY_CYCLES"] / NSQ
                    clock = elapsed / us                               # MHz
                    mf = c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / NSIMD
                    lower = mf / (us * FMAX_MHZ)
                    if clock > FMAX_MHZ * 1.02:
                        print(f"{name[:50]:50s} REFUSED: SQ_BUSY_CYCLES / {NSQ} implies {clock:.0f} MHz over a {us:.2f} us kernel (> {FMAX_MHZ:.0f}): no cycle-based share; "
                              f"mfma_pipe_busy >= {lower:.3f} of the duration at {FMAX_MHZ:.0f} MHz")
                    else:
                        extra = ""
                        if name in g2 and "SQ_ACTIVE_INST_LDS" in g2[name]:
                            extra = f", lds-instruction cycles per SIMD / elapsed {g2[name]['SQ_ACTIVE_INST_LDS'] * 4.0 / NSIMD / elapsed:.3f}"
                        print(f"{name[:50]:50s} mfma_pipe_busy / sq_busy cycles {mf / elapsed:.3f} (implied clock {clock:.0f} MHz; >= {lower:.3f} of the duration at {FMAX_MHZ:.0f} MHz){extra}")
                    if name in g2 and "GRBM_GUI_ACTIVE" in g2[name] and name in us_of:
                        gclk = g2[name]["GRBM_GUI_ACTIVE"] / 8.0 / us
                        flag = "  <- wider than the kernel: not used" if gclk > FMAX_MHZ * 1.02 else ""
                        print(f"{name[:50]:50s} cross-check: GRBM_GUI_ACTIVE / 8 of the second pass implies {gclk:.0f} MHz{flag}")
            elif "avg_us" in line or "shares of wave" in line or "SQ_VALU_MFMA_BUSY_CYCLES" in line or "SQ_INSTS_MFMA" in line:
                print(f"{name[:50]:50s} {line.strip()}")
else:
    summarize(sys.argv[1])
