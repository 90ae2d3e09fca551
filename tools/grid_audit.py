"""Tile-count audit of a step from a rocprofv3 kernel trace (grid size, LDS, registers per dispatch):
    python tools/grid_audit.py gpurun_out/ab_trace_f32.csv
For every dispatch of the LAST step in the file: workgroups, the number a CU can hold (LDS, registers, 2048 threads), `rounds` = workgroups /
(256 CUs x that number), and the time it would take if the last, partly filled round cost only its share.  Lists the dispatches where a
short tail round wastes the most time -- the cases the round-4 changes to the bf16 convolution's position space were found with."""
import csv, sys, re, collections

def waves_by_regs(v):
    alloc = (v + 7) // 8 * 8
    return max(1, min(8, 512 // alloc)) if alloc > 0 else 8

rows = list(csv.DictReader(open(sys.argv[1])))
rows = [r for r in rows if r["Kind"] == "KERNEL_DISPATCH"]
# last step: from the last weight_jobs_kernel dispatch on
start = max(i for i, r in enumerate(rows) if "weight_jobs_kernel" in r["Kernel_Name"])
out = []
for r in rows[start:]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]); name = re.sub(r"\(.*", "", name).replace("void ", "")
    wg = int(r["Workgroup_Size_X"]) * int(r["Workgroup_Size_Y"]) * int(r["Workgroup_Size_Z"])
    grid = int(r["Grid_Size_X"]) * int(r["Grid_Size_Y"]) * int(r["Grid_Size_Z"])
    blocks = grid // wg
    lds = int(r["LDS_Block_Size"]); regs = int(r["VGPR_Count"]) + int(r["Accum_VGPR_Count"])
    waves = (wg + 63) // 64
    per_cu = min(2048 // wg, (4 * waves_by_regs(regs)) // waves if waves else 1, (160 * 1024) // lds if lds else 99)
    per_cu = max(per_cu, 1)
    slots = 256 * per_cu
    rounds = blocks / slots
    dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3
    full = int(rounds); frac = rounds - full
    ideal = dur * rounds / (full + (1 if frac > 0 else 0)) if rounds > 0 else dur
    out.append((dur - ideal, dur, name[:70], blocks, per_cu, rounds))
out.sort(reverse=True)
print(f"{'waste us':>9} {'dur us':>8} {'blocks':>7} {'/CU':>4} {'rounds':>7}  kernel")
for w, d, n, b, p, rd in out[:28]:
    print(f"{w:9.1f} {d:8.1f} {b:7d} {p:4d} {rd:7.2f}  {n}")
print("sum of the waste column:", round(sum(o[0] for o in out) / 1e3, 3), "ms of", round(sum(o[1] for o in out) / 1e3, 3))
