"""HBM bytes per launch of one kernel from the PMC summaries of tools/profile_round.sh (pmc_p2.txt: FETCH_SIZE; pmc_p3.txt: WRITE_SIZE,
TCC_HIT_sum, TCC_MISS_sum) -> the JSON bench.py reads for `roofline.traffic`.

    python tools/traffic_json.py gpurun_out/prof_r03 "gemm_planes_x3_kernel" > profiles/r03_traffic.json
"""
import json
import sys


def table(path):
    rows = {}
    with open(path) as f:
        head = f.readline().split()
        for line in f:
            cells = line.rstrip("\n")
            name, rest = cells[:49].strip(), cells[49:].split()
            rows[name] = dict(zip(head[1:], [float(v) for v in rest]))
    return rows


def main():
    d, kernel = sys.argv[1], sys.argv[2]
    tag = d.rstrip("/").split("prof_")[-1]                 # gpurun_out/prof_r03_bf16 -> r03_bf16
    p2, p3 = table(d + "/pmc_p2.txt"), table(d + "/pmc_p3.txt")
    key = next(k for k in p2 if k.startswith(kernel[:48]))
    a, b = p2[key], p3[key]
    launches = int(a["disp"])
    fetch_kb, write_kb = a["FETCH_SIZE"], b["WRITE_SIZE"]
    out = {
        "kernel": kernel,
        "bytes_per_launch": round((2 * fetch_kb + write_kb) * 1024 / launches, -4),
        "launches": launches,
        "fetch_size_kb_sum": fetch_kb,
        "write_size_kb_sum": write_kb,
        "l2_hit_rate": round(b["TCC_HIT_sum"] / (b["TCC_HIT_sum"] + b["TCC_MISS_sum"]), 3),
        "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum in separate passes over `bench.py --steps 1 --warmup 1` "
                  f"(all steps of the process: the sums and the launch count come from the same pass; profiles/{tag}_pmc_p2.txt, {tag}_pmc_p3.txt); bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 / launches (gfx950: "
                  "FETCH_SIZE counts 64 B per 128-B request for 16-B-per-lane streaming reads, MI355X_MICROARCH.md section HBM)",
    }
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
