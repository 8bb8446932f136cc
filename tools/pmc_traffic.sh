#!/bin/bash
# HBM traffic of the implicit-GEMM conv launches (conv_igemm + upconv_fir) of bench.py's step from the PMC counters (run on the GPU box):
#   tools/pmc_traffic.sh            -> gpurun_out/r01_conv_traffic.json (+ the raw per-kernel table)
# Two separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass), kernel-trace only, no other
# trace domains.  FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md "HBM").
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-synthesis"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- $CMD > /tmp/pmc_$c.log 2>&1
done
mkdir -p $R/gpurun_out
python - "$R" <<'PY'
import csv, glob, json, sys, collections
R = sys.argv[1]
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    n, s = 0, 0.0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        v = float(r["Counter_Value"])
        per[k][c][0] += 1; per[k][c][1] += v
        if "conv_igemm_kernel" in k or "upconv_fir_kernel" in k:
            n += 1; s += v
    tot[c] = (n, s)
nl = tot["FETCH_SIZE"][0]
fetch = 2.0 * tot["FETCH_SIZE"][1] * 1024 / nl        # KB -> bytes, x2 gfx950 correction
write = tot["WRITE_SIZE"][1] * 1024 / tot["WRITE_SIZE"][0]
out = {"kernel": "conv_igemm_kernel<*> + upconv_fir_kernel", "launches_profiled": nl, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported",
       "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-synthesis"}
json.dump(out, open(f"{R}/gpurun_out/r01_conv_traffic.json", "w"), indent=1)
print(json.dumps(out))
with open(f"{R}/gpurun_out/r01_pmc_traffic_by_kernel.txt", "w") as fo:
    fo.write("# per-kernel PMC totals over the profiled run (3 steps): launches, FETCH_SIZE KB (uncorrected), WRITE_SIZE KB\n")
    for k, d in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1])):
        fo.write(f"{d['FETCH_SIZE'][0]:6d} {d['FETCH_SIZE'][1]:14.0f} {d['WRITE_SIZE'][1]:14.0f}  {k[:140]}\n")
PY
