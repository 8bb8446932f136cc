#!/bin/bash
# HBM traffic and matrix-pipe utilisation of the conv launches (conv_igemm + conv_stream + upconv_fir) of bench.py's step from the
# PMC counters (run on the GPU box):
#   tools/pmc_traffic.sh [tag]      -> gpurun_out/<tag>_conv_traffic.json, <tag>_pmc_traffic_by_kernel.txt, <tag>_pmc_mfma_by_kernel.txt
# Separate --pmc passes (FETCH_SIZE and WRITE_SIZE do not fit one pass; the SQ / GRBM counters get their own), kernel-trace
# only, no other trace domains.  FETCH_SIZE is doubled (gfx950: 128-B requests tallied at 64 B, MI355X_MICROARCH.md "HBM");
# WRITE_SIZE is taken as reported (calibration: conv_stream 32->32 @1024^2 writes 537 MB by construction and the counter's
# TCC_EA0_WRREQ x 64 B gives 537 MB, profiles/r02_conv_stream_pmc_32x32_1024.txt).
TAG=${1:-r06}
cd /tmp && export TMPDIR=/tmp
# one stream: counter collection serialises kernels (a step with its side streams did not finish under --pmc), and the stage
# tables want every launch at its isolated duration
export DGE_SIDE_STREAMS=0
R=${GRAFT_REPO_ROOT:-/root/repo}
CMD="python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-synthesis --no-extras"
for c in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/pmc_$c
  rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c -o p --output-format csv -- $CMD > /tmp/pmc_$c.log 2>&1
done
rm -rf /tmp/pmc_MFMA
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -d /tmp/pmc_MFMA -o p --output-format csv -- $CMD > /tmp/pmc_MFMA.log 2>&1
mkdir -p $R/gpurun_out
python - "$R" "$TAG" <<'PY'
import csv, glob, json, sys, collections
R, TAG = sys.argv[1], sys.argv[2]
FAMILY = ("conv_igemm_kernel", "conv_pp_kernel", "conv_stream_kernel", "conv_small_kernel", "conv_pw_kernel", "upconv_fir_kernel", "upconv_stream_kernel", "up_s4_kernel", "up_pp_kernel")
tot = {}
per = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0.0]))
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"/tmp/pmc_{c}/**/*counter_collection.csv", recursive=True)[0]
    n, s = 0, 0.0
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] != c:
            continue
        k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        v = float(r["Counter_Value"])
        per[k][c][0] += 1; per[k][c][1] += v
        if any(f in k for f in FAMILY):
            n += 1; s += v
    tot[c] = (n, s)
nl = tot["FETCH_SIZE"][0]
fetch = 2.0 * tot["FETCH_SIZE"][1] * 1024 / nl        # KB -> bytes, x2 gfx950 correction
write = tot["WRITE_SIZE"][1] * 1024 / tot["WRITE_SIZE"][0]
out = {"kernel": "conv_igemm_kernel<*> + conv_pp_kernel<*> + conv_stream_kernel<*> + conv_small_kernel<*> + conv_pw_kernel<*> + up_s4_kernel + upconv_stream_kernel", "launches_profiled": nl, "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write,
       "traffic_bytes_per_launch": fetch + write, "correction": "FETCH_SIZE x2 (gfx950), WRITE_SIZE as reported",
       "command": "rocprofv3 --pmc {FETCH_SIZE|WRITE_SIZE} --kernel-trace -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline --no-synthesis"}
json.dump(out, open(f"{R}/gpurun_out/{TAG}_conv_traffic.json", "w"), indent=1)
print(json.dumps(out))
with open(f"{R}/gpurun_out/{TAG}_pmc_traffic_by_kernel.txt", "w") as fo:
    fo.write("# per-kernel PMC totals over the profiled run (3 steps): launches, FETCH_SIZE KB (uncorrected), WRITE_SIZE KB\n")
    for k, d in sorted(per.items(), key=lambda kv: -(kv[1]["FETCH_SIZE"][1] + kv[1]["WRITE_SIZE"][1])):
        fo.write(f"{d['FETCH_SIZE'][0]:6d} {d['FETCH_SIZE'][1]:14.0f} {d['WRITE_SIZE'][1]:14.0f}  {k[:140]}\n")
# ---- matrix-pipe utilisation per kernel: SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs) against the cycles the
#      kernel had: GRBM_GUI_ACTIVE is summed over the 8 XCDs, so (GUI_ACTIVE / 8) x 1024 SIMDs
f = glob.glob("/tmp/pmc_MFMA/**/*counter_collection.csv", recursive=True)[0]
acc = collections.defaultdict(lambda: collections.defaultdict(float))
cnt = collections.defaultdict(int)
for r in csv.DictReader(open(f)):
    k = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    if r["Counter_Name"] == "GRBM_GUI_ACTIVE":
        cnt[k] += 1
with open(f"{R}/gpurun_out/{TAG}_pmc_mfma_by_kernel.txt", "w") as fo:
    fo.write("# rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_WAVES --kernel-trace -- python bench.py --steps 2 --warmup 1 ...\n")
    fo.write("# per kernel, summed over the profiled launches (3 steps): launches, MFMA busy cycles, GUI_ACTIVE (sum over 8 XCDs),\n")
    fo.write("# mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs)\n")
    rows = sorted(acc.items(), key=lambda kv: -kv[1].get("GRBM_GUI_ACTIVE", 0.0))
    for k, d in rows[:60]:
        gui = d.get("GRBM_GUI_ACTIVE", 0.0)
        util = d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (gui / 8.0 * 1024.0) if gui > 0 else 0.0
        fo.write(f"{cnt[k]:6d} {d.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0):16.0f} {gui:14.0f}  util {util:6.3f}  {k[:120]}\n")
fam = {k: d for k, d in acc.items() if any(ff in k for ff in FAMILY)}
gui = sum(d.get("GRBM_GUI_ACTIVE", 0.0) for d in fam.values())
busy = sum(d.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) for d in fam.values())
print("conv family mfma_util", busy / (gui / 8.0 * 1024.0) if gui else None)
PY
head -40 $R/gpurun_out/${TAG}_pmc_mfma_by_kernel.txt
