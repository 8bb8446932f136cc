#!/bin/bash
# PMC breakdown of up_s4 on one layer (dev tool):  tools/probes/pmc_up_s4.sh  -> gpurun_out/r06_pmc_up_s4.txt
cd /tmp && export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
cat > /tmp/one_up.py <<'PY'
import math, os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from dge_amd import ops
B, H, cin, cout = [int(v) for v in sys.argv[1:5]]
x = torch.randn(B, H, H, cin, device="cuda").bfloat16(); w = torch.randn(cout, cin, 3, 3, device="cuda")
s = 1 + 0.3 * torch.randn(B, cin, device="cuda"); d = 0.5 + torch.rand(B, cout, device="cuda")
nz = torch.randn(1, 2 * H, 2 * H, device="cuda"); nw = torch.full((1,), 0.3, device="cuda"); bias = torch.randn(cout, device="cuda")
wimg = ops.pack_up_pp(ops.pack_upconv_weight(w, ops.BF16, 1 / math.sqrt(9 * cin)), cout, cin, in_scale=s, out_scale=d, gain=2 ** 0.5)
for _ in range(4):
    ops.up_pp(x, wimg, cout, bias=bias, bias_scale=1.0, noise=nz, noise_w=nw, act=1, gain=2 ** 0.5)
torch.cuda.synchronize()
PY
out=$R/gpurun_out/r06_pmc_up_s4.txt
: > $out
for shape in "8 256 128 64" "8 32 512 512"; do
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_VALU" "SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS GRBM_GUI_ACTIVE"; do
  rm -rf /tmp/pmc_s4
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_s4 -o p --output-format csv -- python /tmp/one_up.py $shape > /tmp/pmc_s4.log 2>&1
  f=$(find /tmp/pmc_s4 -name "*counter_collection.csv" | head -1)
  python - "$f" "$shape" >> $out <<'PY'
import csv, sys, collections
acc = collections.defaultdict(float); n = collections.defaultdict(int)
for r in csv.DictReader(open(sys.argv[1])):
    if "up_s4" in r["Kernel_Name"]:
        acc[r["Counter_Name"]] += float(r["Counter_Value"]); n[r["Counter_Name"]] += 1
for k in acc:
    print(f"[{sys.argv[2]}] {k:28s} per launch {acc[k] / max(n[k], 1):16.0f}  ({n[k]} launches)")
PY
done
done
cat $out
