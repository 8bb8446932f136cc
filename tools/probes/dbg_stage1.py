import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))))
import torch
import dge_amd
from dge_amd.encoder import BE
from dge_amd.lpips import LPIPS
from dge_amd.e_align import EAlignStep
from tests.golden import recipe as R
from tests.helpers import s2_shapes, enc_shapes
from oracle import lpips_ref as LR
def build(stage=1):
    G = dge_amd.StyleGAN2Generator(64, fmaps_base=2048, fmaps_max=128, compute_dtype="f32").cuda()
    G.load_state_dict(R.fill_s2(s2_shapes(64, fmaps_base=2048, fmaps_max=128), seed=11)); G.eval()
    for p in G.parameters(): p.requires_grad_(False)
    E = BE(startf=16, maxf=64, layer_count=5, compute_dtype="f32").cuda()
    sd = R.fill_encoder(enc_shapes(16, 64, 5), seed=31)
    for k in sd:
        if "noise_weight" in k: sd[k] = torch.zeros_like(sd[k])
    E.load_state_dict(sd)
    for k, p in E.named_parameters():
        if "noise_weight" in k: p.requires_grad_(False)
    LP = LPIPS(compute_dtype="f32").cuda(); LP.load_state_dict(LR.seeded_params(0))
    return EAlignStep(G, E, LP, lr=0.0015, batch_size=2, stage=stage), E
for stage in (1, 2):
    a, Ea = build(stage); b, Eb = build(stage)
    la = [float(a.step(it)["loss_w"]) for it in range(4)]
    r = b.capture(warmup=1); lb = ["warm", float(r["loss_w"])]
    for _ in range(2): lb.append(float(b.replay()["loss_w"]))
    print("stage", stage, "eager", la, "graph", lb)
    k = "decode_block.0.conv_1.weight"
    print((Ea.state_dict()[k] - Eb.state_dict()[k]).abs().max().item(), Ea.state_dict()[k].abs().max().item())
a, Ea = build(1)
print("eager 0 then 2:", float(a.step(0)["loss_w"]), float(a.step(2)["loss_w"]), "then 3:", float(a.step(3)["loss_w"]))
a, Ea = build(1)
print("eager 0,1,3:", float(a.step(0)["loss_w"]), float(a.step(1)["loss_w"]), float(a.step(3)["loss_w"]))
