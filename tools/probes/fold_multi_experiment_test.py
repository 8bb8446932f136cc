"""dge_fold_multi (csrc/fold_multi.hip): every per-sample weight image of a synthesis pass - the reference's fused modulation,
stylegan2_generator.py:858-875 - from one launch, bit-identical to the single dge_pack_conv_pp / dge_pack_up_pp launches."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_grouped_fold_is_bit_identical_to_the_single_launches():
    from dge_amd import ops
    g = torch.Generator(device="cuda"); g.manual_seed(5)
    B = 8
    shapes = [(ops.FOLD_UP_PP, 512, 512), (ops.FOLD_CONV_PP, 512, 512), (ops.FOLD_UP_PP, 256, 512), (ops.FOLD_CONV_PP, 256, 256),
              (ops.FOLD_UP_PP, 128, 256), (ops.FOLD_CONV_PP, 128, 128), (ops.FOLD_UP_PP, 64, 128), (ops.FOLD_UP_PP, 96, 160)]
    ents, want = [], []
    for kind, cout, cin in shapes:
        w = torch.randn(cout, cin, 3, 3, device="cuda", generator=g)
        wscale = 1.0 / math.sqrt(9 * cin)
        s = (1.0 + 0.3 * torch.randn(B, cin, device="cuda", generator=g)).contiguous()
        d = (0.5 + torch.rand(B, cout, device="cuda", generator=g)).contiguous()
        if kind == ops.FOLD_UP_PP:
            wu = ops.pack_upconv_weight(w, ops.BF16, wscale)
            ents.append(dict(kind=kind, w=wu, cout=cout, cin=cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0)))
            want.append(ops.pack_up_pp(wu, cout, cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0)))
        else:
            ents.append(dict(kind=kind, w=w, wscale=wscale, cout=cout, cin=cin, in_scale=s, out_scale=d, gain=math.sqrt(2.0)))
            want.append(ops.pack_conv_pp(w, wscale, in_scale=s, out_scale=d, gain=math.sqrt(2.0)))
    got = ops.fold_multi(ents)
    torch.cuda.synchronize()
    assert len(got) == len(want)
    for (kind, cout, cin), a, b in zip(shapes, got, want):
        assert a.shape == b.shape and torch.equal(a.view(torch.int16), b.view(torch.int16)), (kind, cout, cin)
    # more entries than one table holds: split into several launches
    many = ops.fold_multi(ents + ents)
    assert len(many) == 2 * len(ents) and torch.equal(many[-1].view(torch.int16), want[-1].view(torch.int16))


def test_generator_pass_folds_all_of_its_layers_in_one_launch():
    """The 1024^2 generator at batch 8: the images handed to the layers are the ones the layers would fold themselves (same output)."""
    import dge_amd
    from dge_amd import autograd_s2
    from tests.golden import recipe as R
    from tests.helpers import s2_shapes
    G = dge_amd.StyleGAN2Generator(256, compute_dtype="bf16").cuda()
    G.load_state_dict(R.fill_s2(s2_shapes(256), seed=1))
    G.eval()
    wp = torch.randn(8, G.num_layers, 512, device="cuda")
    with torch.no_grad():
        a = G.synthesis(wp)["image"].clone()
        keep = autograd_s2._FOLD_MULTI
        autograd_s2._FOLD_MULTI = False
        try:
            b = G.synthesis(wp)["image"].clone()
        finally:
            autograd_s2._FOLD_MULTI = keep
    assert torch.equal(a, b)
