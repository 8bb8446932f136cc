"""Real-image inversion loop of the reference (embedding_img.py:74-170, BASELINE config 5) on the HIP path:
StyleGAN1 synthesis Gs + E_Blur encoder, per image group the encoder is re-loaded and fine-tuned for `iterations`
two-phase steps:

    const2, w1 = E(imgs1); imgs2 = Gs.forward(w1, lod); const3, w2 = E(imgs2)                       (:86-88)
    loss_msiv = loss_imgs + 0.125*(loss_medium + loss_small)   [both crops detached: value only]    (:92-111)
    backward(retain_graph=True); step
    loss_msLv = 0.01*(space_loss(w1, w2) + space_loss(const2, const3)); backward; step               (:116-127)

Every gradient path of the reference is live: through Gs into w1, through the second encoder call into its input image
(and on through Gs), through both encoder outputs (w and the 4x4 const).  `--optimizeE False` is not offered: in the
reference it re-uses the graph of a single E(imgs1) call across iterations and fails in the second one, and the CLI flag
(`type=bool`) cannot be switched off anyway (:190).
"""
import collections
import math

import torch

from . import losses
from .custom_adam import LREQAdam


class EmbedStep:
    def __init__(self, Gs, E, lpips_model, lr=0.01, beta_1=0.0):
        self.Gs, self.E, self.lpips = Gs, E, lpips_model
        self.opt = LREQAdam([{"params": E.parameters()}], lr=lr, betas=(beta_1, 0.99), weight_decay=0)
        self.lod = Gs.layer_count - 1
        self._ckpt = {k: v.detach().clone() for k, v in E.state_dict().items()}
        self.last = {}

    def begin_image(self):
        """embedding_img.py:82-83: reload the encoder checkpoint and clear the optimizer state for every image group."""
        self.E.load_state_dict(self._ckpt)
        for p in self.E.parameters():
            p._dge_gen = getattr(p, "_dge_gen", 0) + 1           # packed-weight caches key on this counter
        if getattr(self.opt, "_graph_corr", None) is not None:
            self.opt.graph_reset()                               # same device addresses: a captured graph stays valid
        else:
            self.opt.state = collections.defaultdict(dict)

    # ------------------------------------------------------------------ hipGraph replay of the iteration
    def capture(self, imgs1, noises=(None, None, None), warmup=2):
        """Captures one iteration (≈1700 kernel launches, all on the current stream through the C ABI) into a hipGraph.
        At batch 1 the eager loop is bound by host launch overhead (31 ms/iteration against ≈8 ms of GPU work at 1024^2);
        `replay()` re-runs the captured iteration on the static input `imgs1` with one graph launch.  The only host-side
        quantity that changes between iterations, Adam's sqrt(1 - beta2^t), is read from a device scalar that
        `LREQAdam.graph_advance` refreshes before each replay.  Noise is drawn inside the graph by the counter-based
        generator (its seed is a device scalar refreshed per replay) unless static `noises` are given.  `warmup` real iterations run here; the
        captured one is only recorded."""
        from . import ops
        dev = imgs1.device
        self._g_imgs1 = imgs1.detach().clone()
        self.opt.graph_begin(2, dev)
        ops.noise_graph_begin(dev)            # the captured noise kernels read their seed from a device scalar
        self._noise_it = 0
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(warmup):
                self._graph_inputs()
                self.step(self._g_imgs1, noises)
        torch.cuda.current_stream().wait_stream(side)
        from .autograd_enc import prime_pack_tables
        prime_pack_tables(self.E)             # (the all-copies descriptor table must be on the device before the capture)
        self._graph = torch.cuda.CUDAGraph()
        snap, nit = self.opt.graph_snapshot(), self._noise_it       # (the captured iteration is recorded, not executed)
        self._graph_inputs()
        with torch.cuda.graph(self._graph):
            self._g_out = self.step(self._g_imgs1, noises)
        self.opt.graph_restore(snap)
        self._noise_it = nit
        return self._g_out

    def _graph_inputs(self):
        """host-side inputs of one captured iteration: Adam's step factors and the noise seed (a new one per iteration)"""
        from . import ops
        self.opt.graph_advance()
        base = ops.NOISE.seed if getattr(self, "_noise_base", None) is None else self._noise_base
        self._noise_base = base if base is not None else 0
        self._noise_it += 1
        ops.noise_seed(self._noise_base + self._noise_it)

    def set_image(self, imgs1):
        """Makes `imgs1` the image group the captured iteration works on (the graph reads the static buffer capture() cloned
        the FIRST group into; every later group has to be copied there before its replays)."""
        if getattr(self, "_graph", None) is None:
            raise RuntimeError("EmbedStep.set_image: no captured iteration (call capture() first)")
        if tuple(imgs1.shape) != tuple(self._g_imgs1.shape) or imgs1.dtype != self._g_imgs1.dtype or imgs1.device != self._g_imgs1.device:
            raise ValueError(f"EmbedStep.set_image: the captured iteration works on {tuple(self._g_imgs1.shape)} {self._g_imgs1.dtype} on "
                             f"{self._g_imgs1.device}, got {tuple(imgs1.shape)} {imgs1.dtype} on {imgs1.device}; re-capture for a new geometry")
        self._g_imgs1.copy_(imgs1.detach())

    def replay(self):
        self._graph_inputs()
        self._graph.replay()
        self.opt.graph_count_replay()
        return self._g_out

    def step(self, imgs1, noises=(None, None, None)):
        """One iteration; `noises` = optional (E(imgs1), Gs, E(imgs2)) noise lists for parity runs."""
        from . import ops
        E, Gs = self.E, self.Gs
        ops.zero_arena_begin(imgs1.device)
        const2, w1 = E(imgs1, noises=noises[0])
        imgs2 = Gs.forward(w1, self.lod, noises=noises[1])
        const3, w2 = E(imgs2, noises=noises[2])
        loss_msiv, info_img = losses.image_loss_tsa(imgs1, imgs2, self.lpips, weights=(1.0, 0.125, 0.125),
                                                    grad_windows=(True, False, False))
        self.opt.zero_grad()
        loss_msiv.backward(retain_graph=True)
        self.opt.step()
        loss_w, info_w = losses.space_loss(w1, w2, image_space=False)
        loss_c1, info_c1 = losses.space_loss(const2, const3, image_space=False)
        loss_mslv = (loss_w + loss_c1) * 0.01
        self.opt.zero_grad()
        loss_mslv.backward()
        self.opt.step()
        ops.zero_arena_end()
        self.last = dict(w1=w1.detach(), imgs2=imgs2.detach(), w2=w2.detach(), const2=const2.detach(), const3=const3.detach(),
                         loss_msiv=loss_msiv.detach(), info_img=info_img, loss_w=loss_w.detach(), loss_c1=loss_c1.detach())
        return self.last


def invert(st, imgs1, iterations=1500, launch="graph"):
    """The inversion loop on one image group (embedding_img.py:74-170: `iterations` two-phase iterations from the encoder
    checkpoint).  launch="graph" (default: the loop runs at batch 1, where the eager iteration is bound by the host's launch rate -
    15-17 ms against 13 ms of GPU work at 1024^2) captures the iteration once and replays it; "eager" runs EmbedStep.step."""
    st.begin_image()
    done = 0
    if launch == "graph":
        if getattr(st, "_graph", None) is None:
            warm = 1
            st.capture(imgs1, warmup=warm)          # the warm-up iteration is a real iteration of the loop
            done = warm
        else:
            st.set_image(imgs1)                     # later image groups: into the static input of the captured iteration
        run = st.replay
    else:
        run = lambda: st.step(imgs1)
    r = st.last
    for _ in range(max(0, iterations - done)):
        r = run()
    return r


def build_models(img_size=1024, start_features=16, compute_dtype="bf16", device="cuda", seed=0):
    """Models of BASELINE config 5 (StyleGAN1 FFHQ-1024 synthesis + E_Blur) with seeded random-init weights."""
    from .stylegan1 import Generator
    from .encoder_variants import BlurBE
    from .lpips import LPIPS
    torch.manual_seed(seed)
    L = int(math.log2(img_size) - 1)
    Gs = Generator(startf=start_features, maxf=512, layer_count=L, latent_size=512, channels=3, compute_dtype=compute_dtype).to(device)
    for p in Gs.parameters():
        p.requires_grad_(False)
    with torch.no_grad():
        for name, p in Gs.named_parameters():
            if "noise_weight" in name:
                p.fill_(0.05)
    E = BlurBE(startf=start_features, maxf=512, layer_count=L, compute_dtype=compute_dtype).to(device)
    LP = LPIPS(compute_dtype=compute_dtype).to(device)
    return Gs, E, LP
