"""ControlLDM — the stage-2 network object the samplers call (reference cldm.py:20-210), engine-backed.

Keeps the reference surface: ``forward(x_noisy, t, cond)``, ``prepare_condition``,
``vae_encode``, ``vae_decode``, ``cast_dtype``, ``load_pretrained_sd``, ``load_controlnet_from_ckpt``,
``control_scales``, ``eval()``, ``to()``; samplers may re-bind ``model.forward`` (tiling).
"""
import contextlib
import gc
import os
import warnings
from collections import OrderedDict
from typing import Dict, List, Set, Tuple

import torch

from .. import plan
from .clip import FrozenOpenCLIPEmbedder
from .unet import ControlledUnetModel, ControlNet
from .vae import AutoencoderKL

T = torch.Tensor


# skip-connection injections (zero conv + skip add, 12 GEMMs) issued on the ControlNet's stream beside the UNet's middle
# block instead of inside the decoder (DBIR_INJECT_SIDE=0: A/B)
INJECT_ON_SIDE_STREAM = os.environ.get("DBIR_INJECT_SIDE", "1") != "0"
# Deliberate skew between the two encoders (round 5): the UNet encoder (main stream) starts when the ControlNet (side stream)
# has issued its encoder block DBIR_ENC_SKEW (counted after the shared CFG prefix; -1 = no wait, the streams start together
# up to the host's issue order).  Why it can matter: the two encoders run the SAME shapes; started together, both are at the
# 64x64 level (chip-filling, power-limited kernels) and later both at the 16x16 / 8x8 levels (64 - 320 tiles, half-empty
# chip) at the same time; skewed, one stream's small kernels run beside the other's large ones.  A/B: profiles/r5_enc_skew_ab.txt
ENC_SKEW = int(os.environ.get("DBIR_ENC_SKEW", "-1"))
# (HIP stream priorities were tried and removed: no effect under eager launches — the two streams hardly ever co-run,
# profiles/r5_eager_vs_plan_kernel_stats.txt — and -18 % on a replay's side stream: profiles/r5_stream_priority_ab.txt)


class ControlLDM:
    def __init__(self, unet_cfg, vae_cfg, clip_cfg, controlnet_cfg, latent_scale_factor):
        self.unet = ControlledUnetModel(**unet_cfg)
        self.vae = AutoencoderKL(**vae_cfg)
        self.clip = FrozenOpenCLIPEmbedder(**clip_cfg)
        self.controlnet = ControlNet(**controlnet_cfg)
        self.scale_factor = latent_scale_factor
        self.control_scales = [1.0] * 13
        self._mods = [self.unet, self.vae, self.clip, self.controlnet]
        # engine option: run the ControlNet on a second HIP stream while the UNet encoder (which does not depend on it,
        # controlnet.py:30-38) runs on the current one; the streams join before the middle-block control is added.
        # The two networks have the same shapes, and the 16x16 / 8x8 latent levels alone cannot fill 256 CUs.
        # ControlNet on a side stream next to the UNet encoder (DBIR_OVERLAP_STREAMS=0: one stream, A/B)
        self.overlap_streams = os.environ.get("DBIR_OVERLAP_STREAMS", "1") != "0"
        self._side_stream = {}
        # engine option: capture one network evaluation (ControlNet + UNet, ~700 kernel launches from Python through
        # ctypes, ~20 ms of host time) into a HIP graph per (shape, text context, control scales) and replay it every
        # sampling step.  It pays exactly when the host is the bottleneck: at the benchmark's batch 8 (16 samples per
        # evaluation, 26 ms of GPU work) eager launches keep the GPU busy 98.2 % of the sampling loop and replay is not
        # faster (5.69 vs 5.84 img/s, profiles/r2_idle_gaps_*.json; call 20: 5.99 eager / 5.82 graph), nor at batch 4
        # (11.5 eager / 10.6 graph); at batch 1 (2 samples per evaluation) the host's launch time exceeds the GPU work and
        # replay wins (2.06 -> 2.28 img/s).  use_graph: True / False, or None = decide per call: graphs when the
        # evaluation holds at most `graph_auto_rows` latent pixels (samples x h x w) = one 512x512 image under CFG.
        # DBIR_GRAPH=0 / 1 / auto (default).
        # Round 5: (i) a replay used to be REBUILT for every pipeline pass (its key held the prompt tensor's address: two extra
        # evaluations per 50 steps = the "-2 ... -3 % at batch 8" of every earlier graph A/B); with the text context in
        # persistent buffer sets (model/unet.py context_kv) a replay survives a new prompt tensor, and at batch 8 plan replay
        # measures +0.6 ... +1.2 %, graph replay +0.4 % over eager launches, +14 % at batch 1 (profiles/r5_replay_reuse_ab.txt);
        # (ii) the replayer is the engine's own (`use_plan`, below).  `auto` therefore replays every evaluation of up to
        # `graph_auto_rows` latent pixels (64 samples of a 64x64 latent: the benchmark's 16-sample evaluation and the tiled
        # scheduler's 32-sample chunks included); larger ones launch eagerly.
        g = os.environ.get("DBIR_GRAPH", "auto")
        self.use_graph = None if g == "auto" else g == "1"
        self.graph_auto_rows = 64 * 64 * 64
        # engine option (round 5): the same idea as the HIP graph with the engine's OWN executor — the evaluation is recorded
        # once into a `dbir_plan` (diffbir_amd/plan.py, csrc/plan.hip) and replayed from C with ONE host call per evaluation
        # (`dbir_cldm_forward`, the module-level entry point of SURVEY.md 8b).  DBIR_PLAN=1 (default since the end of round 5):
        # replays go through plans; 0: through HIP graphs (`torch.cuda.CUDAGraph`), as in rounds 2 - 4.
        self.use_plan = os.environ.get("DBIR_PLAN", "1") == "1"
        self._graphs: "OrderedDict[tuple, _EvalGraph]" = OrderedDict()
        self._graph_pool = None
        self.max_graphs = 6
        self.max_graph_rows = 2 * self.graph_auto_rows

    # ---- weights ------------------------------------------------------------------------------
    @torch.no_grad()
    def load_pretrained_sd(self, sd: Dict[str, T]) -> Tuple[Set[str], Set[str]]:
        """reference cldm.py:34-62: pick `model.diffusion_model.* / first_stage_model.* / cond_stage_model.*`."""
        module_map = {"unet": "model.diffusion_model", "vae": "first_stage_model", "clip": "cond_stage_model"}
        used, missing = set(), set()
        for name, module in (("unet", self.unet), ("vae", self.vae), ("clip", self.clip)):
            init = {}
            for key, (_, kind) in module._spec.items():
                if kind == "buf":
                    continue
                tk = f"{module_map[name]}.{key}"
                if tk not in sd:
                    missing.add(tk)
                    continue
                init[key] = sd[tk]
                used.add(tk)
            module.load_state_dict(init, strict=False)
        return set(sd.keys()) - used, missing

    @torch.no_grad()
    def load_controlnet_from_ckpt(self, sd: Dict[str, T]) -> None:
        self.controlnet.load_state_dict(sd, strict=True)

    def eval(self):
        return self

    def to(self, device=None, dtype=None):
        for m in self._mods:
            m.to(device, dtype) if dtype is not None else m.to(device)
        return self

    def cast_dtype(self, dtype: torch.dtype) -> "ControlLDM":
        """reference cldm.py:174-210 casts UNet/ControlNet bodies; here it selects the MFMA compute dtype of the
        whole engine (VAE included: under the reference's autocast its convs run in the same 16-bit type)."""
        for m in (self.unet, self.controlnet, self.vae, self.clip):
            m.set_dtype(dtype)
        self.unet.dtype = self.controlnet.dtype = dtype
        return self

    # ---- VAE / conditioning ---------------------------------------------------------------------
    def vae_encode(self, image: T, sample: bool = True, tiled: bool = False, tile_size: int = -1) -> T:
        """reference cldm.py:92-119; `tiled` = the reference's VAEHook algorithm (model/vae.py encode_moments_tiled)."""
        if sample:
            raise NotImplementedError("posterior sampling is a training-only path; inference uses sample=False")
        return self.vae.encode_mode(image, self.scale_factor, tile_size=tile_size if tiled else 0)

    def vae_decode(self, z: T, tiled: bool = False, tile_size: int = -1) -> T:
        """reference cldm.py:121-141."""
        if tiled:
            return self.vae.decode_tiled(z, tile_size, in_scale=1.0 / self.scale_factor)
        return self.vae.decode(z, in_scale=1.0 / self.scale_factor)

    def prepare_condition(self, cond_img: T, txt: List[str], tiled: bool = False, tile_size: int = -1) -> Dict[str, T]:
        """reference cldm.py:143-158: c_img = mode(encoder(img*2-1)) * scale_factor (the `*2-1` is fused into the
        layout-conversion kernel)."""
        return dict(c_txt=self.clip.encode(txt),
                    c_img=self.vae.encode_mode(cond_img, self.scale_factor, in_scale=2.0, in_shift=-1.0,
                                               tile_size=tile_size if tiled else 0))

    # ---- network evaluation ---------------------------------------------------------------------
    def forward(self, x_noisy: T, t: T, cond: Dict[str, T]) -> T:
        """reference cldm.py:160-172. x f32 [B,4,h,w], t [B] (int or fractional), cond {c_txt, c_img} -> f32."""
        from .. import ops
        want = self.use_graph
        if want is None:
            want = x_noisy.shape[0] * x_noisy.shape[2] * x_noisy.shape[3] <= self.graph_auto_rows
        if want and x_noisy.is_cuda and ops._PROFILE is None and ops._TUNER is None \
                and not torch.cuda.is_current_stream_capturing():
            return self._forward_graphed(x_noisy, t, cond)
        return self._forward_eager(x_noisy, t, cond)

    def _forward_graphed(self, x_noisy: T, t: T, cond: Dict[str, T]) -> T:
        """Replay (or capture) the HIP graph / plan of this evaluation.  Static inputs: x, t, c_img (copied in), the time-embedding
        rows (`_ReplayEmb`) and — through the persistent buffer sets — the cross-attention K / V^T of the text context."""
        c_txt, c_img = cond["c_txt"], cond["c_img"]
        self.unet._ensure_packed()
        self.controlnet._ensure_packed()
        # the text context enters through the networks' persistent K / V^T buffer sets (model/unet.py context_kv): a new prompt
        # tensor of the same shape refreshes a set in place (here, on the current stream, in front of the replay), so the key
        # holds the SET's identity, not the prompt tensor's — a replay survives the next pipeline pass instead of being rebuilt
        kvu, kvc = self.unet.context_kv(c_txt), self.controlnet.context_kv(c_txt)
        key = (tuple(x_noisy.shape), str(x_noisy.device), cond.get("cfg_pair"), id(kvu), id(kvc), tuple(c_txt.shape),
               tuple(float(s) for s in self.control_scales), str(self.unet._dtype), bool(self.overlap_streams),
               self.unet._gen, self.controlnet._gen)
        # (split-K workspaces are keyed by the stream current at call time and a recorded plan embeds their raw
        #  pointers — a replay under ANOTHER current stream must not share a replay recorded under this one)
        key = key + (bool(self.use_plan), torch.cuda.current_stream(x_noisy.device).cuda_stream if x_noisy.is_cuda else 0)
        g = self._graphs.get(key)
        if g is None:
            try:
                g = (_EvalPlan if self.use_plan else _EvalGraph)(self, x_noisy, t, c_txt, c_img, cond.get("cfg_pair"))
            except Exception as e:  # capture not possible here: keep launching the same kernels eagerly
                warnings.warn(f"diffbir_amd: HIP graph capture of the network evaluation failed ({e!r}); "
                              "continuing with eager launches")
                self.use_graph = False
                self.reset_graphs()   #  release the pools the kept replays pin: the eager fallback needs the memory
                return self._forward_eager(x_noisy, t, cond)
            g.rows = int(x_noisy.shape[0] * x_noisy.shape[-2] * x_noisy.shape[-1])
            self._graphs[key] = g
            # bounded by count AND by size (every _EvalPlan pins the activations of one evaluation in its own pool, and
            # `graph_auto_rows` admits evaluations of up to 64 samples): the newest replay always stays, older ones go while the
            # kept latent pixels exceed two of the largest admissible evaluations
            while len(self._graphs) > self.max_graphs or \
                    (len(self._graphs) > 1 and sum(getattr(v, "rows", 0) for v in self._graphs.values()) > self.max_graph_rows):
                self._graphs.popitem(last=False)
        else:
            self._graphs.move_to_end(key)
        return g.run(x_noisy, t, c_img, cond.get("t_host"))

    def reset_graphs(self):
        """Drop every captured evaluation.  The shared private pool dies with its last graph, so the handle goes too
        (capturing into a released pool trips an allocator assertion)."""
        self._graphs.clear()
        self._graph_pool = None

    def _forward_eager(self, x_noisy: T, t: T, cond: Dict[str, T]) -> T:
        """cond may carry the engine extension `cfg_pair` = (G, bs): the samplers set it on the [uncond || cond] batch
        they build for classifier-free guidance (identical x / t / c_img in both halves; model/unet.py docstring)."""
        c_txt, c_img = cond["c_txt"], cond["c_img"]
        pair = cond.get("cfg_pair")
        th = cond.get("t_host")  # engine extension: all elements of t equal this host scalar (time-embedding cache)
        eu, ec = cond.get("_emb", (None, None))   # replayed evaluations: the time-embedding rows in static buffers
        if th is not None and os.environ.get("DBIR_CHECK_CFG_PAIR"):
            assert bool((t.float() == float(th)).all()), "t_host set on a batch with other timesteps"
        if pair is not None and os.environ.get("DBIR_CHECK_CFG_PAIR"):
            G, bs = pair
            for v in (x_noisy, t, c_img):
                w = v.reshape(G, 2, bs, *v.shape[1:])
                assert torch.equal(w[:, 0], w[:, 1]), "cfg_pair set on a batch whose halves differ"
        # the zero convs of the ControlNet run inside the UNet, fused with the skip additions (model/unet.py control_feats;
        # DBIR_FUSE_CONTROL=0 = A/B: 13 control tensors + 13 separate additions, the reference's op sequence)
        cn = self.controlnet
        if os.environ.get("DBIR_FUSE_CONTROL", "1") == "0":
            return self._forward_eager_unfused(x_noisy, t, c_txt, c_img, pair)
        if not (self.overlap_streams and x_noisy.is_cuda):
            feats = cn.features(x_noisy, c_img, t, c_txt, pair=pair, t_host=th, emb_all=ec)
            return self.unet(x_noisy, t, c_txt, None, only_mid_control=False, pair=pair, t_host=th, emb_all=eu,
                             control_feats=(feats, cn.zero, self.control_scales))
        main = torch.cuda.current_stream()
        side = self._side_stream.get(x_noisy.device)
        if side is None:
            side = self._side_stream[x_noisy.device] = torch.cuda.Stream(device=x_noisy.device)
        plan.wait_stream(side, main)                # inputs produced on the main stream are ready
        cn._skew_at = ENC_SKEW
        with torch.cuda.stream(side):
            feats = cn.features(x_noisy, c_img, t, c_txt, pair=pair, t_host=th, emb_all=ec)
            done = plan.record_event(side)
        if getattr(cn, "_skew_event", None) is not None:   # the UNet encoder starts once the ControlNet has reached block ENC_SKEW
            plan.wait_event(main, cn._skew_event)
        if not plan.recording():                    # (a recording's allocations live in the plan's private pool: no hand-over)
            for c in feats:                         # allocated on `side`, consumed (and later freed) on `main`
                c.record_stream(main)
        return self.unet(x_noisy, t, c_txt, None, only_mid_control=False, control_ready=done, pair=pair, t_host=th, emb_all=eu,
                         control_feats=(feats, cn.zero, self.control_scales),
                         control_stream=side if INJECT_ON_SIDE_STREAM else None)

    def _forward_eager_unfused(self, x_noisy: T, t: T, c_txt: T, c_img: T, pair) -> T:
        if not (self.overlap_streams and x_noisy.is_cuda):
            control = self.controlnet(x_noisy, c_img, t, c_txt, scales=self.control_scales, pair=pair)
            return self.unet(x_noisy, t, c_txt, control, only_mid_control=False, pair=pair)
        main = torch.cuda.current_stream()
        side = self._side_stream.get(x_noisy.device)
        if side is None:
            side = self._side_stream[x_noisy.device] = torch.cuda.Stream(device=x_noisy.device)
        plan.wait_stream(side, main)
        with torch.cuda.stream(side):
            control = self.controlnet(x_noisy, c_img, t, c_txt, scales=self.control_scales, pair=pair)
            done = plan.record_event(side)
        for c in control:
            c.record_stream(main)
        return self.unet(x_noisy, t, c_txt, control, only_mid_control=False, control_ready=done, pair=pair)

    def __call__(self, *a, **k):
        return self.forward(*a, **k)


@contextlib.contextmanager
def _no_gc():
    """Cyclic garbage collection off inside the block (see _EvalPlan.__init__)."""
    was = gc.isenabled()
    gc.disable()
    try:
        yield
    finally:
        if was:
            gc.enable()


class _ReplayEmb:
    """Time-embedding rows of a replayed evaluation (both networks) in static buffers: a replay cannot consult the per-timestep
    cache itself, and recomputing the rows inside it puts four small dependent GEMMs at the head of each stream (-4 % at batch
    1, -0.4 % at batch 8: profiles/r5_temb_cache_ab.txt).  Filled before every replay from the networks' caches when the sampler
    names the timestep (`t_host`), else computed eagerly from t."""

    def __init__(self, cldm: "ControlLDM", t: T):
        self.cldm = cldm
        self.eu = cldm.unet._time_emb(t).clone()
        self.ec = cldm.controlnet._time_emb(t).clone()

    def fill(self, t: T, t_host) -> None:
        self.eu.copy_(self.cldm.unet._time_emb(t, t_host))
        self.ec.copy_(self.cldm.controlnet._time_emb(t, t_host))


class _EvalGraph:
    """One captured network evaluation.  All activations live in the graph's private memory pool (shared by every
    graph of one ControlLDM: they are replayed one at a time on one stream)."""

    def __init__(self, cldm: ControlLDM, x: T, t: T, c_txt: T, c_img: T, pair=None):
        dev = x.device
        self.x = x.detach().float().contiguous().clone()
        self.t = t.detach().to(torch.float32).contiguous().clone()
        self.c_img = c_img.detach().float().contiguous().clone()
        self.c_txt = c_txt  # kept alive: the context K/V cache of the networks is keyed on its storage
        cldm.unet._ensure_packed()
        cldm.controlnet._ensure_packed()
        self.emb = _ReplayEmb(cldm, self.t)
        cond = dict(c_txt=c_txt, c_img=self.c_img, cfg_pair=pair, _emb=(self.emb.eu, self.emb.ec))
        # warm-up outside the capture: packs weights, fills the context K/V cache, sizes split-K workspaces
        cldm._forward_eager(self.x, self.t, cond)
        torch.cuda.synchronize(dev)
        # the captured kernels read the cached cross-attention K / V^T of this context through raw pointers: own them
        # (the networks' caches are bounded and may drop the entry while this graph is still replayed)
        self.ctx_kv = (cldm.unet.context_kv(c_txt), cldm.controlnet.context_kv(c_txt))
        if cldm._graph_pool is None:
            cldm._graph_pool = torch.cuda.graph_pool_handle()
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph, pool=cldm._graph_pool):
            self.out = cldm._forward_eager(self.x, self.t, cond)
        torch.cuda.synchronize(dev)

    def run(self, x: T, t: T, c_img: T, t_host=None) -> T:
        self.x.copy_(x)
        self.t.copy_(t)
        self.emb.fill(self.t, t_host)
        if c_img.data_ptr() != self.c_img.data_ptr():
            self.c_img.copy_(c_img)
        self.graph.replay()
        return self.out.clone()  # the static output buffer is overwritten by the next replay


class _EvalPlan:
    """One network evaluation recorded into a native `dbir_plan` (csrc/plan.hip) and replayed through `dbir_cldm_forward`:
    same launches on the same operands in the same per-stream order as the eager pass, issued from C.  The activations live
    in a private PyTorch memory pool owned by this object (diffbir_amd/plan.py explains why their reuse pattern is safe to
    replay).  Interface of `_EvalGraph`."""

    def __init__(self, cldm: ControlLDM, x: T, t: T, c_txt: T, c_img: T, pair=None):
        dev = x.device
        self.c_txt = c_txt
        # A cyclic-GC pass that destroys an OLDER plan's MemPool while this thread allocates into a pool aborts the process
        # inside PyTorch's caching allocator (round 5, full GPU suite: "Fatal Python error: Aborted" in a GC pass during the
        # recording below; tools/probes/plan_lifetime_probe.py destroys 24 pools between recordings without trouble).  So:
        # collect garbage NOW, and keep the collector off while a pool context is open.
        gc.collect()
        self.pool = torch.cuda.MemPool()
        with _no_gc(), torch.cuda.use_mem_pool(self.pool, device=dev):
            self.x = x.detach().float().contiguous().clone()
            self.t = t.detach().to(torch.float32).contiguous().clone()
            self.c_img = c_img.detach().float().contiguous().clone()
        cldm.unet._ensure_packed()
        cldm.controlnet._ensure_packed()
        self.emb = _ReplayEmb(cldm, self.t)
        cond = dict(c_txt=c_txt, c_img=self.c_img, cfg_pair=pair, _emb=(self.emb.eu, self.emb.ec))
        # warm-up outside the recording: packs weights, fills the context K/V cache, sizes split-K workspaces, lets the
        # first-use autotuner settle every problem key (it re-runs launches: never inside a recording)
        cldm._forward_eager(self.x, self.t, cond)
        torch.cuda.synchronize(dev)
        self.ctx_kv = (cldm.unet.context_kv(c_txt), cldm.controlnet.context_kv(c_txt))
        self._keep = []
        gc.collect()
        with _no_gc(), torch.cuda.use_mem_pool(self.pool, device=dev):
            with plan.Recorder(torch.cuda.current_stream(dev)) as rec:
                self.out = cldm._forward_eager(self.x, self.t, cond)
        torch.cuda.synchronize(dev)
        self.plan = rec.build()
        self.eps = torch.empty_like(self.out)
        for slot, buf in enumerate((self.x, self.t, self.c_img, self.out)):
            self.plan.bind(slot, buf)
        self.calls, self.n_streams, self.n_events = self.plan.calls, self.plan.n_streams, self.plan.n_events
        # Self-check before the plan is trusted: a replay on OTHER inputs must equal the eager evaluation of those inputs bit
        # for bit.  A device operation that is not a C-ABI call (a stray torch copy inside the evaluation) is invisible to the
        # recorder — the replay would silently reuse that buffer's contents from the recording pass (this is how the tiled
        # scheduler's strided gather was found: 5.7 dB).  Two evaluations, once per plan.
        xv, cv = (self.x * 0.5 + 0.25).contiguous(), (self.c_img * 0.5 - 0.125).contiguous()
        want = cldm._forward_eager(xv, self.t, dict(cond, c_img=cv)).clone()
        got = self.run(xv, self.t, cv)
        torch.cuda.synchronize(dev)
        if not torch.equal(got, want):
            raise RuntimeError(f"dbir_plan self-check failed: replay differs from the eager evaluation (max abs "
                               f"{(got - want).abs().max().item():.3e}) — some device operation of the evaluation is not recorded")

    def run(self, x: T, t: T, c_img: T, t_host=None) -> T:
        xs = x.detach().float().contiguous()
        ts = t.detach().to(torch.float32).contiguous()
        cs = c_img.detach().float().contiguous()
        self.emb.fill(ts, t_host)
        return self.plan.cldm_forward(xs, ts, cs, torch.empty_like(self.out))
