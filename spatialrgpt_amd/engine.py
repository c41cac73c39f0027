"""SrgptEngine: the region-grounded forward/generate path on one MI355X.

Stage map (reference file:line -> method):
  A1  VisionTower.forward            multimodal_encoder/vision_encoder.py:115-132   -> vit()
  A2  feature_refinement             region_extractor/base_extractor.py:137-147     -> feature_refinement()
  A3/4 MaskPooling + connectors      region_extractor/base_extractor.py:32-84,149-173 -> region_extractor()
  A5  mlp_downsample projector       multimodal_projector/base_projector.py:32-94   -> mm_projector()
  A6  token-stream splice            llava_arch.py:333-650                           -> prepare_inputs()
  A7-13 Llama prefill + greedy loop  modeling_llama.py + HF GenerationMixin          -> prefill()/generate_ids()
Every arithmetic step is a HIP kernel (spatialrgpt_amd/ops.py -> libsrgpt_hip.so); torch allocates buffers.
"""
from __future__ import annotations

import ctypes as C
import warnings
from typing import Dict, List, Optional, Sequence

import torch

from . import _lib as L
from . import ops
from .config import SrgptConfig
from .constants import IGNORE_INDEX, IMAGE_TOKEN_INDEX
from .weights import PreparedWeights


class DecodeState:
    """Static KV cache + device-side decode bookkeeping for a (batch, max_pos) geometry."""

    def __init__(self, eng: "SrgptEngine", batch: int, max_pos: int, ws_tokens: int, max_new: int):
        cfg, dev, dt = eng.cfg, eng.device, eng.dtype
        self.batch, self.max_pos, self.ws_tokens, self.max_new = batch, max_pos, ws_tokens, max_new
        shape = (cfg.layers, batch, cfg.kv_heads, max_pos, cfg.head_dim)
        self.kcache = torch.empty(shape, device=dev, dtype=dt)
        self.vcache = torch.empty(shape, device=dev, dtype=dt)
        self.pos = torch.zeros((batch,), device=dev, dtype=torch.int32)
        self.tok = torch.zeros((batch,), device=dev, dtype=torch.int64)
        self.out_ids = torch.zeros((batch, max_new), device=dev, dtype=torch.int64)
        self.step = torch.zeros((1,), device=dev, dtype=torch.int32)
        self.logits = torch.empty((batch, eng.w.vocab), device=dev, dtype=torch.float32)
        lib = L.load()
        nbytes = lib.srgpt_llm_ws_bytes(C.byref(eng.w.llm), batch, ws_tokens)
        if nbytes < 0:
            raise RuntimeError("srgpt_llm_ws_bytes failed")
        self.ws = torch.zeros((nbytes,), device=dev, dtype=torch.uint8)  # zeroed once: holds the decode-attention arrival tickets
        st = L.LlmState()
        st.batch, st.max_pos, st.max_new, st.ws_tokens = batch, max_pos, max_new, ws_tokens
        st.kcache, st.vcache = self.kcache.data_ptr(), self.vcache.data_ptr()
        st.pos, st.tok, st.out_ids, st.step = self.pos.data_ptr(), self.tok.data_ptr(), self.out_ids.data_ptr(), self.step.data_ptr()
        st.ws, st.logits = self.ws.data_ptr(), self.logits.data_ptr()
        st.sampling = None
        self.c = st
        self.host_len = [0] * batch  # cached positions per row as known on the host (prefill lengths + steps taken)
        self.graphs = {}             # "greedy" / "sample" -> captured decode step (the pick kernels differ)
        self.sampling: Optional[ops.SamplingParams] = None  # device parameter block of the draw; ONE address for the state's lifetime
        self._eng = eng

    def seq_len(self) -> int:
        """cached positions (longest row) -- what `past_key_values[-1][-1].shape[-2]` is for the reference (llava_arch.py:364)."""
        return max(self.host_len)

    def set_sampling(self, sampling: Optional[dict]):
        """None: the step picks argmax.  dict(temperature, top_k, top_p, seed): the step DRAWS (sample.hip); the parameters go to
        the state's device block (one small H2D copy), so the graph captured for sampling serves every setting."""
        if sampling is None:
            self.c.sampling = None
            return
        if self.sampling is None:
            self.sampling = ops.SamplingParams(self._eng.device, self.batch)
        self.sampling.set(sampling["temperature"], sampling.get("top_k"), sampling.get("top_p"), sampling.get("seed", 0))
        self.c.sampling = self.sampling.ptr()

    def ensure_graph(self):
        key = "greedy" if not self.c.sampling else "sample"
        if key not in self.graphs:
            g = L.vp()
            L.check(L.load().srgpt_llm_decode_graph_create(C.byref(self._eng.w.llm), C.byref(self.c), ops._stream(), C.byref(g)))
            self.graphs[key] = g
        return self.graphs[key]

    def __del__(self):
        try:
            for g in self.graphs.values():
                L.load().srgpt_graph_destroy(g)
        except Exception:
            pass


class SplicePlan:
    """what srgpt_splice_plan left on the device (row -> source table, per-prompt facts) + the host's copy of the facts"""

    def __init__(self, **kw):
        self.__dict__.update(kw)


class SrgptEngine:
    def __init__(self, cfg: SrgptConfig, state_dict: Dict[str, torch.Tensor], device="cuda", dtype=torch.bfloat16,
                 rope_positions: int = 0, consume_state_dict: bool = False, llm_weight_format: str = "native", parts=None,
                 decode_layout: str = "packed"):
        L.load()  # fail loudly if the HIP extension is missing
        self.cfg, self.device, self.dtype = cfg, torch.device(device), dtype
        if self.device.type != "cuda":
            raise RuntimeError("SrgptEngine runs on an MI355X (torch device 'cuda'); there is no CPU path")
        if cfg.mm_projector_type != "mlp_downsample":
            raise ValueError(f"Unknown projector type: {cfg.mm_projector_type}")
        if cfg.enable_region and cfg.region_extractor_type != "regiongpt":
            raise NotImplementedError(f"{cfg.region_extractor_type} not implemented")  # base_extractor.py:160-161
        if cfg.select_feature not in ("cls_patch", "patch"):
            raise ValueError(f"Unexpected select feature: {cfg.select_feature}")
        # decode_layout "packed": fp8 LLM weights also get MFMA-operand-order copies for the batched decode step (weights.py)
        self.w = PreparedWeights(cfg, state_dict, self.device, dtype, rope_positions, consume=consume_state_dict,
                                 llm_weight_format=llm_weight_format, parts=parts, decode_layout=decode_layout)
        self._state: Optional[DecodeState] = None
        self._vit_ws: Optional[torch.Tensor] = None
        self.use_graph = True
        # the decode loop runs on its own stream: hipGraph capture is illegal on the legacy default stream
        self.stream = torch.cuda.Stream(device=self.device)

    # ------------------------------------------------------------------ A1
    def vit(self, images: torch.Tensor, out_dtype: Optional[torch.dtype] = None) -> torch.Tensor:
        """[n,3,S,S] (any float dtype: cast to the engine dtype at the boundary, vision_encoder.py:127) ->
        hidden_states[select_layer] [n, grid^2, C] in the ENGINE dtype; `out_dtype` reproduces the module-level cast back
        to the caller's image dtype (vision_encoder.py:130) for callers that use the tower on its own."""
        self._need("vit")
        x = images.to(device=self.device, dtype=self.dtype).contiguous()  # vision_encoder.py:127
        n, ch, S, S2 = x.shape
        if ch != 3 or S != self.cfg.image_size or S2 != S:
            raise ValueError(f"vision tower expects [n,3,{self.cfg.image_size},{self.cfg.image_size}] images, got {tuple(x.shape)}")
        lib = L.load()
        need = lib.srgpt_vit_ws_bytes(C.byref(self.w.vit), n)
        if self._vit_ws is None or self._vit_ws.numel() < need:
            self._vit_ws = torch.empty((need,), device=self.device, dtype=torch.uint8)
        out = torch.empty((n, self.cfg.tower_tokens, self.cfg.vit_hidden), device=self.device, dtype=self.dtype)
        L.check(lib.srgpt_vit_forward(C.byref(self.w.vit), x.data_ptr(), out.data_ptr(), self._vit_ws.data_ptr(), n, ops._stream()))
        if self.cfg.select_feature == "patch":
            out = out[:, 1:]  # feature_select drops token 0 whatever the tower (vision_encoder.py:28-29)
        return out if out_dtype is None or out_dtype == out.dtype else out.to(out_dtype)  # vision_encoder.py:130

    # ------------------------------------------------------------------ A2
    def _need(self, part: str):
        if part not in self.w.parts:
            raise RuntimeError(f"this engine was built without its {part!r} component (parts = {self.w.parts})")

    def feature_refinement(self, tower: torch.Tensor):
        self._need("region")
        tower = tower.to(device=self.device, dtype=self.dtype)
        n, HW, Cc = tower.shape
        g = int(HW ** 0.5)
        if g * g != HW:
            raise ValueError(f"feature_refinement needs a square token grid, got {HW} tokens")  # einops error in the reference
        w = self.w
        x = tower.reshape(n * HW, Cc)
        # ConvT(2,2) -> LayerNorm2d -> GELU -> ConvT(2,2) -> GELU, all channels-last
        y = ops.gemm(x, w.dc1_w, w.dc1_b, bias_mod=Cc, out_mode=L.OUT_DECONV2X, gw=g, out_shape=(n * 4 * HW, Cc))
        y = ops.layernorm(y, w.ln2d_w, w.ln2d_b, 1e-6, act=L.ACT_GELU_ERF)
        hres = ops.gemm(y, w.dc2_w, w.dc2_b, act=L.ACT_GELU_ERF, bias_mod=Cc, out_mode=L.OUT_DECONV2X, gw=2 * g,
                        out_shape=(n * 16 * HW, Cc))
        lres = ops.avgpool(hres, n, 4 * g, 27)  # AdaptiveAvgPool2d(27) is hard-coded (base_extractor.py:123)
        return hres.reshape(n, 16 * HW, Cc), lres

    # ------------------------------------------------------------------ A3 / A4
    def mask_pooling(self, feats: torch.Tensor, masks: Optional[Sequence[Optional[torch.Tensor]]]):
        n = feats.shape[0]
        if masks is None:
            masks = [None] * n
        out = []
        for i in range(n):
            m = masks[i]
            if m is None:
                out.append(None)
            elif m.dtype == torch.uint8:
                # raw uint8 masks [K, H, W]
                if self.cfg.image_aspect_ratio == "pad":
                    # process_regions' pad mode (mm_utils.py:505-531): zero square + the processor's bicubic resize on the device,
                    # then the pooling of float masks (soft edges: the nearest-tap fusion below does not apply)
                    from .mm_utils import masks_pad_resize_device

                    S = self.cfg.image_size
                    out.append(ops.region_pool(feats[i].to(self.dtype), masks_pad_resize_device(m.to(self.device), S, S, self.dtype)))
                    continue
                if self.cfg.image_aspect_ratio != "resize":
                    raise NotImplementedError(f"raw uint8 masks: image_aspect_ratio {self.cfg.image_aspect_ratio!r} "
                                              "(process_regions handles 'resize' and 'pad')")
                # "resize": nearest resize to the processor size + float + resample fused into the pooling kernel
                out.append(ops.region_pool_u8(feats[i].to(self.dtype), m.to(self.device), self.cfg.image_size))
            else:
                out.append(ops.region_pool(feats[i].to(self.dtype), m.to(self.device)))
        return out

    def region_extractor(self, hres, depth_features, masks):
        self._need("region")
        w = self.w

        def connect(pooled, W, b):
            return [None if p is None else ops.gemm(p, W, b) for p in pooled]

        mask_embeds = connect(self.mask_pooling(hres, masks), w.rgb_w, w.rgb_b)
        depth_embeds = None
        if depth_features is not None:
            depth_embeds = connect(self.mask_pooling(depth_features, masks), w.depth_w, w.depth_b)
        return mask_embeds, depth_embeds

    # ------------------------------------------------------------------ A5
    def mm_projector(self, lres: torch.Tensor) -> torch.Tensor:
        self._need("projector")
        w = self.w
        lres = lres.to(device=self.device, dtype=self.dtype)
        n = lres.shape[0]
        x = ops.s2d(lres)
        tokens = x.shape[1]
        x = ops.layernorm(x.reshape(n * tokens, -1), w.mp_ln_w, w.mp_ln_b, 1e-5)
        x = ops.gemm(x, w.mp_w1, w.mp_b1, act=L.ACT_GELU_ERF)
        x = ops.gemm(x, w.mp_w2, w.mp_b2)
        return x.reshape(n, tokens, -1)

    # ------------------------------------------------------------------ llava_arch.py:387-411
    def encode_visual(self, images, depths, masks, stages: Optional[dict] = None):
        cfg = self.cfg
        if isinstance(images, (list, tuple)):
            images = torch.cat(list(images), dim=0)
        elif images.ndim == 5:
            images = images.flatten(0, 1)
        if depths is not None:
            if isinstance(depths, (list, tuple)):
                depths = torch.cat(list(depths), dim=0)
            elif depths.ndim == 5:
                depths = depths.flatten(0, 1)
        n = images.shape[0]
        use_depth = cfg.enable_region and cfg.enable_depth and depths is not None
        # RGB and depth go through the tower as ONE batch of 2n images (same weights)
        # fp16 / fp32 callers (the reference's loader builds fp16: builder.py:62, eval_region_cls.py:316-317) are cast to the
        # engine dtype here and every intermediate stays in it -- the weights are only held in the engine dtype
        images = images.to(device=self.device, dtype=self.dtype)
        both = torch.cat([images, depths.to(device=self.device, dtype=self.dtype)], dim=0) if use_depth else images
        feats = self.vit(both)
        tower = feats[:n]
        mask_embeds = depth_embeds = None
        if cfg.enable_region:
            hres, lres = self.feature_refinement(tower.contiguous())
            depth_features = feats[n:].contiguous() if use_depth else None
            mask_embeds, depth_embeds = self.region_extractor(hres, depth_features, masks)
            if stages is not None:
                stages.update(hres=hres, lres=lres, depth_features=depth_features)
        else:
            lres = tower
        image_features = self.mm_projector(lres)
        if stages is not None:
            stages.update(tower_features=tower, image_features=image_features, mask_embeds=mask_embeds,
                          depth_embeds=depth_embeds)
        return image_features, mask_embeds, depth_embeds

    # ------------------------------------------------------------------ A6
    def _region_counts(self, masks, n_images: int):
        """rows of region embeddings each image will get (None: none) -- known from the mask list before any kernel runs"""
        if not self.cfg.enable_region:
            return None
        if masks is None:
            return [None] * n_images
        return [None if m is None else int(m.shape[0]) for m in masks]

    def splice_plan(self, input_ids: torch.Tensor, attention_mask, n_images: int, region_counts, have_depths: bool,
                    nimg_feat: Optional[int] = None) -> "SplicePlan":
        """Token-stream splice, step 1 (llava_arch.py:420-611): the row -> source table of the whole batch, built ON THE DEVICE from
        the ids (csrc/splice.hip) -- the ids never travel; what comes back is 8 ints per prompt (lengths, counts, id range) for the
        batch length T and the reference's error / warning conditions.  prepare_inputs() calls this BEFORE it launches the vision
        tower, so the one read-back of the request waits for nothing."""
        cfg, dev = self.cfg, self.device
        ids = input_ids.detach().to(device=dev, dtype=torch.int64).contiguous()
        B, P = ids.shape
        am8 = None
        if attention_mask is not None:
            am8 = attention_mask.detach().to(device=dev).ne(0).to(torch.uint8).contiguous()
        nf = cfg.llm_image_tokens if nimg_feat is None else int(nimg_feat)
        use_masks = bool(cfg.enable_region and region_counts is not None)
        use_depths = bool(cfg.enable_region and cfg.enable_depth and have_depths and region_counts is not None)
        info, off = [], 0
        for i in range(n_images):
            c = region_counts[i] if (region_counts is not None and i < len(region_counts)) else None
            info += [-1 if c is None else c, off]
            off += 0 if c is None else c
        info_dev = torch.tensor(info or [0, 0], dtype=torch.int32).to(dev)
        Tcap = max(P + n_images * max(nf - 1, 0), 1)
        lib = L.load()
        desc = torch.empty((B, Tcap, 2), device=dev, dtype=torch.int32)
        stats = torch.empty((B, L.SPLICE_STATS), device=dev, dtype=torch.int32)
        scratch = torch.empty((max(int(lib.srgpt_splice_scratch_ints(B, n_images)), 1),), device=dev, dtype=torch.int32)
        mx = cfg.tokenizer_model_max_length
        L.check(lib.srgpt_splice_plan(ids.data_ptr(), None if am8 is None else am8.data_ptr(), B, P, nf, n_images, info_dev.data_ptr(),
                                      int(use_masks), int(use_depths), int(cfg.mask_token_id), int(cfg.depth_token_id),
                                      int(mx) if mx is not None else 0, Tcap, desc.data_ptr(), stats.data_ptr(), scratch.data_ptr(),
                                      ops._stream()))
        host = stats.cpu().tolist()  # the request's one device -> host copy (B x 8 ints)
        for b, (ln, ln_raw, n_img, n_mask, n_depth, first_img, mn, mxid) in enumerate(host):
            if n_img > 0:
                if first_img + n_img > n_images:
                    raise IndexError(f"index {first_img + n_img - 1} is out of bounds for dimension 0 with size {n_images}: "
                                     "more <image> tokens than images")
                me = region_counts[first_img] if region_counts is not None else None
                if cfg.enable_region and me is None and n_mask > 0:
                    print("Error: mask embed is None, but the num of <mask> is not 0!!!")
                if cfg.enable_region and cfg.enable_depth and have_depths and me is None and n_depth > 0:
                    print("Error: depth embed is None, but the num of <depth> is not 0!!!")
                if use_masks and me is not None and n_mask > me:
                    raise RuntimeError(f"shape mismatch: {n_mask} <mask> tokens but only {me} mask embeddings")
                if use_depths and me is not None and n_depth > me:
                    raise RuntimeError(f"shape mismatch: {n_depth} <depth> tokens but only {me} depth embeddings")
        lo, hi = min(r[6] for r in host), max(r[7] for r in host)  # (0, 0) for a prompt without text rows
        if lo < 0 or hi >= self.w.vocab:
            raise IndexError(f"index out of range in self: token ids must be in [0, {self.w.vocab}), got [{lo}, {hi}]")
        if mx is not None and any(r[1] > mx for r in host):
            warnings.warn("Inputs truncated!")
        lens = [r[0] for r in host]
        return SplicePlan(ids=ids, B=B, P=P, nimg_feat=nf, n_images=n_images, Tcap=Tcap, desc=desc, stats=stats, lens=lens,
                          T=max(max(lens), 1), have_attention_mask=attention_mask is not None,
                          am_dtype=None if attention_mask is None else attention_mask.dtype,
                          region_counts=region_counts, have_depths=have_depths, attention_mask=attention_mask)

    def splice_apply(self, plan: "SplicePlan", image_features, mask_embeds, depth_embeds, labels=None):
        """Token-stream splice, step 2: ONE gather launch writes every row of inputs_embeds [B, T, H] from its source (embedding
        table, image features, region embeddings, zeros for padding) and, when asked, the spliced labels / attention mask."""
        cfg, dev = self.cfg, self.device
        B, T, H = plan.B, plan.T, cfg.hidden
        # the plan's offsets into the concatenated region tables were computed from PREDICTED row counts (masks[i].shape[0]) before
        # the tower ran: what is gathered must have exactly those shapes, or the kernel reads the wrong rows / past a table
        if image_features.shape[0] != plan.n_images or image_features.shape[1] != plan.nimg_feat:
            raise RuntimeError(f"splice: planned for {plan.n_images} x {plan.nimg_feat} image rows, got {tuple(image_features.shape[:2])}")
        counts = plan.region_counts
        for what, embeds, planned in (("mask", mask_embeds, cfg.enable_region and counts is not None),
                                     ("depth", depth_embeds, cfg.enable_region and cfg.enable_depth and plan.have_depths and counts is not None)):
            if not planned:
                continue
            got = [] if embeds is None else [None if e is None else int(e.shape[0]) for e in embeds]
            got += [None] * (plan.n_images - len(got))
            if got != [counts[i] if i < len(counts) else None for i in range(plan.n_images)]:
                raise RuntimeError(f"splice: planned for {what} embeddings of {list(counts)} rows per image, got {got}")
        out = torch.empty((B * T, H), device=dev, dtype=self.dtype)
        feats = image_features.reshape(-1, H).to(dtype=self.dtype).contiguous()
        me = de = None
        if mask_embeds is not None and any(e is not None for e in mask_embeds):
            me = torch.cat([e for e in mask_embeds if e is not None], 0).to(self.dtype).contiguous()
        if depth_embeds is not None and any(e is not None for e in depth_embeds):
            de = torch.cat([e for e in depth_embeds if e is not None], 0).to(self.dtype).contiguous()
        lab = None if labels is None else labels.detach().to(device=dev, dtype=torch.int64).contiguous()
        lab_out = torch.empty((B, T), device=dev, dtype=torch.int64) if labels is not None else None
        am_out = torch.empty((B, T), device=dev, dtype=torch.uint8) if plan.have_attention_mask else None
        L.check(L.load().srgpt_splice_gather(plan.desc.data_ptr(), plan.stats.data_ptr(), B, plan.Tcap, T,
                                             int(cfg.padding_side == "left"), H, ops.dt_code(out), self.w.embed.data_ptr(),
                                             feats.data_ptr(), None if me is None else me.data_ptr(),
                                             None if de is None else de.data_ptr(), None if lab is None else lab.data_ptr(), plan.P,
                                             IGNORE_INDEX, out.data_ptr(), None if lab_out is None else lab_out.data_ptr(),
                                             None if am_out is None else am_out.data_ptr(), ops._stream()))
        am = None if am_out is None else am_out.to(plan.am_dtype)
        if labels is not None:
            return out.reshape(B, T, H), am, plan.lens, lab_out
        return out.reshape(B, T, H), am, plan.lens

    def splice(self, input_ids: torch.Tensor, attention_mask, image_features, mask_embeds, depth_embeds, have_depths,
               labels=None):
        """plan + gather for callers that already hold the visual features.
        Returns (inputs_embeds [B,T,H], attention_mask|None, lengths list); with `labels` ([B,P] int64) a 4th value: the
        spliced labels [B,T] on the device (IGNORE_INDEX over image rows and padding, llava_arch.py:513-533, :558-611)."""
        counts = None
        if self.cfg.enable_region and mask_embeds is not None:
            counts = [None if e is None else int(e.shape[0]) for e in mask_embeds]
        plan = self.splice_plan(input_ids, attention_mask, image_features.shape[0], counts,
                                have_depths and depth_embeds is not None, nimg_feat=image_features.shape[1])
        return self.splice_apply(plan, image_features, mask_embeds, depth_embeds, labels)

    def prepare_inputs(self, input_ids, images, depths=None, masks=None, attention_mask=None, stages=None, labels=None):
        # the splice is PLANNED first: its read-back (lengths, the reference's error conditions) happens while the GPU has nothing
        # queued; then the tower / refinement / pooling / projector launches go out back to back and the gather follows them
        if isinstance(images, (list, tuple)):
            n_images = sum(int(im.shape[0]) for im in images)
        else:
            n_images = int(images.shape[0] * images.shape[1]) if images.ndim == 5 else int(images.shape[0])
        counts = self._region_counts(masks, n_images)
        use_depth = self.cfg.enable_region and self.cfg.enable_depth and depths is not None
        plan = self.splice_plan(input_ids, attention_mask, n_images, counts, use_depth)
        image_features, mask_embeds, depth_embeds = self.encode_visual(images, depths, masks, stages)
        if image_features.shape[1] != plan.nimg_feat or image_features.shape[0] != plan.n_images:  # a geometry the config did not predict
            plan = self.splice_plan(input_ids, attention_mask, image_features.shape[0], counts, use_depth,
                                    nimg_feat=image_features.shape[1])
        res = self.splice_apply(plan, image_features, mask_embeds, depth_embeds, labels)
        if stages is not None:
            stages["inputs_embeds"] = res[0]
        return res

    def _check_ids(self, ids) -> None:
        """nn.Embedding raises IndexError for ids outside [0, vocab); the row-gather kernel would read out of bounds
        (e.g. the -200 sentinel on an images=None path, or <mask>/<depth> ids past a table that was never resized)."""
        if isinstance(ids, torch.Tensor):
            if ids.numel() == 0:
                return
            lo, hi = int(ids.min()), int(ids.max())
        else:
            if not ids:
                return
            lo, hi = min(ids), max(ids)
        if lo < 0 or hi >= self.w.vocab:
            raise IndexError(f"index out of range in self: token ids must be in [0, {self.w.vocab}), got [{lo}, {hi}]")

    def embed_tokens(self, input_ids: torch.Tensor) -> torch.Tensor:
        B, P = input_ids.shape
        self._check_ids(input_ids)
        return ops.embed_rows(self.w.embed, input_ids.to(self.device)).reshape(B, P, -1)

    # ------------------------------------------------------------------ A7-A13
    def _get_state(self, batch: int, T: int, max_new: int, fresh: bool = False) -> DecodeState:
        """Cache/workspace for `batch` sequences of T prompt positions + max_new generated ones.  The pooled state serves the
        internal generate() path; `fresh=True` allocates an independent one (a handle returned to the caller as
        `past_key_values` must not be overwritten by the next call -- the reference returns independent caches)."""
        self._need("llm")
        need_pos = T + max_new
        if need_pos > self.w.rope_len:
            raise ValueError(f"sequence of {need_pos} positions exceeds the RoPE table ({self.w.rope_len})")
        # positions are capped by the RoPE table: a cache longer than it would let decode steps index past its end
        max_pos = min((need_pos + 127) // 128 * 128, self.w.rope_len)
        if fresh:
            return DecodeState(self, batch, max_pos, max(T, 1), max_new)
        s = self._state
        if s is None or s.batch != batch or s.max_pos < need_pos or s.ws_tokens < T or s.max_new < max_new:
            self._state = None
            s = DecodeState(self, batch, max_pos, max(T, 1), max_new)
            self._state = s
        return s

    def prefill(self, inputs_embeds: torch.Tensor, max_new: int = 1, all_logits: bool = False, hidden_states: bool = False,
                lens: Optional[torch.Tensor] = None, fresh_state: bool = False):
        """inputs_embeds [B,T,H]; `lens` (int [B]) marks a RIGHT-padded ragged batch: row b has lens[b] valid positions,
        its logits come from position lens[b]-1 and decoding continues there.  Returns (state, logits_all|None, hiddens|None)."""
        B, T, H = inputs_embeds.shape
        x = inputs_embeds.to(device=self.device, dtype=self.dtype).contiguous()
        st = self._get_state(B, T, max_new, fresh=fresh_state)
        al = torch.empty((B, T, self.w.vocab), device=self.device, dtype=torch.float32) if all_logits else None
        hs = torch.empty((self.cfg.layers + 1, B, T, H), device=self.device, dtype=self.dtype) if hidden_states else None
        if lens is None:
            L.check(L.load().srgpt_llm_prefill(C.byref(self.w.llm), C.byref(st.c), x.data_ptr(), T,
                                               None if al is None else al.data_ptr(), None if hs is None else hs.data_ptr(),
                                               ops._stream()))
        else:
            ld = lens.to(device=self.device, dtype=torch.int32).contiguous()
            if ld.shape != (B,):
                raise ValueError(f"prefill: lens must have shape ({B},)")
            L.check(L.load().srgpt_llm_prefill_ragged(C.byref(self.w.llm), C.byref(st.c), x.data_ptr(), T, ld.data_ptr(),
                                                      None if al is None else al.data_ptr(),
                                                      None if hs is None else hs.data_ptr(), ops._stream()))
        st.host_len = [T] * B if lens is None else [int(v) for v in lens.tolist()]
        return st, al, hs

    def step(self, st: DecodeState, input_ids: torch.Tensor) -> torch.Tensor:
        """One incremental forward over a cached state (HF `forward(input_ids[B,1], past_key_values=...)`): appends the
        tokens at every row's next position and returns the fp32 logits [B, vocab]."""
        if input_ids.shape != (st.batch, 1):
            raise ValueError(f"step: input_ids must be [{st.batch}, 1]")
        if max(st.host_len) >= min(st.max_pos, self.w.rope_len):
            raise ValueError(f"KV cache full ({min(st.max_pos, self.w.rope_len)} positions): re-run forward() with a larger "
                             "cache_reserve / a longer RoPE table")
        self._check_ids(input_ids)
        st.tok.copy_(input_ids[:, 0].to(device=self.device, dtype=torch.int64))
        L.check(L.load().srgpt_llm_decode_step(C.byref(self.w.llm), C.byref(st.c), ops._stream()))
        st.host_len = [n + 1 for n in st.host_len]
        return st.logits.clone()

    def greedy_decode(self, st: DecodeState, max_new_tokens: int, eos_token_id=None, pad_token_id=None,
                      stopping_criteria=None, sampling: Optional[dict] = None) -> torch.Tensor:
        """HF generation-loop semantics (new ids only, finished rows padded), device-side steps via hipGraph.
        sampling = None: greedy.  sampling = dict(temperature, top_k, top_p, seed): every step DRAWS its token on the device
        (temperature -> top-k -> top-p -> categorical, sample.hip) -- same loop, same graph mechanism, no per-token host work."""
        cur = torch.cuda.current_stream(self.device)
        self.stream.wait_stream(cur)
        with torch.cuda.stream(self.stream):
            st.set_sampling(sampling)
            try:
                n_keep, eos = self._decode_loop(st, max_new_tokens, eos_token_id, stopping_criteria)
            finally:
                st.c.sampling = None
        cur.wait_stream(self.stream)
        # the decode attention hands partials between workgroups inside a launch (arrival tickets): a ticket left non-zero means a
        # launch merged nothing and later steps used stale attention output -- fail loudly, never return such ids
        L.check(L.load().srgpt_llm_decode_sync_state(C.byref(self.w.llm), C.byref(st.c), ops._stream()))
        out = st.out_ids[:, :n_keep].clone()
        if eos:
            # rows that finished early are padded with pad_token_id (HF behaviour)
            pad = pad_token_id if pad_token_id is not None else eos[0]  # HF: pad defaults to the FIRST eos id
            is_eos = (out[:, :, None] == torch.tensor(eos, device=out.device, dtype=out.dtype)[None, None, :]).any(-1)
            after = (is_eos.int().cumsum(dim=1) - is_eos.int()) > 0  # strictly after a row's first EOS
            out = torch.where(after, torch.full_like(out, pad), out)
        return out

    def _decode_loop(self, st: DecodeState, max_new_tokens: int, eos_token_id, stopping_criteria):
        lib = L.load()
        stream = ops._stream()
        L.check(lib.srgpt_llm_sample_first(C.byref(self.w.llm), C.byref(st.c), stream))
        eos = None  # ordered list of EOS ids (HF accepts an int or a list; Llama-3 checkpoints list two)
        if eos_token_id is not None:
            eos = [int(e) for e in eos_token_id] if isinstance(eos_token_id, (list, tuple, set)) else [int(eos_token_id)]
            eos = eos or None
        interactive = stopping_criteria is not None and len(stopping_criteria) > 0
        graph = st.ensure_graph() if self.use_graph and max_new_tokens > 1 else None
        B = st.batch
        finished = [False] * B

        def launch(n):
            if graph is not None:
                L.check(lib.srgpt_graph_launch(graph, n, stream))
            else:
                for _ in range(n):
                    L.check(lib.srgpt_llm_decode_step(C.byref(self.w.llm), C.byref(st.c), stream))

        def judge(ids, lo, hi):
            """host-side EOS / stopping-criteria scan of steps [lo, hi) of `ids` (CPU int64 [B, >= hi]); -> stop step or None."""
            for s_ in range(lo, hi):
                for b in range(B):
                    if eos and not finished[b] and int(ids[b, s_]) in eos:
                        finished[b] = True
                if eos and all(finished):
                    return s_ + 1
                if interactive:
                    # HF passes only the generated ids when generate() was fed inputs_embeds
                    full = ids[:, :s_ + 1]
                    for crit in stopping_criteria:
                        r = crit(full, None)
                        if bool(r.all()) if isinstance(r, torch.Tensor) else bool(r):
                            return s_ + 1
            return None

        if interactive or eos:
            # anything that can end the request early is judged EVERY step, one step behind the device.  Round 3 scanned for EOS
            # every `check_every` = 8 steps: a request that ends in EOS -- the reference's real eval mode, eval_spatial.py:221-237,
            # answers of a few dozen tokens -- ran 3.5 steps (10 ms at 2.9 ms per token) past its end on average; the run-ahead
            # loop wastes at most ONE step and costs 1.6 % while it runs (profiles/r04_sampling.txt)
            return self._decode_loop_run_ahead(st, max_new_tokens, launch, judge), eos
        # nothing can stop the request early (the benchmark: EOS disabled): all steps are queued at once, no host work per token
        if max_new_tokens > 1:
            launch(max_new_tokens - 1)
        return max_new_tokens, eos

    def _decode_loop_run_ahead(self, st: DecodeState, max_new_tokens: int, launch, judge) -> int:
        """A stopping criterion (the demo's KeywordsStoppingCriteria, gradio_web_server_multi.py:195-197; model_vqa.py's conv stop
        strings) is host Python over the ids so far: HF evaluates it after every token.  Round 3 did launch -> D2H -> Python ->
        launch, serialising the device behind the host every token.  Here step t + 1 is launched BEFORE step t is judged: the
        ids of step t travel on a side stream behind an event recorded after step t (so the copy does not wait for step t + 1),
        the criteria run while step t + 1 computes, and a stop discards the one step that ran ahead (it only wrote cache rows
        and ids beyond the kept prefix).  Same ids, same stop step as the serial loop."""
        B = st.batch
        if getattr(self, "_copy_stream", None) is None:
            self._copy_stream = torch.cuda.Stream(device=self.device)
        # One step's ids are the COLUMN out_ids[:, t]: a strided source for B > 1, and torch serves a strided D2H through a pageable
        # temporary plus a host-side copy that no event covers.  So the column is gathered into a contiguous device
        # scratch on the copy stream and leaves as ONE contiguous copy into a pinned, step-major buffer: the `done` event covers
        # everything the judge reads.
        host_t = torch.empty((max_new_tokens, B), dtype=torch.int64).pin_memory()
        host = host_t.t()  # [B, steps] view for the judge (HF criteria index ids[b, -n:])
        scratch = torch.empty((B,), dtype=torch.int64, device=self.device)
        assert host_t.is_pinned() and host_t[0].is_contiguous() and scratch.is_contiguous()
        dec = torch.cuda.current_stream(self.device)

        def mark():
            ev = torch.cuda.Event()
            ev.record(dec)
            return ev

        def fetch(col, ev):
            with torch.cuda.stream(self._copy_stream):
                self._copy_stream.wait_event(ev)
                scratch.copy_(st.out_ids[:, col])                 # device gather (copy stream: ordered behind the previous D2H)
                host_t[col].copy_(scratch, non_blocking=True)     # contiguous, pinned: truly asynchronous
                done = torch.cuda.Event()
                done.record(self._copy_stream)
            return done

        ev = mark()            # step 0 (srgpt_llm_sample_first) is on the stream
        produced = 1
        for t in range(max_new_tokens):
            ready = fetch(t, ev)
            if produced < max_new_tokens:  # run ahead: step t + 1 goes out before step t is judged
                launch(1)
                produced += 1
                ev = mark()
            ready.synchronize()
            if judge(host, t, t + 1) is not None:
                dec.wait_stream(self._copy_stream)
                return t + 1
        dec.wait_stream(self._copy_stream)
        return max_new_tokens
