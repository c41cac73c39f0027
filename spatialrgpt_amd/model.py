"""Drop-in Python surface of the reference VLM wrapper.

Mirrors `LlavaLlamaModel` (llava/model/language_model/llava_llama.py:48-213) and the accessor /
`prepare_inputs_labels_for_multimodal` surface of llava/model/llava_arch.py:252-650 -- same method names,
argument meaning and error behaviour -- on top of SrgptEngine (HIP kernels).  `LlavaLlamaForCausalLM`, the
name BASELINE.json uses, is an alias (the reference never defines it: SURVEY section 0).
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch

from .config import SrgptConfig
from .configuration import LlavaConfig, LlavaLlamaConfig, register_auto_classes
from .constants import IGNORE_INDEX
from .engine import SrgptEngine
from .generation import NOT_GIVEN, resolve_generation, warp_logits


class CausalLMOutputWithPast:
    """What the reference's forward() returns (llava_llama.py:177-192 -> transformers.modeling_outputs.CausalLMOutputWithPast):
    attribute access, key access, integer / slice access over the fields that are not None (HF's ModelOutput.to_tuple order:
    loss, logits, past_key_values, hidden_states, attentions), and `return_dict=False` -> that tuple.
    Seam differences, stated rather than hidden:
      * `past_key_values` is the engine's DecodeState (static KV cache + device bookkeeping), not a tuple of per-layer (k, v)
        tensors -- callers hand it back to forward()/generate unchanged, which is all the reference's callers do with it;
      * `attentions` is always None: the reference's LLM is built with FlashAttention2 (modeling_llama.py:615-618), which cannot
        return attention weights either (`output_attentions` is accepted and ignored there too)."""

    _fields = ("loss", "logits", "past_key_values", "hidden_states", "attentions")

    def __init__(self, loss=None, logits=None, past_key_values=None, hidden_states=None, attentions=None):
        self.loss, self.logits, self.past_key_values, self.hidden_states, self.attentions = \
            loss, logits, past_key_values, hidden_states, attentions

    def to_tuple(self):
        return tuple(getattr(self, k) for k in self._fields if getattr(self, k) is not None)

    def keys(self):
        return [k for k in self._fields if getattr(self, k) is not None]

    def __getitem__(self, k):
        if isinstance(k, str):
            if k not in self._fields or getattr(self, k) is None:
                raise KeyError(k)
            return getattr(self, k)
        return self.to_tuple()[k]

    def __contains__(self, k):
        return k in self.keys()

    def __iter__(self):
        return iter(self.keys())

    def __len__(self):
        return len(self.keys())

    def get(self, k, default=None):
        return self[k] if k in self else default


class _Facade:
    """nn.Module-looking handle (callers only use .to/.eval/.config/.is_loaded on sub-modules)."""

    is_loaded = True

    def to(self, *a, **k):
        return self

    def eval(self):
        return self

    def cuda(self):
        return self


class _VisionTower(_Facade):
    def __init__(self, eng: SrgptEngine, image_processor=None):
        self._eng = eng
        self.image_processor = image_processor
        c = eng.cfg
        # the loader records the special-token ids here (llava/model/builder.py:186-192)
        self.config = SimpleNamespace(hidden_size=c.vit_hidden, image_size=c.image_size, patch_size=c.patch_size,
                                      llm_mask_token_id=c.mask_token_id, llm_depth_token_id=c.depth_token_id)

    def __call__(self, images):
        # module-level contract (vision_encoder.py:115-132): features come back in the caller's image dtype
        if isinstance(images, list):
            return [self._eng.vit(im.unsqueeze(0), out_dtype=im.dtype) for im in images]
        return self._eng.vit(images, out_dtype=images.dtype)

    forward = __call__

    @property
    def device(self):
        return self._eng.device

    @property
    def dtype(self):
        return self._eng.dtype


class _RegionExtractor(_Facade):
    def __init__(self, eng: SrgptEngine):
        self._eng = eng
        self.config = SimpleNamespace(region_extractor_type=eng.cfg.region_extractor_type)

    def feature_refinement(self, tower_features):
        return self._eng.feature_refinement(tower_features.to(self._eng.dtype).contiguous())

    def __call__(self, image_features, depth_features, masks, *a, **k):
        dt = self._eng.dtype
        return self._eng.region_extractor(image_features.to(dt), None if depth_features is None else depth_features.to(dt), masks)

    forward = __call__


class _Projector(_Facade):
    def __init__(self, eng: SrgptEngine):
        self._eng = eng
        self.config = SimpleNamespace(mm_projector_type=eng.cfg.mm_projector_type)

    def __call__(self, x, *a, **k):
        return self._eng.mm_projector(x.to(self._eng.dtype).contiguous())

    forward = __call__


class _Llm(_Facade):
    def __init__(self, model: "LlavaLlamaModel"):
        self._m = model
        c = model.engine.cfg
        self.config = SimpleNamespace(hidden_size=c.hidden, vocab_size=c.vocab, num_hidden_layers=c.layers,
                                      tokenizer_model_max_length=c.tokenizer_model_max_length,
                                      tokenizer_padding_side=c.padding_side, eos_token_id=c.eos_token_id,
                                      pad_token_id=c.pad_token_id)

    def generate(self, inputs_embeds=None, attention_mask=None, **kw):
        return self._m._generate_from_embeds(inputs_embeds, attention_mask, **kw)

    def get_input_embeddings(self):
        return self._m.get_input_embeddings()


class LlavaLlamaModel:
    """`LlavaLlamaModel` of llava/model/language_model/llava_llama.py:48-213 on the MI355X engine.

    Two construction forms:
      * `LlavaLlamaModel(SrgptConfig, state_dict, ...)`           -- weights in memory (tests, benchmarks)
      * `LlavaLlamaModel(config=LlavaLlamaConfig, low_cpu_mem_usage=True, **kw)` / `LlavaLlamaModel.from_pretrained(path)` /
        `AutoModel.from_pretrained(path)`                         -- the reference's own sequence (builder.py:142-158): the
        checkpoint directory is `config._name_or_path` or `config.resume_path` (llava/model/utils.py:28-31)

    dtype policy: the engine holds its weights in ONE dtype (bfloat16 by default, float32 for bit-exact checks) and computes
    in it.  The reference's loader builds float16 (builder.py:62) and some callers feed float16 tensors without ever casting
    the model (eval_region_cls.py:316-317, model_vqa.py:71, the demo's non-bf16 branch): every float input is cast to the
    engine dtype at the boundary (exactly where the reference casts: vision_encoder.py:127, llava_llama.py:210), logits come
    back float32 (modeling_llama.py:1045), and `model.to(torch.float16)` / `.half()` / `.to(torch.float32)` are accepted as
    requests about the INTERFACE dtype only -- they never raise and never re-materialise weights (one warning per model)."""

    config_class = LlavaLlamaConfig
    main_input_name = "input_embeds"
    supports_gradient_checkpointing = True

    def __init__(self, config=None, state_dict: Dict[str, torch.Tensor] = None, device="cuda",
                 dtype=torch.bfloat16, tokenizer=None, image_processor=None, rope_positions: int = 0,
                 consume_state_dict: bool = False, llm_weight_format: str = "native", parts=None, **hf_kwargs):
        self.hf_config = None
        if isinstance(config, LlavaConfig):
            # the reference's construction path: config carries the checkpoint location
            from .builder import load_image_processor, load_tokenizer, read_checkpoint

            root = config.checkpoint_root()
            if root is None:
                raise ValueError("LlavaLlamaModel(config): config has neither `_name_or_path` nor `resume_path`")
            self.hf_config = config
            vr = config.vision_resolution if config.vision_resolution not in (None,) else -1
            cfg, state_dict = read_checkpoint(root, vr, config.interpolate_mode or "linear")
            tokenizer = load_tokenizer(root, cfg, state_dict)
            image_processor = load_image_processor(root, cfg)
            consume_state_dict = True
            md = hf_kwargs.pop("torch_dtype", None) or config.model_dtype
            if isinstance(md, str):
                md = getattr(torch, md.replace("torch.", ""))
            dtype = md if md in (torch.bfloat16, torch.float32) else dtype
            config = cfg
        elif state_dict is None:
            raise TypeError("LlavaLlamaModel needs (SrgptConfig, state_dict) or a LlavaLlamaConfig that points at a checkpoint")
        self.config = config
        self.engine = SrgptEngine(config, state_dict, device=device, dtype=dtype, rope_positions=rope_positions,
                                  consume_state_dict=consume_state_dict, llm_weight_format=llm_weight_format, parts=parts)
        self.tokenizer = tokenizer
        self.vision_tower = _VisionTower(self.engine, image_processor)
        self.region_extractor = _RegionExtractor(self.engine) if config.enable_region else None
        self.mm_projector = _Projector(self.engine)
        self.llm = _Llm(self)
        self.is_loaded = True
        self.training = False
        self.cache_reserve = 128  # forward(use_cache=True): cache positions reserved beyond the prompt
        self._warned_dtype = False

    @classmethod
    def from_pretrained(cls, pretrained_model_name_or_path, *model_args, config=None, **kwargs):
        """llava_llama.py:58-94: the entry `AutoModel.from_pretrained` lands on.  Only the arguments that mean something for
        this engine are honoured (`torch_dtype`, `device_map`/`device`, `load_in_8bit`); hub downloads are not attempted."""
        import os

        from .builder import load_model

        if not os.path.isdir(str(pretrained_model_name_or_path)):
            raise OSError(f"{pretrained_model_name_or_path} is not a local checkpoint directory")
        dtype = kwargs.pop("torch_dtype", None) or kwargs.pop("dtype", None) or torch.bfloat16
        if isinstance(dtype, str):
            dtype = getattr(torch, dtype.replace("torch.", ""))
        if dtype not in (torch.bfloat16, torch.float32):
            dtype = torch.bfloat16  # float16 requests: see the dtype policy above
        device = kwargs.pop("device", None) or "cuda"
        vr = getattr(config, "vision_resolution", None) if config is not None else None
        _, model, _ = load_model(str(pretrained_model_name_or_path), device=device, dtype=dtype,
                                 llm_weight_format="fp8" if kwargs.pop("load_in_8bit", False) else "native",
                                 vision_resolution=-1 if vr is None else vr)
        model.hf_config = config
        return model

    # ---- nn.Module / PreTrainedModel look-alikes used by the reference's callers ----
    @property
    def device(self):
        return self.engine.device

    @property
    def dtype(self):
        return self.engine.dtype

    def eval(self):
        return self

    def train(self, mode: bool = True):
        if mode:
            raise NotImplementedError("this is the inference path; training is out of scope")
        return self

    def requires_grad_(self, *a, **k):
        return self

    def cuda(self, *a, **k):
        return self

    def to(self, *args, **kwargs):
        """device moves are no-ops (the engine lives on its GPU); dtype requests follow the dtype policy of the class."""
        dtype = kwargs.get("dtype")
        for a in args:
            if isinstance(a, torch.dtype):
                dtype = a
        if dtype is not None and dtype != self.engine.dtype and not self._warned_dtype:
            import warnings

            warnings.warn(f"LlavaLlamaModel.to({dtype}): the MI355X engine keeps computing in {self.engine.dtype}; "
                          f"{dtype} inputs are cast at the boundary (documented dtype policy)")
            self._warned_dtype = True
        return self

    def half(self):
        return self.to(torch.float16)

    def bfloat16(self):
        return self.to(torch.bfloat16)

    def float(self):
        return self.to(torch.float32)

    def get_llm(self):
        return self.llm

    def get_lm_head(self):
        """the lm_head weight [vocab, hidden] (llava_arch.py accessors).  With fp8 LLM weights the bf16 matrix does not exist in HBM
        (the fp8 bytes are the only copy): it is rebuilt ONCE on first use (1 GB at 128k x 4096) and kept -- callers that poll the
        accessor must not allocate a fresh gigabyte per call."""
        w = self.engine.w
        if w.lm_head is not None:
            return w.lm_head
        if getattr(self, "_lm_head_dequantised", None) is None:
            self._lm_head_dequantised = w.dequantised("lm_head")
        return self._lm_head_dequantised

    def get_vision_tower(self):
        return self.vision_tower

    def get_mm_projector(self):
        return self.mm_projector

    def get_region_extractor(self):
        return self.region_extractor

    def get_input_embeddings(self):
        return self.engine.embed_tokens

    def resize_token_embeddings(self, embed_size):
        """llava_arch.py accessor -> HF `resize_token_embeddings` (the reference's loader calls it on the live model after adding
        `<mask>` / `<depth>`, builder.py:186-199): embed_tokens and lm_head to `embed_size` rows, common rows kept, new rows = the mean
        of the old ones (as the loader initialises the rows it adds).  Pooled decode states (their logits rows are vocabulary-sized) are dropped; a `past_key_values` handle
        returned before the resize must not be stepped afterwards."""
        embed_size = int(embed_size)
        if embed_size != self.engine.w.vocab:
            self.engine.w.resize_vocab(embed_size)
            self.engine._state = None
            self.config.vocab_size = embed_size
        return self.engine.embed_tokens

    def freezed_module_patch(self):
        return None

    def encode_images(self, images):
        return self.mm_projector(self.vision_tower(images))

    # ---- llava_arch.py:333-650 ----
    def prepare_inputs_labels_for_multimodal(self, input_ids, position_ids, attention_mask, past_key_values, labels,
                                             images, masks=None, depths=None):
        if images is None or input_ids.shape[1] == 1:
            if past_key_values is not None and images is not None and input_ids.shape[1] == 1:
                target = past_key_values.seq_len() + 1  # llava_arch.py:355-385
                attention_mask = torch.cat((attention_mask, torch.ones(
                    (attention_mask.shape[0], target - attention_mask.shape[1]), dtype=attention_mask.dtype,
                    device=attention_mask.device)), dim=1)
                position_ids = torch.sum(attention_mask, dim=1).unsqueeze(-1) - 1
            return input_ids, position_ids, attention_mask, past_key_values, None, labels
        if getattr(self.config, "turn_mm_projector", False) and self.config.mm_use_im_start_end:
            raise NotImplementedError
        res = self.engine.prepare_inputs(input_ids, images, depths, masks, attention_mask, labels=labels)
        embeds, am, lens = res[0], res[1], res[2]
        new_labels = res[3] if labels is not None else None  # llava_arch.py:513-533, :558-611
        pos = None
        if position_ids is not None:
            T = embeds.shape[1]
            pos = torch.zeros((len(lens), T), dtype=torch.long, device=embeds.device)
            for b, n in enumerate(lens):
                if self.config.padding_side == "left":
                    pos[b, T - n:] = torch.arange(n, device=embeds.device)
                else:
                    pos[b, :n] = torch.arange(n, device=embeds.device)
        return None, pos, am, past_key_values, embeds, new_labels

    # ---- llava_llama.py:100-192 (inference branch) ----
    def forward(self, input_ids=None, images=None, masks=None, depths=None, attention_mask=None, position_ids=None,
                past_key_values=None, seqlens_in_batch=None, inputs_embeds=None, labels=None, use_cache=None,
                output_attentions=None, output_hidden_states=None, return_dict=None, dpo_forward=False):
        # HF resolves use_cache=None from config.use_cache (LlamaConfig default True; the reference forwards to LlamaForCausalLM, which
        # does `use_cache if use_cache is not None else self.config.use_cache`, modeling_llama.py:1017-1020 of the vendored file)
        if use_cache is None:
            use_cache = bool(getattr(self.config, "use_cache", True))
        if past_key_values is not None:
            # incremental step over the state a previous forward(use_cache=True) returned (what HF's generate loop does with
            # the reference model: llava_arch.py:355-385 early-out, then the LLM with a cache)
            if input_ids is None or input_ids.shape[1] != 1 or labels is not None:
                raise NotImplementedError("forward() with past_key_values takes input_ids [B, 1] and no labels")
            logits = self.engine.step(past_key_values, input_ids)
            out = CausalLMOutputWithPast(logits=logits[:, None, :], past_key_values=past_key_values)
            if dpo_forward:
                return out.logits, None
            return out if return_dict is not False else out.to_tuple()
        if inputs_embeds is not None:
            inputs_embeds = inputs_embeds.to(self.dtype)
        if inputs_embeds is None:
            if images is None:
                inputs_embeds = self.engine.embed_tokens(input_ids)
            else:
                (_, position_ids, attention_mask, past_key_values, inputs_embeds, labels) = \
                    self.prepare_inputs_labels_for_multimodal(input_ids, position_ids, attention_mask, past_key_values,
                                                              labels, images, masks, depths)
        if attention_mask is None:
            # reference: `attention_mask.sum(-1)` on None -> AttributeError (SURVEY 3.2 gotcha)
            raise AttributeError("'NoneType' object has no attribute 'sum'")
        B, T, _ = inputs_embeds.shape
        # positions kept free after the prompt for incremental steps, clamped to what the RoPE table can still address
        reserve = max(1, min(self.cache_reserve, self.engine.w.rope_len - T)) if use_cache else 1
        keep = attention_mask.bool()
        ragged = not bool(keep.all())
        if ragged:
            # padded batch: valid positions packed to the front for the ragged prefill (causal attention never looks right),
            # logits scattered back to the caller's padded layout; padded positions are zeros (unspecified in the reference)
            lens = keep.sum(dim=1)
            # one stable argsort moves every row's valid positions to the front (order kept); its inverse scatters the
            # results back -- no per-row host sync, no boolean-index launches per row
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)
            valid = (torch.arange(T, device=keep.device)[None, :] < lens[:, None])
            packed = torch.gather(inputs_embeds, 1, order[:, :, None].expand(-1, -1, inputs_embeds.shape[2]))
            packed.masked_fill_(~valid[:, :, None], 0)
            st, plog, hs = self.engine.prefill(packed, max_new=reserve, all_logits=True,
                                               hidden_states=bool(output_hidden_states), lens=lens, fresh_state=bool(use_cache))
            # one logits-sized result, nothing else of that size (~1 GB per row at T = 2048, V = 128k -- the where /
            # zeros_like / scatter form held three): padded positions are zeroed in place, then every packed row is scattered to
            # its padded position -- `order` is a permutation of the T positions, so the scatter fills the whole output
            plog.masked_fill_(~valid[:, :, None], 0)
            logits = torch.empty_like(plog).scatter_(1, order[:, :, None].expand(-1, -1, plog.shape[2]), plog)
            del plog
            if hs is not None:
                hs.masked_fill_(~valid[None, :, :, None], 0)
                hs = torch.empty_like(hs).scatter_(2, order[None, :, :, None].expand(hs.shape[0], -1, -1, hs.shape[3]), hs)
        else:
            st, logits, hs = self.engine.prefill(inputs_embeds, max_new=reserve, all_logits=True,
                                                 hidden_states=bool(output_hidden_states), fresh_state=bool(use_cache))
        loss = None
        if labels is not None:
            # LlamaForCausalLM (modeling_llama.py:1047-1058): shift by one, mean CE over labels != IGNORE_INDEX
            from . import ops

            lab = labels.to(device=logits.device, dtype=torch.int64)
            shift_logits = logits[:, :-1, :].reshape(-1, logits.shape[-1])
            shift_labels = lab[:, 1:].reshape(-1)
            loss, _ = ops.cross_entropy(shift_logits, shift_labels, IGNORE_INDEX)
        out = CausalLMOutputWithPast(loss=loss, logits=logits, past_key_values=st if use_cache else None,
                                     hidden_states=None if hs is None else tuple(hs[i] for i in range(hs.shape[0])))
        if dpo_forward:
            return out.logits, labels
        return out if return_dict is not False else out.to_tuple()

    __call__ = forward

    # ---- llava_llama.py:194-213 ----
    @torch.no_grad()
    def generate(self, input_ids=None, images=None, depths=None, masks=None, attention_mask=None, **generation_kwargs):
        if images is not None:
            (_, _, attention_mask, _, inputs_embeds, _) = self.prepare_inputs_labels_for_multimodal(
                input_ids, None, attention_mask, None, None, images, masks, depths)
        else:
            inputs_embeds = self.engine.embed_tokens(input_ids)
        inputs_embeds = inputs_embeds.to(self.dtype)
        return self.llm.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask, **generation_kwargs)

    def generation_defaults(self) -> dict:
        """the stored generation config `llm.generate` starts from (HF: <ckpt>/llm/generation_config.json, else the generation
        fields of llm/config.json); in-memory models carry only the config's eos / pad ids."""
        c = self.config
        if getattr(c, "generation_config", None) is not None:
            return dict(c.generation_config)
        return {k: v for k, v in (("eos_token_id", c.eos_token_id), ("pad_token_id", c.pad_token_id)) if v is not None}

    def _generate_from_embeds(self, inputs_embeds, attention_mask=None, do_sample=NOT_GIVEN, temperature=NOT_GIVEN,
                              top_p=NOT_GIVEN, top_k=NOT_GIVEN, num_beams=NOT_GIVEN, max_new_tokens=NOT_GIVEN,
                              max_length=NOT_GIVEN, min_new_tokens=NOT_GIVEN, use_cache=True, stopping_criteria=None,
                              pad_token_id=NOT_GIVEN, eos_token_id=NOT_GIVEN, **unused):
        """HF `GenerationMixin.generate(inputs_embeds=...)` as the reference reaches it (llava_llama.py:212): the stored
        generation config overwritten by the call's keywords, explicit Nones included (spatialrgpt_amd/generation.py)."""
        g = resolve_generation(self.generation_defaults(), do_sample=do_sample, temperature=temperature, top_p=top_p, top_k=top_k,
                               num_beams=num_beams, max_new_tokens=max_new_tokens, max_length=max_length,
                               min_new_tokens=min_new_tokens, pad_token_id=pad_token_id, eos_token_id=eos_token_id)
        max_new_tokens = g.max_new_tokens
        eos_ids = g.eos_token_ids
        if g.min_new_tokens is not None and g.min_new_tokens >= max_new_tokens:
            eos_ids = None
        B, T, _ = inputs_embeds.shape
        lens = None
        if attention_mask is not None and not bool(attention_mask.bool().all()):
            # ragged batch (llava_arch.py:549-611 pads to the longest row, right or left): pack every row's valid positions
            # to the front (right padding) and hand the lengths to the ragged prefill; decode is batched with one position
            # per sequence, exactly like the reference's padded generate()
            keep = attention_mask.bool()
            lens = keep.sum(dim=1)
            lens_h = [int(v) for v in lens.tolist()]  # ONE device->host copy for the whole batch
            if min(lens_h) <= 0:
                raise ValueError("generate: a row of the batch has no valid position")
            Tmax = max(lens_h)
            # stable sort of the keep flags moves every row's valid positions to the front, order preserved: one gather
            order = torch.argsort((~keep).to(torch.int8), dim=1, stable=True)[:, :Tmax]
            packed = torch.gather(inputs_embeds, 1, order[:, :, None].expand(-1, -1, inputs_embeds.shape[2]))
            packed = torch.where((torch.arange(Tmax, device=packed.device)[None, :] < lens[:, None])[:, :, None], packed,
                                 torch.zeros_like(packed))
            inputs_embeds = packed.contiguous()
            if min(lens_h) == Tmax:
                lens = None
        if g.num_beams != 1:
            # `--num_beams N` of the eval CLIs; with their default `--temperature 0.2` (do_sample = temperature > 0) this is HF's
            # beam-SAMPLE, with `--temperature 0` beam search proper
            sample = bool(g.do_sample and g.temperature is not None and g.temperature > 0)
            return self._beam_search(inputs_embeds, lens, g.num_beams, max_new_tokens, eos_ids, g.pad_token_id, stopping_criteria,
                                     sampling=dict(temperature=g.temperature, top_k=g.top_k, top_p=g.top_p) if sample else None)
        st, _, _ = self.engine.prefill(inputs_embeds, max_new=max_new_tokens, lens=lens)
        if g.do_sample and g.temperature is not None and g.temperature > 0:
            from . import ops

            if ops.SamplingParams.supported(g.temperature, g.top_k, g.top_p, self.engine.w.vocab):
                # the draw runs INSIDE the captured decode step (csrc/sample.hip): temperature -> top-k -> top-p -> categorical on a
                # Philox stream seeded from torch's default CPU generator -- torch.manual_seed / transformers.set_seed make a
                # request reproducible, as with HF's torch.multinomial (whose stream itself cannot be matched)
                seed = int(torch.randint(0, 2 ** 62, (1,)).item())
                return self.engine.greedy_decode(st, max_new_tokens, eos_token_id=eos_ids, pad_token_id=g.pad_token_id,
                                                 stopping_criteria=stopping_criteria,
                                                 sampling=dict(temperature=g.temperature, top_k=g.top_k, top_p=g.top_p, seed=seed))
            # settings the device sampler does not serve (top_k > 64; top-p without top-k): warpers + draw as torch ops per token
            return self._sample_loop(st, max_new_tokens, g.temperature, g.top_p, g.top_k, eos_ids, g.pad_token_id,
                                     stopping_criteria)
        return self.engine.greedy_decode(st, max_new_tokens, eos_token_id=eos_ids, pad_token_id=g.pad_token_id,
                                         stopping_criteria=stopping_criteria)

    def _beam_search(self, inputs_embeds, lens, num_beams, max_new_tokens, eos_ids, pad_token_id, stopping_criteria, sampling=None):
        """`generate(num_beams > 1)` -- the `--num_beams` flag of eval_spatial.py:234, eval_region_cls.py:321, model_vqa.py:75 (default
        1: the benchmarked path is the greedy loop); `sampling` (temperature / top_k / top_p) makes it beam-SAMPLE, which is what those
        CLIs ask for under their default `--temperature 0.2`.  HF 4.37.2 semantics in spatialrgpt_amd/generation.beam_generate; here
        only the device side: every batch item's prompt is prefilled once per beam (HF's `_expand_inputs_for_generation`), each step
        runs the HIP decode step on batch * num_beams rows, and the live KV-cache rows follow the beam indices the search picked
        (srgpt_kv_beam_reorder: in place, one launch).
        The beam bookkeeping itself (log-softmax, warpers, top-k / Gumbel-top-k over num_beams * vocab, a [batch, 2 * num_beams]
        hand-over to the host scorer per step) is torch: a rarely used, latency-tolerant mode -- not part of the measured path."""
        import ctypes as C

        from . import _lib as L
        from . import ops
        from .generation import beam_generate

        eng = self.engine
        B = inputs_embeds.shape[0]
        emb = inputs_embeds.repeat_interleave(num_beams, dim=0)
        lens_x = None if lens is None else lens.repeat_interleave(num_beams, dim=0)
        st, _, _ = eng.prefill(emb, max_new=max_new_tokens, lens=lens_x)
        ident = torch.arange(B * num_beams, device=self.device)

        def step(tokens, beam_idx):
            # rows of the next step continue from the caches of the beams the search kept.  Only the LIVE positions move, and
            # nothing moves when the permutation is the identity (the whole-capacity index_select built a cache-sized
            # temporary for K and again for V every token)
            if not torch.equal(beam_idx, ident):
                n = max(st.host_len)
                if not ops.kv_beam_reorder(st.kcache, st.vcache, beam_idx, num_beams, n):  # (> 8 beams: torch ops)
                    st.kcache[:, :, :, :n].copy_(st.kcache[:, :, :, :n].index_select(1, beam_idx))
                    st.vcache[:, :, :, :n].copy_(st.vcache[:, :, :, :n].index_select(1, beam_idx))
            return eng.step(st, tokens[:, None])

        gen = None
        kw = {}
        if sampling is not None:
            # seeded from torch's default CPU generator, like the sampling path: torch.manual_seed makes a request reproducible
            gen = torch.Generator(device=self.device)
            gen.manual_seed(int(torch.randint(0, 2 ** 62, (1,)).item()))
            kw = dict(do_sample=True, temperature=sampling["temperature"], top_k=sampling["top_k"], top_p=sampling["top_p"], generator=gen)
        out = beam_generate(st.logits.clone(), step, B, num_beams, max_new_tokens, eos_ids, pad_token_id,
                            stopping_criteria=stopping_criteria, **kw)
        # the decode attention's hand-off health (arrival tickets re-armed), as on the greedy path
        L.check(L.load().srgpt_llm_decode_sync_state(C.byref(eng.w.llm), C.byref(st.c), ops._stream()))
        return out

    def _sample_loop(self, st, max_new_tokens, temperature, top_p, top_k, eos_token_id, pad_token_id, stopping_criteria,
                     check_every: int = 8):
        """temperature / top-k / top-p sampling for the settings the device sampler (csrc/sample.hip) does not serve -- top_k > 64, or
        a top-p filter without top-k; every setting of the reference's callers (demo: temperature 0.2, top_k 50; model_vqa.py) runs on
        the device instead.  The transformer steps are the HIP decode step; the warpers (generation.warp_logits, pinned to HF's) and
        the categorical draw over the final logits use torch.  Host round trips: none per step unless a stopping criterion is given
        (HF semantics need the ids on the host then); the all-rows-finished test is read back every `check_every` steps."""
        import ctypes as C

        from . import _lib as L
        from . import ops

        eng, lib = self.engine, L.load()
        eos = None if eos_token_id is None else [int(e) for e in ([eos_token_id] if isinstance(eos_token_id, int) else eos_token_id)]
        B = st.batch
        finished = torch.zeros(B, dtype=torch.bool, device=self.device)
        eos_t = None if not eos else torch.tensor(eos, device=self.device, dtype=torch.int64)
        padv = None if not eos else (pad_token_id if pad_token_id is not None else eos[0])
        out = torch.empty((B, max_new_tokens), dtype=torch.int64, device=self.device)
        n = 0
        for step in range(max_new_tokens):
            if step > 0:
                L.check(lib.srgpt_llm_decode_step(C.byref(eng.w.llm), C.byref(st.c), ops._stream()))
            probs = warp_logits(st.logits, temperature, top_k, top_p).softmax(-1)
            tok = torch.multinomial(probs, 1).squeeze(1)
            if eos:
                tok = torch.where(finished, torch.full_like(tok, padv), tok)
            st.tok.copy_(tok)  # the next decode step embeds this token
            if step == 0:
                st.step.zero_()
            out[:, step] = tok
            n = step + 1
            if eos:
                finished |= (tok[:, None] == eos_t[None, :]).any(dim=1)
            if stopping_criteria:
                ids = out[:, :n].cpu()
                stop = False
                for crit in stopping_criteria:
                    r = crit(ids, None)
                    stop |= bool(r.all()) if isinstance(r, torch.Tensor) else bool(r)
                if stop:
                    break
            if eos and (n % check_every == 0 or n == max_new_tokens) and bool(finished.all()):
                # rows finished somewhere inside the last `check_every` steps: everything after a row's EOS is already pad, so
                # the output is cut at the first column where every row had finished
                done_col = (out[:, :n, None] == eos_t[None, None, :]).any(-1).int().argmax(dim=1)  # first EOS per row
                n = int(done_col.max()) + 1
                break
        return out[:, :n].clone()


LlavaLlamaForCausalLM = LlavaLlamaModel

register_auto_classes(LlavaLlamaModel)  # llava_llama.py:216-217
