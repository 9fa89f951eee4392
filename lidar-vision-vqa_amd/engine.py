"""Inference engine on the HIP modules.

Caller contract kept from the reference's `InferenceEngine` (encoder-decoder/inference/inference_engine.py:12-336): the
constructor takes the `models` dict of `ModelLoader.load_all()`; `format_prompt`, `process_lidar`, `process_vision`,
`build_inputs_embeds`, `generate`, `generate_batch` keep their names, arguments and results -- the prompt strings, the marker
tokens and the embedding layout are the interface a checkpoint was trained against, and they are pinned by goldens taken from
the unmodified reference class (tools/make_engine_golden.py).  Everything behind that contract is organised for this build:

  * decoding settings travel as one `Decoding` value; `base_model.generate` may be `StandInHead.generate` (greedy or sampled:
    the reference's DEFAULT call is do_sample=True, temperature 0.7, top_k 50, top_p 0.9 -> lvq_sample_rows) or any
    transformers-style model object;
  * the modal splice is a generic "replace marker pairs" pass over the prompt ids (`_splice`);
  * a stored BEV (fp16 `.npy`, the reference's on-disk format) crosses PCIe as fp16 and is up-cast on the device (`bev.f16_to_f32`).

Two deliberate differences from the reference:
  * it decodes `outputs[0][inputs_embeds.shape[1]:]`, but an inputs_embeds-only `generate` returns ONLY the new tokens, so that
    slice is empty whenever max_new_tokens < prompt length and the reference answers "" (recorded in tests/golden/engine.npz).
    The new tokens are what is decoded here.
  * `process_vision` needs the SAM / CLIP DeepEncoder towers (weights fetched by URL: out of scope); the six
    per-view token tensors come from `models["multiview_tokens_fn"](sample_token)` instead of `runtime` / `nusc`.
"""
from __future__ import annotations

from dataclasses import dataclass
from pathlib import Path
from typing import Dict, List, Optional, Sequence, Tuple, Union

import numpy as np
import torch

from . import _ffi as F
from . import bev as B

BevInput = Union[torch.Tensor, np.ndarray, str, Path]
MODALITIES = ("vision", "lidar")            # splice order = the order the markers appear in format_prompt's output


@dataclass(frozen=True)
class Decoding:
    """Settings of one `generate` call, with the reference's defaults (inference_engine.py:236-240)."""
    max_new_tokens: int = 64
    temperature: float = 0.7
    top_p: float = 0.9
    top_k: int = 50
    do_sample: bool = True
    num_beams: int = 1

    def kwargs(self, tokenizer) -> dict:
        """Keyword block for `base_model.generate`; a greedy call carries the neutral warper values, as the reference sends them."""
        warp = dict(temperature=self.temperature, top_p=self.top_p, top_k=self.top_k) if self.do_sample else \
            dict(temperature=1.0, top_p=1.0, top_k=50)
        return dict(max_new_tokens=self.max_new_tokens, do_sample=self.do_sample, num_beams=self.num_beams,
                    pad_token_id=tokenizer.pad_token_id, eos_token_id=tokenizer.eos_token_id, **warp)


def _splice(ids: torch.Tensor, text: torch.Tensor, inserts: Sequence[Tuple[int, int, torch.Tensor]]) -> torch.Tensor:
    """ids [n], text [1, n, d].  For every (start_id, end_id, rows [1, m, d]) in order: the first start marker and the first end
    marker in `ids` stay, `rows` go between them, whatever text sat between them is dropped; a pair with a missing marker is
    skipped.  Text outside the pairs is kept.  Row gathers and one concatenation -- no arithmetic."""
    out, cursor = [], 0
    for start_id, end_id, rows in inserts:
        s = torch.nonzero(ids == start_id).flatten()
        e = torch.nonzero(ids == end_id).flatten()
        if s.numel() == 0 or e.numel() == 0:
            continue
        s0, e0 = int(s[0]), int(e[0])
        out += [text[:, cursor:s0 + 1], rows, text[:, e0:e0 + 1]]
        cursor = e0 + 1
    out.append(text[:, cursor:])
    return torch.cat([t for t in out if t.shape[1] > 0], dim=1)


class InferenceEngine:
    def __init__(self, models: Dict):
        need = ("tokenizer", "base_model", "vat_lidar", "config", "device", "d_model")
        missing = [k for k in need if k not in models]
        if missing:
            raise KeyError(f"InferenceEngine: models dict lacks {missing}")
        self.tokenizer, self.base_model, self.vat_lidar = models["tokenizer"], models["base_model"], models["vat_lidar"]
        self.vat_vision, self.vision_adapter = models.get("vat_vision"), models.get("vision_adapter")
        self.runtime, self.nusc = models.get("runtime"), models.get("nusc")
        self.multiview_tokens_fn = models.get("multiview_tokens_fn")
        self.config, self.device, self.d_model = models["config"], models["device"], models["d_model"]
        self.use_vision = bool(self.config.get("use_vision", False)) and self.vat_vision is not None
        self.prefix_scale = self.config.get("prefix_scale", 0.2)
        self.system_prompt = self.config.get("system_prompt", "")
        tid = self.tokenizer.convert_tokens_to_ids
        self.marker_ids = {m: (tid(f"<{m}_start>"), tid(f"<{m}_end>")) for m in MODALITIES}
        (self.vision_start_id, self.vision_end_id), (self.lidar_start_id, self.lidar_end_id) = (self.marker_ids[m] for m in MODALITIES)

    # ---- prompt text (the strings a checkpoint is trained against: inference_engine.py:48-70) ----
    def format_prompt(self, question: str, include_vision: bool = True) -> str:
        body = f"{self.system_prompt}\n\n{question}" if self.system_prompt else question
        mods = MODALITIES if (self.use_vision and include_vision) else MODALITIES[1:]
        return "".join(f"<{m}_start><{m}_end>" for m in mods) + f"{body}\nAnswer:"

    # ---- modal prefixes ----
    @torch.no_grad()
    def process_lidar(self, bev: torch.Tensor) -> torch.Tensor:
        """bev [B, C, H, W] or [C, H, W] -> LiDAR prompts [B, n_queries, d_model] (inference_engine.py:72-102)."""
        return self.vat_lidar((bev if bev.ndim == 4 else bev[None]).to(self.device))

    @torch.no_grad()
    def process_vision(self, sample_token: str) -> Optional[torch.Tensor]:
        """Six camera views -> VisionAdapter -> VATVision prompts [1, n_queries, d_model]; None when vision is off
        (inference_engine.py:104-137)."""
        if not self.use_vision:
            return None
        if self.multiview_tokens_fn is None:
            raise F.LvqError("process_vision: pass models['multiview_tokens_fn'](sample_token) -> six [HW, d_in] tensors; the "
                             "DeepEncoder towers are outside this build (DESIGN section 6)")
        views = [t.to(self.device) for t in self.multiview_tokens_fn(sample_token)]
        return self.vat_vision(self.vision_adapter(views)[None])

    def _load_bev(self, bev: BevInput) -> torch.Tensor:
        if isinstance(bev, (str, Path)):
            bev = np.load(bev)
        if isinstance(bev, np.ndarray):
            if bev.dtype == np.float16:
                return B.f16_to_f32(torch.from_numpy(np.ascontiguousarray(bev)).to(self.device))
            return torch.from_numpy(bev).float()
        return bev

    # ---- prompt embeddings with the scaled prefixes between their markers (inference_engine.py:139-227) ----
    @torch.no_grad()
    def build_inputs_embeds(self, prompt: str, lidar_prompts: torch.Tensor, vision_prompts: Optional[torch.Tensor] = None) -> tuple:
        """Returns (inputs_embeds [1, L, d], all-ones attention mask [1, L])."""
        ids = self.tokenizer(prompt, return_tensors="pt", add_special_tokens=False)["input_ids"].to(self.device)
        text = self.base_model.get_input_embeddings()(ids)
        given = {"vision": vision_prompts if self.use_vision else None, "lidar": lidar_prompts}
        inserts = [(*self.marker_ids[m], given[m] * self.prefix_scale) for m in MODALITIES if given[m] is not None]
        emb = _splice(ids[0], text, inserts)
        return emb, torch.ones(emb.shape[:2], dtype=torch.long, device=self.device)

    # ---- answers ----
    def _prefixes(self, bev: BevInput, sample_token: Optional[str]) -> Tuple[torch.Tensor, Optional[torch.Tensor]]:
        lidar = self.process_lidar(self._load_bev(bev))
        vision = None
        if self.use_vision and sample_token is not None:
            try:
                vision = self.process_vision(sample_token)
            except Exception as err:       # LiDAR-only answer instead of no answer, like the reference (inference_engine.py:262-268)
                print(f"[engine] vision prefix unavailable for {sample_token!r} ({err}); answering from LiDAR only")
        return lidar, vision

    @torch.no_grad()
    def answer(self, question: str, bev: BevInput, sample_token: Optional[str], decoding: Decoding, generator=None) -> str:
        lidar, vision = self._prefixes(bev, sample_token)
        prompt = self.format_prompt(question, include_vision=vision is not None)
        emb, mask = self.build_inputs_embeds(prompt, lidar, vision)
        extra = {"generator": generator} if generator is not None else {}
        new_ids = self.base_model.generate(inputs_embeds=emb, attention_mask=mask, **decoding.kwargs(self.tokenizer), **extra)
        return self.tokenizer.decode(new_ids[0].tolist(), skip_special_tokens=True).strip()

    def generate(self, question: str, bev: BevInput, sample_token: Optional[str] = None, max_new_tokens: int = 64,
                 temperature: float = 0.7, top_p: float = 0.9, top_k: int = 50, do_sample: bool = True, num_beams: int = 1,
                 generator: Optional[torch.Generator] = None) -> str:
        """The reference's signature and defaults (inference_engine.py:229-240); `generator` seeds the sampled path."""
        return self.answer(question, bev, sample_token, Decoding(max_new_tokens, temperature, top_p, top_k, do_sample, num_beams), generator)

    def generate_batch(self, questions: List[str], bevs: List[BevInput], sample_tokens: Optional[List[str]] = None,
                       **generation_kwargs) -> List[str]:
        """One answer per (question, bev[, sample_token]) triple, in order (inference_engine.py:306-336)."""
        tokens = sample_tokens if sample_tokens is not None else [None] * len(questions)
        return [self.generate(q, b, t, **generation_kwargs) for q, b, t in zip(questions, bevs, tokens)]
