"""Inference engine: the reference's `InferenceEngine` API (encoder-decoder/inference/
inference_engine.py:12-336) on the HIP modules -- prompt formatting, LiDAR / vision prefix processing, interleaving of
text and modal prompt embeddings, greedy generation.

The class takes the same `models` dict as the reference (`ModelLoader.load_all()`, model_loader.py): tokenizer,
base_model, vat_lidar, vat_vision, vision_adapter, runtime, nusc, config, device, d_model.  Differences, all deliberate:
  * `base_model.generate` is StandInHead.generate: greedy decoding (do_sample=False, num_beams=1); the reference's defaults
    (do_sample=True, temperature 0.7) raise LvqError here instead of silently changing behaviour.
  * The reference decodes `outputs[0][inputs_embeds.shape[1]:]`; transformers returns ONLY the new tokens for an
    inputs_embeds-only call, so that slice is empty whenever max_new_tokens < prompt length and the reference's generate()
    returns "" (verified against the unmodified class in tools/make_engine_golden.py).  The evident intent -- decode the
    new tokens -- is what this class does.
  * process_vision needs the SAM/CLIP DeepEncoder towers (pretrained weights fetched by URL: out of scope); the
    six per-view token tensors come from `models["multiview_tokens_fn"](sample_token)` instead of `runtime` / `nusc`.
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, List, Optional, Union

import numpy as np
import torch

from . import _ffi as F
from . import bev as B


class InferenceEngine:
    def __init__(self, models: Dict):
        self.tokenizer = models["tokenizer"]
        self.base_model = models["base_model"]
        self.vat_lidar = models["vat_lidar"]
        self.vat_vision = models.get("vat_vision")
        self.vision_adapter = models.get("vision_adapter")
        self.runtime = models.get("runtime")
        self.nusc = models.get("nusc")
        self.multiview_tokens_fn = models.get("multiview_tokens_fn")
        self.config = models["config"]
        self.device = models["device"]
        self.d_model = models["d_model"]
        self.use_vision = self.config.get("use_vision", False) and self.vat_vision is not None
        self.prefix_scale = self.config.get("prefix_scale", 0.2)
        self.system_prompt = self.config.get("system_prompt", "")
        self.lidar_start_id = self.tokenizer.convert_tokens_to_ids("<lidar_start>")
        self.lidar_end_id = self.tokenizer.convert_tokens_to_ids("<lidar_end>")
        self.vision_start_id = self.tokenizer.convert_tokens_to_ids("<vision_start>")
        self.vision_end_id = self.tokenizer.convert_tokens_to_ids("<vision_end>")

    def format_prompt(self, question: str, include_vision: bool = True) -> str:
        if self.system_prompt:
            question = f"{self.system_prompt}\n\n{question}"
        if self.use_vision and include_vision:
            return f"<vision_start><vision_end><lidar_start><lidar_end>{question}\nAnswer:"
        return f"<lidar_start><lidar_end>{question}\nAnswer:"

    @torch.no_grad()
    def process_lidar(self, bev: torch.Tensor) -> torch.Tensor:
        """bev [B,C,H,W] or [C,H,W] -> LiDAR prompts [B, n_queries, d_model]."""
        if bev.ndim == 3:
            bev = bev.unsqueeze(0)
        return self.vat_lidar(bev.to(self.device))

    @torch.no_grad()
    def process_vision(self, sample_token: str) -> Optional[torch.Tensor]:
        if not self.use_vision:
            return None
        if self.multiview_tokens_fn is None:
            raise F.LvqError("process_vision: pass models['multiview_tokens_fn'](sample_token) -> six [HW, d_in] tensors; the "
                             "DeepEncoder towers are outside this build (DESIGN §6)")
        vt = [t.to(self.device) for t in self.multiview_tokens_fn(sample_token)]
        kv_tokens = self.vision_adapter(vt).unsqueeze(0)             # [1, 6*HW, d_in]
        return self.vat_vision(kv_tokens)

    @torch.no_grad()
    def build_inputs_embeds(self, prompt: str, lidar_prompts: torch.Tensor, vision_prompts: Optional[torch.Tensor] = None) -> tuple:
        """Text embeddings with the scaled modal prompts spliced in between their marker tokens (inference_engine.py:139-227):
        for vision (when enabled and given) and then LiDAR, the first <x_start> / <x_end> pair found in the prompt ids is
        replaced by  E(<x_start>), prompts * prefix_scale, E(<x_end>);  text in front of, between and behind the pairs is kept.
        The pieces are row gathers, a scale and a concatenation.  Returns (inputs_embeds [1, L, d], all-ones mask [1, L])."""
        ids = self.tokenizer(prompt, return_tensors="pt", add_special_tokens=False)["input_ids"].to(self.device)
        text = self.base_model.get_input_embeddings()(ids)                  # [1, n_text, d]
        row = ids[0]
        modal = []
        if self.use_vision and vision_prompts is not None:
            modal.append((self.vision_start_id, self.vision_end_id, vision_prompts))
        modal.append((self.lidar_start_id, self.lidar_end_id, lidar_prompts))
        chunks, cursor = [], 0
        for start_id, end_id, prompts in modal:
            starts = torch.nonzero(row == start_id).flatten()
            ends = torch.nonzero(row == end_id).flatten()
            if starts.numel() == 0 or ends.numel() == 0:
                continue                                                     # marker pair absent: this modality is not spliced
            s0, e0 = int(starts[0]), int(ends[0])
            if s0 > cursor:
                chunks.append(text[:, cursor:s0])
            chunks += [text[:, s0:s0 + 1], prompts * self.prefix_scale, text[:, e0:e0 + 1]]
            cursor = e0 + 1
        if cursor < text.shape[1]:
            chunks.append(text[:, cursor:])
        inputs_embeds = torch.cat(chunks, dim=1)
        mask = torch.ones((1, inputs_embeds.shape[1]), dtype=torch.long, device=self.device)
        return inputs_embeds, mask

    def _load_bev(self, bev) -> torch.Tensor:
        if isinstance(bev, (str, Path)):
            bev = np.load(bev)
        if isinstance(bev, np.ndarray):
            if bev.dtype == np.float16:                                   # stored format: fp16 over PCIe, exact up-cast on the device
                return B.f16_to_f32(torch.from_numpy(np.ascontiguousarray(bev)).to(self.device))
            bev = torch.from_numpy(bev).float()
        return bev

    @torch.no_grad()
    def generate(self, question: str, bev: Union[torch.Tensor, np.ndarray, str, Path], sample_token: Optional[str] = None,
                 max_new_tokens: int = 64, temperature: float = 0.7, top_p: float = 0.9, top_k: int = 50, do_sample: bool = True,
                 num_beams: int = 1) -> str:
        lidar_prompts = self.process_lidar(self._load_bev(bev))
        vision_prompts = None
        include_vision = False
        if self.use_vision and sample_token is not None:
            try:
                vision_prompts = self.process_vision(sample_token)
                include_vision = True
            except Exception as e:                                         # the reference degrades to LiDAR-only the same way
                print(f"[engine] Warning: Failed to process vision: {e}")
        prompt = self.format_prompt(question, include_vision=include_vision)
        inputs_embeds, attention_mask = self.build_inputs_embeds(prompt, lidar_prompts, vision_prompts)
        outputs = self.base_model.generate(inputs_embeds=inputs_embeds, attention_mask=attention_mask, max_new_tokens=max_new_tokens,
                                           temperature=temperature if do_sample else 1.0, top_p=top_p if do_sample else 1.0,
                                           top_k=top_k if do_sample else 50, do_sample=do_sample, num_beams=num_beams,
                                           pad_token_id=self.tokenizer.pad_token_id, eos_token_id=self.tokenizer.eos_token_id)
        answer = self.tokenizer.decode(outputs[0].tolist(), skip_special_tokens=True)      # the new tokens (see the module docstring)
        return answer.strip()

    @torch.no_grad()
    def generate_batch(self, questions: List[str], bevs: List[Union[torch.Tensor, np.ndarray, str, Path]],
                       sample_tokens: Optional[List[str]] = None, **generation_kwargs) -> List[str]:
        if sample_tokens is None:
            sample_tokens = [None] * len(questions)
        return [self.generate(q, bev, token, **generation_kwargs) for q, bev, token in zip(questions, bevs, sample_tokens)]
