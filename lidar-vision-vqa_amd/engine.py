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
        """Interleave text embeddings and the scaled modal prompts between their start / end tokens
        (inference_engine.py:139-227).  The pieces are row gathers, a scale and a concatenation."""
        enc = self.tokenizer(prompt, return_tensors="pt", add_special_tokens=False)
        input_ids = enc["input_ids"].to(self.device)
        text_embeds = self.base_model.get_input_embeddings()(input_ids)
        ids_flat = input_ids[0]
        embeds_list = []
        pos = 0
        if self.use_vision and vision_prompts is not None:
            vs_pos = (ids_flat == self.vision_start_id).nonzero(as_tuple=True)[0]
            ve_pos = (ids_flat == self.vision_end_id).nonzero(as_tuple=True)[0]
            if len(vs_pos) > 0 and len(ve_pos) > 0:
                vs, ve = vs_pos[0].item(), ve_pos[0].item()
                if vs > pos:
                    embeds_list.append(text_embeds[:, pos:vs, :])
                embeds_list.append(text_embeds[:, vs:vs + 1, :])
                embeds_list.append(vision_prompts * self.prefix_scale)
                embeds_list.append(text_embeds[:, ve:ve + 1, :])
                pos = ve + 1
        ls_pos = (ids_flat == self.lidar_start_id).nonzero(as_tuple=True)[0]
        le_pos = (ids_flat == self.lidar_end_id).nonzero(as_tuple=True)[0]
        if len(ls_pos) > 0 and len(le_pos) > 0:
            ls, le = ls_pos[0].item(), le_pos[0].item()
            if ls > pos:
                embeds_list.append(text_embeds[:, pos:ls, :])
            embeds_list.append(text_embeds[:, ls:ls + 1, :])
            embeds_list.append(lidar_prompts * self.prefix_scale)
            embeds_list.append(text_embeds[:, le:le + 1, :])
            pos = le + 1
        if pos < text_embeds.shape[1]:
            embeds_list.append(text_embeds[:, pos:, :])
        inputs_embeds = torch.cat(embeds_list, dim=1)
        attention_mask = torch.ones(1, inputs_embeds.shape[1], dtype=torch.long, device=self.device)
        return inputs_embeds, attention_mask

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
