#!/usr/bin/env python3
"""Golden vectors for the inference engine (inference/inference_engine.py:139-304): the UNMODIFIED reference
`InferenceEngine` is instantiated in the build container with the reference's own VATLiDAR / VATVision (CPU), transformers'
Qwen2ForCausalLM as base model (seeded stand-in weights) and the character-level DummyTokenizer, and its format_prompt /
process_lidar / build_inputs_embeds / generate(do_sample=False) are run.  Outputs (data only) -> tests/golden/engine.npz.

    python tools/make_engine_golden.py
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import cases  # noqa: E402
import make_goldens as MG  # noqa: E402
from lidar_vision_vqa_amd import synth  # noqa: E402

QUESTION = "How many cars are ahead of the ego vehicle?"
SYSTEM = "You are a driving assistant."
N_NEW = 8


def main():
    from transformers import Qwen2Config, Qwen2ForCausalLM
    R = MG.import_reference()
    spec = importlib.util.spec_from_file_location("ref_inference_engine", os.path.join(MG.REF, "encoder-decoder", "inference", "inference_engine.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    hc = cases.HEAD_CASE
    d = hc["d"]
    cfg = Qwen2Config(vocab_size=hc["vocab"], hidden_size=d, intermediate_size=hc["inter"], num_attention_heads=hc["n_heads"],
                      num_key_value_heads=hc["n_kv_heads"], num_hidden_layers=hc["n_layers"], tie_word_embeddings=True,
                      rms_norm_eps=hc["rms_eps"], rope_theta=hc["rope_theta"], max_position_embeddings=512, attn_implementation="eager")
    base = Qwen2ForCausalLM(cfg).eval()
    sd = {k: torch.from_numpy(synth.seeded_array(k, tuple(v.shape), hc["seed"])) for k, v in base.state_dict().items() if k != "lm_head.weight"}
    sd["lm_head.weight"] = sd["model.embed_tokens.weight"]
    base.load_state_dict(sd)
    vl = synth.load_seeded(R.VATLiDAR(16, d, hc["nq_lidar"], 1, 4).eval(), hc["seed"] + 1)
    vv = synth.load_seeded(R.VATVision(64, d, 48, 2, 1, 4).eval(), hc["seed"] + 3)
    va = synth.load_seeded(R.VisionAdapter(64, 0.1).eval(), hc["seed"] + 2)
    tok = synth.DummyTokenizer(hc["vocab"])
    out = {}
    for use_vision in (False, True):
        models = dict(tokenizer=tok, base_model=base, vat_lidar=vl, vat_vision=vv if use_vision else None, vision_adapter=va,
                      runtime=None, nusc=None, device=torch.device("cpu"), d_model=d,
                      config=dict(use_vision=use_vision, prefix_scale=0.2, system_prompt=SYSTEM if use_vision else ""))
        eng = mod.InferenceEngine(models)
        bev = torch.from_numpy(synth.randn((16, 10, 10), hc["seed"] + 40))            # [C,H,W]: process_lidar adds the batch axis
        with torch.no_grad():
            lp = eng.process_lidar(bev)
            vp = None
            if use_vision:
                kv = va([torch.from_numpy(synth.randn((8, 64), hc["seed"] + 50 + v)) for v in range(6)]).unsqueeze(0)
                vp = vv(kv)
            prompt = eng.format_prompt(QUESTION, include_vision=use_vision)
            emb, attn = eng.build_inputs_embeds(prompt, lp, vp)
            ids = base.generate(inputs_embeds=emb, attention_mask=attn, max_new_tokens=N_NEW, do_sample=False, num_beams=1,
                                pad_token_id=tok.pad_token_id, eos_token_id=tok.eos_token_id)
        tag = "v" if use_vision else "l"
        out[f"{tag}_inputs_embeds"] = emb.numpy()
        out[f"{tag}_ids"] = ids.numpy().astype(np.int64)
        out[f"{tag}_prompt_ids"] = np.asarray(tok.encode(prompt), dtype=np.int64)
        print(tag, "prompt:", repr(prompt), "L =", emb.shape[1], "ids", ids.tolist(), "->", repr(tok.decode(ids[0])))
        if not use_vision:
            # the engine's own generate(): it decodes `outputs[0][inputs_embeds.shape[1]:]` -- transformers returns only the
            # new tokens for an inputs_embeds-only call, so that slice is EMPTY for max_new_tokens < prompt length
            ans = eng.generate(QUESTION, bev, max_new_tokens=N_NEW, do_sample=False)
            out["l_reference_answer_is_empty"] = np.int64(ans == "")
            print("reference generate() ->", repr(ans))
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "engine.npz"), **out)


if __name__ == "__main__":
    main()
