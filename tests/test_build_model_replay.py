"""Construction-time half of the top-level surface: the reference's `build_model` (evaluation_walkgpt.py:202-335) replayed call by
call on the adapter -- from_pretrained with the 20 keyword arguments, `.config.*` writes, initialize_vision_modules,
get_vision_tower, initialize_walkgpt_modules, resize_token_embeddings, load_state_dict(strict=True) with the reference's key layout,
.to(), .eval().  CPU only: nothing here runs a HIP op."""
import os
import tempfile
from types import SimpleNamespace

import pytest
import torch

SAM = dict(embed_dim=64, depth=2, heads=2, global_idx=[1], img=448)
CLIP = dict(hidden_size=128, intermediate_size=512, num_hidden_layers=3, num_attention_heads=2, image_size=336, patch_size=14,
            layer_norm_eps=1e-5)


def _llama_config(**extra):
    from transformers import LlamaConfig
    return LlamaConfig(vocab_size=96, hidden_size=64, intermediate_size=128, num_hidden_layers=2, num_attention_heads=4,
                       num_key_value_heads=4, max_position_embeddings=1024, **extra)


def _model_args(**over):
    """evaluation_walkgpt.py:204-225, values as the reference's argparse defaults would set them."""
    a = dict(train_mask_decoder=True, out_dim=256, ce_loss_weight=1.0, dice_loss_weight=0.5, bce_loss_weight=2.0, seg_token_idx=90,
             vision_pretrained=None, vision_tower="openai/clip-vit-large-patch14-336", use_mm_start_end=False, seg_token_num=1,
             logger=None, tokenizer=None, local_rank=0, pad_train_clip_images=False, resize_vision_tower=False,
             resize_vision_tower_size=224, vision_tower_for_mask=True, separate_mm_projector=False, masks_process_with_clip=False,
             image_feature_scale_num=3)
    a.update(over)
    return a


def test_build_model_sequence_replays_on_the_adapter():
    from model.walkgpt import walkgptForCausalLM
    tok = SimpleNamespace(eos_token_id=2, bos_token_id=1, pad_token_id=0)
    # evaluation_walkgpt.py:233-238
    model = walkgptForCausalLM.from_pretrained(_llama_config(mm_hidden_size=128), torch_dtype=torch.bfloat16, low_cpu_mem_usage=True,
                                               sam=SAM, clip_config=CLIP, **_model_args())
    model.config.eos_token_id = tok.eos_token_id
    model.config.bos_token_id = tok.bos_token_id
    model.config.pad_token_id = tok.pad_token_id
    # the overrides of walkgpt.py:174-181 win over what the caller passed
    cfg = model.get_model().config
    assert cfg is model.config
    assert (cfg.resize_vision_tower, cfg.resize_vision_tower_size, cfg.pad_train_clip_images) == (True, 448, True)
    assert cfg.vision_tower_for_mask is False and cfg.separate_mm_projector is True and cfg.image_feature_scale_num == 1
    assert model.image_feature_scale_num == 1 and model.seg_token_idx == 90
    assert cfg.use_cache is False and cfg.mm_vision_select_feature == "patch" and cfg.vision_tower == cfg.mm_vision_tower
    # :244-248
    model.get_model().initialize_vision_modules(model.get_model().config)
    vision_tower = model.get_model().get_vision_tower()
    assert vision_tower is model.get_vision_tower() and vision_tower.is_loaded
    vision_tower.to(dtype=torch.bfloat16, device=torch.device("cpu"))
    old_sam = model.get_model().visual_model
    model.get_model().initialize_walkgpt_modules(model.get_model().config)
    assert model.get_model().visual_model is not old_sam                          # rebuilt, as the reference does
    assert cfg.mm_hidden_size == 128 and cfg.use_mm_proj is True
    # the tower runs at 448 px with the position table created at that size (clip_encoder.py:38-55)
    assert vision_tower.vision_tower.vision_model.embeddings.position_embedding.weight.shape == (32 * 32 + 1, 128)
    for p in vision_tower.parameters():
        p.requires_grad = False
    # requires_grad pattern of walkgpt.py:129-146 with train_mask_decoder=True
    vm = model.get_model().visual_model
    assert not any(p.requires_grad for p in vm.image_encoder.parameters())
    assert all(p.requires_grad for p in vm.mask_decoder.parameters())
    assert all(p.requires_grad for p in model.get_model().text_hidden_fcs.parameters())
    # :297
    model.resize_token_embeddings(100)
    assert model.get_input_embeddings().weight.shape == (100, 64) and model.config.vocab_size == 100
    # :299-310: a checkpoint in the reference's key layout loads strict=True
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    heads = {k.split(".")[1] for k in sd if k.startswith("model.")}
    assert {"embed_tokens", "layers", "norm", "visual_model", "out_mm_projector", "text_hidden_fcs", "tiny_xattn", "vision_tower",
            "mm_projector"} <= heads and "lm_head.weight" in sd and not any(k.startswith("llm.") for k in sd)
    assert sd["model.mm_projector.0.weight"].shape == (128, 128) and sd["model.mm_projector.2.weight"].shape == (64, 128)
    sd["model.text_hidden_fcs.0.log_temp"] = torch.full_like(sd["model.text_hidden_fcs.0.log_temp"], 0.25)
    sd["model.layers.1.mlp.up_proj.weight"] = torch.full_like(sd["model.layers.1.mlp.up_proj.weight"], 0.5)
    missing, unexpected = model.load_state_dict(sd, strict=True)
    assert not missing and not unexpected
    assert model.get_model().text_hidden_fcs[0].log_temp.item() == 0.25
    assert model.llm.model.layers[1].mlp.up_proj.weight[0, 0].item() == 0.5
    # :330-335
    model.to(device=torch.device("cpu"), dtype=torch.bfloat16)
    model.eval()
    assert not model.training and next(model.parameters()).dtype == torch.bfloat16
    # the projector-only checkpoint of utils_walkgpt.py:360-371 / evaluation_walkgpt.py:312-328
    proj = model.get_model().out_mm_projector
    proj.load_state_dict({k: v for k, v in proj.state_dict().items()}, strict=True)


def test_from_pretrained_sources():
    from model.walkgpt import walkgptForCausalLM
    with pytest.raises(RuntimeError, match="Hub downloads"):
        walkgptForCausalLM.from_pretrained("liuhaotian/llava-llama-2-13b-chat-lightning-preview", **_model_args(), sam=SAM)
    with pytest.raises(KeyError):                                                # walkgpt.py:207 pops seg_token_idx without a default
        a = _model_args()
        a.pop("seg_token_idx")
        walkgptForCausalLM.from_pretrained(_llama_config(), sam=SAM, **a)
    # a local checkpoint directory: config.json (model_type "llava") + language-model weights + grounding weights side by side
    from safetensors.torch import save_file
    src = walkgptForCausalLM.from_pretrained(_llama_config(mm_hidden_size=128), sam=SAM, **_model_args())
    with torch.no_grad():
        src.get_model().text_hidden_fcs[0].log_temp.fill_(0.75)
        src.llm.model.norm.weight.fill_(0.125)
    with tempfile.TemporaryDirectory() as d:
        src.llm.save_pretrained(d, safe_serialization=True)
        import json
        cfgp = os.path.join(d, "config.json")
        c = json.load(open(cfgp))
        c["model_type"] = "llava"
        c["architectures"] = ["LlavaLlamaForCausalLM"]
        json.dump(c, open(cfgp, "w"))
        g = {"model." + k: v.contiguous() for k, v in src.get_model().state_dict().items() if not k.startswith("vision_tower.")}
        save_file(g, os.path.join(d, "grounding.safetensors"))
        got = walkgptForCausalLM.from_pretrained(d, torch_dtype=torch.float32, low_cpu_mem_usage=True, sam=SAM, **_model_args())
    assert got.get_model().text_hidden_fcs[0].log_temp.item() == 0.75
    assert got.llm.model.norm.weight[0].item() == 0.125
    assert "text_hidden_fcs.0.log_temp" in got.loaded_grounding_keys and got.config.mm_hidden_size == 128


def test_fsdp_form_keeps_the_tower_out_of_the_module_tree():
    """llava_arch.py:60-63: with fsdp the tower is kept in a list (not registered); get_vision_tower unwraps it (:43-47)."""
    from walkgpt_amd.walkgpt import WalkGPTGrounding
    g = WalkGPTGrounding(sam=SAM, llm_hidden=64, with_clip=False)
    args = SimpleNamespace(vision_tower="openai/clip-vit-large-patch14-336", mm_vision_select_layer=-2, clip_config=CLIP,
                           resize_vision_tower=True, resize_vision_tower_size=448, pad_train_clip_images=True, pretrain_mm_mlp_adapter=None)
    g.initialize_vision_modules(args, fsdp=["full_shard"])
    assert type(g.vision_tower) is list and g.get_vision_tower() is g.vision_tower[0]
    assert not any(k.startswith("vision_tower.") for k in g.state_dict())
    assert isinstance(g.mm_projector, torch.nn.Linear) and g.mm_projector.weight.shape == (64, 128)   # created only if absent (:70-73)
    with pytest.raises(ValueError, match="Unknown vision tower"):
        g.initialize_vision_modules(SimpleNamespace(vision_tower="facebook/dinov2"))
