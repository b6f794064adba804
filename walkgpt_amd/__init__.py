"""walkgpt_amd: MI355X-native (gfx950) grounded-segmentation forward path of WalkGPT.

  csrc/                 hand-written HIP kernels + the C-ABI (libwalkgpt_hip.so, declared in include/walkgpt_hip.h)
  ops.py                tensor-level wrappers over the C-ABI (torch = HBM buffers and streams only)
  segment_anything/     SAM image encoder / prompt encoder / mask decoder behind the reference's module surface
  clip_encoder.py       CLIP ViT-L/14 tower behind the reference's CLIPVisionTower surface
  utils_walkgpt.py      MSQP and CTP behind the reference's class names
  walkgpt.py            the evaluate()-style composition
  causal_lm.py          walkgptForCausalLM: the reference's top-level surface around an injected / built language model
  autograd.py           differentiable forms of the head's operators (HIP forward + HIP backward in torch.autograd.Function)
  train_head.py         the trainable grounding head (CTP, mask decoder, postprocess, mask losses) composed from them
  synth.py              deterministic synthetic weights / inputs (tests, golden vectors, bench)
There is no CPU path in this package; the CPU oracle lives in /oracle and is test infrastructure only.
"""
__version__ = "0.1.0"
