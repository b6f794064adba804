"""`model.walkgpt` of the reference (/root/reference/model/walkgpt.py) -> walkgpt_amd.causal_lm."""
from walkgpt_amd.causal_lm import walkgptForCausalLM  # noqa: F401
from walkgpt_amd.walkgpt import WalkGPTGrounding  # noqa: F401
