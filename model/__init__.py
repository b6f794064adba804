"""Alias package: the reference's import lines (`from model.walkgpt import walkgptForCausalLM`, `from model.segment_anything import
build_sam_vit_h`, train_walkgpt.py:19, evaluation_walkgpt.py:18) resolve to the MI355X-native modules of walkgpt_amd.  Nothing lives
here; see INTEGRATION.md."""
