"""`model.segment_anything` of the reference (/root/reference/model/segment_anything/__init__.py:7-15, build_sam.py) -> walkgpt_amd."""
from walkgpt_amd.segment_anything.modeling import (  # noqa: F401
    ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer, build_sam, build_sam_vit_b, build_sam_vit_h, build_sam_vit_l,
    sam_model_registry)
