"""HIP-backed SAM modules with the reference's public names (model/segment_anything/__init__.py, build_sam.py)."""
from .modeling import (ImageEncoderViT, MaskDecoder, PromptEncoder, Sam, TwoWayTransformer, build_sam,  # noqa: F401
                       build_sam_vit_b, build_sam_vit_h, build_sam_vit_l, sam_model_registry)
