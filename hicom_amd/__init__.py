"""hicom_amd -- MI355X-native (gfx950) implementation of HICom's hybrid-level, instruction-injected
video-token compressor behind the reference's projector API."""
from .projector import (GlobalCompressor, GuideInjector, HIComProjector, IdentityMap, LocalCompressor,  # noqa: F401
                        MultiheadAttention, build_mlp, build_vision_projector)
from .mm_utils import post_process_visual_feature  # noqa: F401
from .encoder import siglip_head_embed, siglip_head_scores  # noqa: F401
from .splice import prepare_inputs_labels_for_multimodal  # noqa: F401
from .native import invalidate_weight_caches  # noqa: F401

__all__ = ["build_vision_projector", "HIComProjector", "LocalCompressor", "GlobalCompressor", "GuideInjector",
           "MultiheadAttention", "IdentityMap", "build_mlp", "post_process_visual_feature", "siglip_head_embed", "siglip_head_scores",
           "prepare_inputs_labels_for_multimodal", "invalidate_weight_caches"]
