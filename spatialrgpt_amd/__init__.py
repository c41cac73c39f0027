"""spatialrgpt_amd -- MI355X-native implementation of SpatialRGPT's region-grounded multimodal
forward/generate path.  The compute lives in libsrgpt_hip.so (hand-written HIP for gfx950);
this package is the Python host mirroring the reference's call surface."""
from .config import SrgptConfig
from .constants import (DEFAULT_DEPTH_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_MASK_TOKEN, IGNORE_INDEX,
                        IMAGE_TOKEN_INDEX)
from .mm_utils import (KeywordsStoppingCriteria, get_model_name_from_path, process_images, process_images_device,
                       process_regions, process_regions_device,
                       tokenizer_image_token)

__all__ = ["SrgptConfig", "IMAGE_TOKEN_INDEX", "IGNORE_INDEX", "DEFAULT_IMAGE_TOKEN", "DEFAULT_MASK_TOKEN",
           "DEFAULT_DEPTH_TOKEN", "tokenizer_image_token", "KeywordsStoppingCriteria", "process_images",
           "process_regions", "process_images_device", "process_regions_device", "get_model_name_from_path", "LlavaLlamaModel", "LlavaLlamaForCausalLM",
           "LlavaLlamaConfig", "load_pretrained_model", "SrgptEngine"]


def __getattr__(name):  # lazy: importing the package must not require the built extension / a GPU
    if name in ("LlavaLlamaModel", "LlavaLlamaForCausalLM", "LlavaLlamaConfig"):
        from . import model
        return getattr(model, name)
    if name == "load_pretrained_model":
        from .builder import load_pretrained_model
        return load_pretrained_model
    if name == "SrgptEngine":
        from .engine import SrgptEngine
        return SrgptEngine
    raise AttributeError(name)
