"""shim: llava/mm_utils.py -> spatialrgpt_amd.mm_utils"""
from spatialrgpt_amd.mm_utils import (KeywordsStoppingCriteria, get_model_name_from_path, process_image,  # noqa: F401
                                      process_images, process_regions, tokenizer_image_token)
