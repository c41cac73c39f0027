"""Sentinel ids / special tokens of the token stream (reference: llava/constants.py:25-33)."""
IGNORE_INDEX = -100
IMAGE_TOKEN_INDEX = -200
DEFAULT_IMAGE_TOKEN = "<image>"
DEFAULT_IMAGE_PATCH_TOKEN = "<im_patch>"
DEFAULT_IM_START_TOKEN = "<im_start>"
DEFAULT_IM_END_TOKEN = "<im_end>"
IMAGE_PLACEHOLDER = "<image-placeholder>"
DEFAULT_MASK_TOKEN = "<mask>"
DEFAULT_DEPTH_TOKEN = "<depth>"
