"""shim: llava/constants.py -> spatialrgpt_amd.constants (+ the two server constants and LOGDIR that llava.utils reads)"""
from spatialrgpt_amd.constants import *  # noqa: F401,F403
from spatialrgpt_amd.constants import (DEFAULT_DEPTH_TOKEN, DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN,  # noqa: F401
                                       DEFAULT_IMAGE_PATCH_TOKEN, DEFAULT_IMAGE_TOKEN, DEFAULT_MASK_TOKEN, IGNORE_INDEX,
                                       IMAGE_PLACEHOLDER, IMAGE_TOKEN_INDEX)

CONTROLLER_HEART_BEAT_EXPIRATION = 30
WORKER_HEART_BEAT_INTERVAL = 15
LOGDIR = "."
