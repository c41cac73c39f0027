"""shim: llava/model/builder.py -> spatialrgpt_amd.builder"""
from spatialrgpt_amd.builder import load_pretrained_model  # noqa: F401
