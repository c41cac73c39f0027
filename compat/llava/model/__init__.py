"""shim: llava/model -> spatialrgpt_amd.model (llava/model/__init__.py:1 exports LlavaLlamaConfig / LlavaLlamaModel)"""
from spatialrgpt_amd.model import LlavaLlamaConfig, LlavaLlamaForCausalLM, LlavaLlamaModel  # noqa: F401
