"""CPU: `PYTHONPATH=compat` resolves the import lines of the reference's callers (eval_spatial.py:17-21, eval_region_cls.py:16-20,
model_vqa.py:13-17, demo/gradio_web_server_multi.py:23-26) to this package, with the callers byte-unchanged; with a reference
checkout named by SRGPT_REFERENCE_ROOT the out-of-scope host modules (conversation templates, llava.utils) load from it."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CALLER_IMPORTS = """
from llava.constants import DEFAULT_IM_END_TOKEN, DEFAULT_IM_START_TOKEN, DEFAULT_IMAGE_TOKEN, IMAGE_TOKEN_INDEX
from llava.mm_utils import KeywordsStoppingCriteria, get_model_name_from_path, process_images, process_regions, tokenizer_image_token
from llava.model.builder import load_pretrained_model
import spatialrgpt_amd, spatialrgpt_amd.builder, spatialrgpt_amd.mm_utils
assert load_pretrained_model is spatialrgpt_amd.builder.load_pretrained_model
assert tokenizer_image_token is spatialrgpt_amd.mm_utils.tokenizer_image_token
assert process_regions is spatialrgpt_amd.mm_utils.process_regions and KeywordsStoppingCriteria is spatialrgpt_amd.KeywordsStoppingCriteria
assert IMAGE_TOKEN_INDEX == -200 and DEFAULT_IMAGE_TOKEN == "<image>"
from llava.model import LlavaLlamaModel
assert LlavaLlamaModel is spatialrgpt_amd.LlavaLlamaModel
"""


def _run(code, extra_env=None):
    env = dict(os.environ)
    env["PYTHONPATH"] = os.path.join(ROOT, "compat") + os.pathsep + ROOT
    env.pop("SRGPT_REFERENCE_ROOT", None)
    env.update(extra_env or {})
    return subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)


def test_caller_import_lines_resolve_to_this_package():
    r = _run(CALLER_IMPORTS + "print('SHIM_OK')")
    assert r.returncode == 0 and "SHIM_OK" in r.stdout, r.stderr[-2000:]


def test_out_of_scope_host_modules_load_from_a_reference_checkout():
    ref = "/root/reference"
    if not os.path.isdir(os.path.join(ref, "llava")):  # the build container has it, the GPU box does not
        import pytest

        pytest.skip("no reference checkout on this machine")
    code = CALLER_IMPORTS + """
from llava.conversation import SeparatorStyle, conv_templates
from llava.utils import disable_torch_init
import llava.conversation, llava.model.builder
assert llava.conversation.__file__.startswith(%r) and "compat" in llava.model.builder.__file__
assert "llama_3" in conv_templates or len(conv_templates) > 3
print('SHIM_OK')
""" % ref
    r = _run(code, {"SRGPT_REFERENCE_ROOT": ref})
    assert r.returncode == 0 and "SHIM_OK" in r.stdout, r.stderr[-2000:]
