"""IMPORT SHIM, no logic: `PYTHONPATH=compat` makes the reference's callers (llava/eval/eval_spatial.py:17-21,
eval_region_cls.py:16-20, model_vqa.py:13-17, demo/gradio_web_server_multi.py:23-26) resolve

    llava.constants   llava.mm_utils   llava.model   llava.model.builder

to the MI355X implementation (spatialrgpt_amd) with their files byte-unchanged.  Everything else those callers import from
`llava` (conversation templates, llava.utils, the eval scripts themselves) is host-side Python outside the hot path: when a
reference checkout is named by SRGPT_REFERENCE_ROOT, its `llava/` directory is appended to this package's search path and those
modules load from there, untouched."""
import os

_ref = os.environ.get("SRGPT_REFERENCE_ROOT")
if _ref and os.path.isdir(os.path.join(_ref, "llava")):
    __path__.append(os.path.join(_ref, "llava"))
