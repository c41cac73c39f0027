"""TEST INFRASTRUCTURE ONLY.  tests/golden/cv2_nearest_kat.json: source-index tables of cv2.resize(..., interpolation=INTER_NEAREST)
(the mask resize of process_regions, llava/mm_utils.py:520) for sizes at which the obvious formula is WRONG.

OpenCV is not installed in the build image and is absent from /root/reference (a pip dependency, opencv-python==4.8.0.74,
pyproject.toml:25), so its PUBLISHED algorithm is restated -- modules/imgproc/src/resize.cpp, cv::resize -> resizeNN:

    inv_scale_x = (double)dsize.width / ssize.width            // cv::resize, dsize given
    double ifx = 1. / inv_scale_x;                             // resizeNN(src, dst, fx = inv_scale_x, fy)
    for x in [0, dsize.width):  sx = cvFloor(x * ifx);  x_ofs[x] = min(sx, ssize.width - 1)      (rows likewise with ify)

Every step is one correctly rounded IEEE-754 double operation (a division, a reciprocal, a product of an exactly representable
integer with a double), then a floor.  This script derives the tables WITHOUT floating-point code of its own: each rounded step is
an exact rational (fractions.Fraction) rounded to the nearest double by float() -- Python guarantees correct rounding there -- so it
is independent of numpy and of spatialrgpt_amd.mm_utils.cv2_nearest_index, which it pins.  It also records, per size pair, where the
naive floor(x * in / out) differs: the reason the reciprocal form matters.   python oracle/make_cv2_nearest_kat.py"""
import json
import math
import os
from fractions import Fraction

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def resize_nn_index(n_in: int, n_out: int):
    fx = float(Fraction(n_out, n_in))                 # RN(out / in)
    ifx = float(Fraction(1) / Fraction(fx))           # RN(1 / fx)
    out = []
    for x in range(n_out):
        p = float(Fraction(x) * Fraction(ifx))        # RN(x * ifx)
        out.append(min(math.floor(p), n_in - 1))
    return out


def naive_index(n_in: int, n_out: int):
    return [min((x * n_in) // n_out, n_in - 1) for x in range(n_out)]   # exact floor(x * in / out)


PAIRS = [(480, 384), (640, 384), (333, 384), (500, 384), (1080, 384), (1920, 384), (427, 384), (375, 384), (72, 224), (76, 336),
         (36, 448), (54, 336), (68, 384), (200, 336), (300, 336), (97, 378), (1024, 378), (384, 384), (7, 384), (5000, 384)]

if __name__ == "__main__":
    cases = []
    for n_in, n_out in PAIRS:
        idx = resize_nn_index(n_in, n_out)
        nv = naive_index(n_in, n_out)
        cases.append({"in": n_in, "out": n_out, "index": idx,
                      "differs_from_floor_x_in_over_out_at": [x for x in range(n_out) if idx[x] != nv[x]]})
    path = os.path.join(ROOT, "tests", "golden", "cv2_nearest_kat.json")
    with open(path, "w") as f:
        json.dump({"source": "OpenCV 4.8 modules/imgproc/src/resize.cpp resizeNN, restated (cv2 not installable here)", "cases": cases}, f)
    n_diff = sum(1 for c in cases if c["differs_from_floor_x_in_over_out_at"])
    print(f"wrote {path}: {len(cases)} size pairs, {n_diff} of them differ from floor(x * in / out)")
