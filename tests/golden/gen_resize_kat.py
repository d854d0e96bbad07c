#!/usr/bin/env python3
"""Known-answer vectors for ``skimage.transform.resize`` as utils.py:318-339,389-408 call it
-> tests/golden/resize_kat.json.   python tests/golden/gen_resize_kat.py        (no scipy, no skimage, no torch: exact rationals)

scikit-image is not in this image, so no golden can be generated from the reference's own call.  These vectors are DERIVED BY
HAND from the algorithm skimage publishes (skimage/transform/_warps.py, resize(), >= 0.19) and evaluated here in exact rational
arithmetic (fractions.Fraction) -- independent of scipy, of the oracle's restatement (oracle.skimage_resize, which calls
scipy.ndimage.zoom) and of the device kernel (cfun_resize3d); tests/test_oracle_golden.py holds BOTH to them.

The reference's calls:  resize(image, out_shape, order = 1 | 0, mode = 'constant', cval = 0, clip = True, preserve_range = True,
anti_aliasing = False).  What _warps.resize does with that for an array of more than two dimensions whose scale is not 1:

  1. no anti-aliasing filter (anti_aliasing = False);
  2. out = scipy.ndimage.zoom(image, out_shape / in_shape, order = order, mode = 'grid-constant' (skimage's name 'constant',
     _to_ndimage_mode / _fix_ndimage_mode), cval = 0, grid_mode = True).  With grid_mode = True pixel CENTRES map onto pixel
     centres of the rescaled grid: output index o along an axis of n_in -> n_out samples reads the input at the coordinate
                  c(o) = (o + 1/2) * n_in / n_out - 1/2 ;
     order 1: linear interpolation between floor(c) and floor(c) + 1 with weights (1 - t, t), t = c - floor(c); a sample index
              outside [0, n_in) reads the constant cval = 0 ('grid-constant': the padding takes part in the interpolation, so
              the outermost outputs of an up-sampled axis fade towards 0); separable over the axes;
     order 0: the nearest sample, index floor(c + 1/2) = floor((o + 1/2) * n_in / n_out) (always inside the array);
  3. clip = True (_clip_warp_output): out is clipped to [min(image), max(image)]; elements EQUAL to cval keep it when cval lies
     outside that range (never the case below);
  4. preserve_range = True: values are not rescaled.

Every order-0 case below avoids INEXACT ties -- (o + 1/2) * n_in / n_out an integer while n_in / n_out is not a binary fraction
(e.g. 2 -> 3 at o = 1, 4 -> 6 at o = 1) -- where the double-precision product may land on either side of the integer (an
implementation detail of scipy, not of the algorithm): odd -> odd sizes never tie ((2o + 1) n_in is odd), ratios 2 and 1/2 are
exact in binary."""
import json
import os
from fractions import Fraction as Fr
from itertools import product
from math import floor


def coord(o, n_in, n_out):
    return (Fr(2 * o + 1, 2)) * Fr(n_in, n_out) - Fr(1, 2)


def axis_weights(o, n_in, n_out, order):
    """[(input index, exact weight)] of output o along one axis; indices outside [0, n_in) are the zero padding (dropped)."""
    c = coord(o, n_in, n_out)
    if order == 0:
        return [(floor(c + Fr(1, 2)), Fr(1))]
    f = floor(c)
    t = c - f
    return [(i, w) for i, w in ((f, 1 - t), (f + 1, t)) if 0 <= i < n_in and w != 0]


def resize(image, shape_in, shape_out, order, clip):
    """image: nested-list values addressed image[i][j][k] (ints); returns a flat list of exact Fractions, C order."""
    lo = min(v for a in image for b in a for v in b)
    hi = max(v for a in image for b in a for v in b)
    out = []
    for o in product(*[range(m) for m in shape_out]):
        ws = [axis_weights(o[d], shape_in[d], shape_out[d], order) for d in range(3)]
        v = Fr(0)
        for (i, wi), (j, wj), (k, wk) in product(*ws):
            v += wi * wj * wk * image[i][j][k]
        if clip and order:
            v = min(max(v, Fr(lo)), Fr(hi))
        out.append(v)
    return out


def arange_image(shape, base, step_=1, sign_every=0):
    vals, n = [], base
    img = [[[0] * shape[2] for _ in range(shape[1])] for _ in range(shape[0])]
    for i, j, k in product(*[range(s) for s in shape]):
        v = n
        if sign_every and (i + j + k) % sign_every == 0:
            v = -v
        img[i][j][k] = v
        vals.append(v)
        n += step_
    return img, vals


CASES = [
    # name, in shape, out shape, order, clip, image(base, step, sign_every), what it pins
    ("up2_cube_order1", (2, 2, 2), (4, 4, 4), 1, False, (1, 1, 0),
     "x2 up-sampling: c = -1/4, 1/4, 3/4, 5/4 -> weights (1/4 pad, 3/4), (3/4, 1/4), (1/4, 3/4), (3/4, 1/4 pad): the outer "
     "samples fade towards the zero padding (grid-constant)"),
    ("up2_cube_order1_clip", (2, 2, 2), (4, 4, 4), 1, True, (100, 7, 0),
     "the same with an all-positive image and clip = True: the faded outer samples are clipped UP to min(image) = 100"),
    ("down_mixed_order1", (5, 4, 3), (2, 3, 3), 1, True, (-20, 3, 0),
     "down-sampling 5 -> 2 (c = 3/4, 13/4) and 4 -> 3 (c = 1/6, 3/2, 17/6), identity axis 3 -> 3 (c = o)"),
    ("size1_axes_order1", (1, 3, 1), (3, 3, 2), 1, False, (10, 5, 0),
     "size-1 axes up-sampled: 1 -> 3 reads c = -1/3, 0, 1/3 (weights 2/3, 1, 2/3 on the single sample), 1 -> 2 reads "
     "c = -1/4, 1/4 (3/4, 3/4)"),
    ("nonint_up_order1", (3, 2, 4), (7, 5, 6), 1, True, (-9, 4, 3),
     "non-integer up-sampling 3 -> 7, 2 -> 5, 4 -> 6 with mixed signs (the clip range spans 0)"),
    ("labels_up_order0", (3, 5, 4), (7, 9, 8), 0, False, (0, 1, 0),
     "order 0 (resize_mask, utils.py:404): index floor((o + 1/2) * n_in / n_out); odd -> odd sizes (3 -> 7, 5 -> 9) cannot "
     "tie, 4 -> 8 has the binary ratio 1/2"),
    ("labels_down_order0", (7, 5, 8), (3, 3, 4), 0, False, (0, 1, 0),
     "order 0 down-sampling 7 -> 3, 5 -> 3, 8 -> 4 (8 -> 4: c + 1/2 = 2o + 1 exactly, the SECOND sample of each pair)"),
]


def main():
    out = {"derivation": __doc__, "cases": []}
    for name, sin, sout, order, clip, (base, step_, sign), what in CASES:
        img, vals = arange_image(sin, base, step_, sign)
        res = resize(img, sin, sout, order, clip)
        out["cases"].append({
            "name": name, "what": what, "in_shape": list(sin), "out_shape": list(sout), "order": order, "clip": clip,
            "image": vals, "expected_fraction": ["%d/%d" % (v.numerator, v.denominator) for v in res],
            "expected": [float(v) for v in res]})
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "resize_kat.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", path, sum(len(c["expected"]) for c in out["cases"]), "values")


# self-check of the tie rule at generation time
for _name, _sin, _sout, _order, *_ in CASES:
    if _order == 0:
        for _a, _b in zip(_sin, _sout):
            for _o in range(_b):
                _v = Fr(2 * _o + 1, 2) * Fr(_a, _b)
                assert _v.denominator != 1 or Fr(_a, _b).denominator in (1, 2), (_name, _a, _b, _o)


if __name__ == "__main__":
    main()
