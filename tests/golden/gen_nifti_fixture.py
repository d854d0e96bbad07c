#!/usr/bin/env python3
"""Byte-level NIfTI-1 fixtures written FROM THE FORMAT'S SPECIFICATION (nifti1.h, the NIfTI-1.1 data format standard), not
through cfun_amd.nifti and not through nibabel (absent from this image)  -> tests/golden/nifti1_sform.nii, nifti1_qform_be.nii
(until round 4 the reader was only checked against files its own writer produced).

    python tests/golden/gen_nifti_fixture.py

The 348-byte header, fields this path reads (byte offset, C type, name -- nifti1.h `struct nifti_1_header`):
      0  int    sizeof_hdr   = 348                       40  short  dim[8]       dim[0] = rank, dim[1..] extents
     70  short  datatype     (4 = int16, 16 = float32)   72  short  bitpix
     76  float  pixdim[8]    pixdim[0] = qfac (-1 | 1)  108  float  vox_offset   = 352 for a .nii without extensions
    112  float  scl_slope   116  float  scl_inter        value = stored * scl_slope + scl_inter  (slope 0: unscaled)
    252  short  qform_code  254  short  sform_code
    256  float  quatern_b, _c, _d   268  float qoffset_x, _y, _z
    280  float  srow_x[4]   296  float  srow_y[4]   312  float srow_z[4]
    344  char   magic[4]     = "n+1\\0" (single file)
followed by 4 extension-flag bytes (0) and the voxels, FIRST index fastest (x, then y, then z).

Orientation (the standard's "METHOD 3" / "METHOD 2"): sform_code > 0 -> the affine's rows are srow_x / _y / _z; else
qform_code > 0 -> R(quaternion a, b, c, d with a = sqrt(1 - b^2 - c^2 - d^2)) * diag(pixdim[1], pixdim[2], qfac * pixdim[3]) and
the offsets.  The expected arrays / affines are typed out in tests/test_nifti.py, by hand from these rules."""
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))


def header(end, dim, datatype, bitpix, pixdim, slope, inter, qcode, scode, quat, qoff, srow):
    h = bytearray(348)
    struct.pack_into(end + "i", h, 0, 348)
    struct.pack_into(end + "8h", h, 40, *dim)
    struct.pack_into(end + "2h", h, 70, datatype, bitpix)
    struct.pack_into(end + "8f", h, 76, *pixdim)
    struct.pack_into(end + "3f", h, 108, 352.0, slope, inter)
    struct.pack_into(end + "2h", h, 252, qcode, scode)
    struct.pack_into(end + "6f", h, 256, *quat, *qoff)
    struct.pack_into(end + "12f", h, 280, *srow)
    h[344:348] = b"n+1\0"
    return bytes(h) + b"\0\0\0\0"


def main():
    # (1) little-endian int16 [3, 2, 2], scl_slope 2 / scl_inter -5, sform_code 1 with an oblique affine; the qform fields hold a
    #     DIFFERENT orientation with qform_code 1 as well: the sform must win (nibabel's get_best_affine order)
    vox = list(range(12))                                     # stored value = x + 3 * y + 6 * z
    srow = [2.0, 0.0, 0.5, -10.0,   0.0, 1.5, 0.0, 20.0,   -0.25, 0.0, 3.0, 30.0]
    with open(os.path.join(HERE, "nifti1_sform.nii"), "wb") as f:
        f.write(header("<", [3, 3, 2, 2, 1, 1, 1, 1], 4, 16, [1.0, 2.0, 1.5, 3.0, 1.0, 1.0, 1.0, 1.0], 2.0, -5.0, 1, 1,
                       [0.0, 0.0, 0.0], [1.0, 2.0, 3.0], srow))
        f.write(struct.pack("<12h", *vox))
    # (2) BIG-endian float32 [2, 3, 1, 2] (a 4-D file), unscaled (slope 0), qform only: quaternion (b, c, d) = (0, 0, sqrt(1/2)) is
    #     the rotation by 90 degrees about z -- a = sqrt(1/2); R = [[0,-1,0],[1,0,0],[0,0,1]] -- pixdim (2, 3, 4), qfac = -1
    vals = [0.5 * k - 1.0 for k in range(12)]
    q = 0.5 ** 0.5
    with open(os.path.join(HERE, "nifti1_qform_be.nii"), "wb") as f:
        f.write(header(">", [4, 2, 3, 1, 2, 1, 1, 1], 16, 32, [-1.0, 2.0, 3.0, 4.0, 1.0, 1.0, 1.0, 1.0], 0.0, 0.0, 1, 0,
                       [0.0, 0.0, q], [7.0, -8.0, 9.0], [0.0] * 12))
        f.write(struct.pack(">12f", *vals))


if __name__ == "__main__":
    main()
