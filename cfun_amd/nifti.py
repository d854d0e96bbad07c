"""Dependency-free NIfTI-1 reader / writer for the calls the reference makes through nibabel:

    nib.load(path).get_data()            heart_main.py:211,223,252,300   utils.py:307
    nib.load(path).affine                heart_main.py:301-302 (label.affine)
    nib.Nifti1Image(array, affine)       heart_main.py:349
    nib.save(img, path)                  heart_main.py:352

Single-file ``.nii`` / ``.nii.gz`` (magic ``n+1``), either byte order, the scalar data types the medical-segmentation
benchmarks ship (uint8 ... float64), ``scl_slope`` / ``scl_inter`` applied as nibabel's ``get_data`` does.  Arrays come
back in the file's own index order -- [X, Y, Z] = the reference's [H, W, D] (it transposes / resizes from there,
utils.py:389-408 -> ``cfun_amd.utils.resize_image``).  The affine is the sform (``srow_*``) when ``sform_code`` > 0,
else the qform (quaternion + offsets + pixdim), else the pixdim scaling -- nibabel's ``get_best_affine`` order.
NIfTI-2, the two-file .hdr / .img form and header extensions beyond skipping them are not needed by the reference.
"""
import gzip
import struct

import numpy as np

_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16,
           768: np.uint32, 1024: np.int64, 1280: np.uint64}
_CODES = {np.dtype(v).name: k for k, v in _DTYPES.items()}


class Nifti1Image:
    """The three things the reference touches: ``get_data()`` / ``get_fdata()``, ``affine``, ``header`` (a dict)."""

    def __init__(self, dataobj, affine=None, header=None):
        self._data = np.asarray(dataobj)
        self.affine = np.eye(4) if affine is None else np.asarray(affine, dtype=np.float64).reshape(4, 4)
        self.header = dict(header or {})

    @property
    def shape(self):
        return self._data.shape

    def get_data(self):
        return self._data

    def get_fdata(self):
        return self._data.astype(np.float64)


def _open(path, mode):
    return gzip.open(path, mode) if str(path).endswith(".gz") else open(path, mode)


def _qform_affine(h):
    b, c, d = h["quatern_b"], h["quatern_c"], h["quatern_d"]
    a = np.sqrt(max(0.0, 1.0 - (b * b + c * c + d * d)))
    r = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    qfac = -1.0 if h["pixdim"][0] < 0 else 1.0
    s = np.array([h["pixdim"][1], h["pixdim"][2], h["pixdim"][3] * qfac])
    aff = np.eye(4)
    aff[:3, :3] = r * s
    aff[:3, 3] = [h["qoffset_x"], h["qoffset_y"], h["qoffset_z"]]
    return aff


def load(path):
    """-> Nifti1Image (data in file index order [X, Y, Z(, T...)], scaled by scl_slope / scl_inter when set)."""
    with _open(path, "rb") as f:
        raw = f.read(352)
        if len(raw) >= 348:
            return _load_body(path, f, raw)
    raise ValueError("%s: shorter than a NIfTI-1 header" % path)


def _load_body(path, f, raw):
    end = "<" if struct.unpack("<i", raw[:4])[0] == 348 else ">"
    if struct.unpack(end + "i", raw[:4])[0] != 348:
        raise ValueError("%s: sizeof_hdr is not 348 (NIfTI-2 / not NIfTI)" % path)
    magic = raw[344:348]
    if magic[:3] not in (b"n+1", b"ni1"):
        raise ValueError("%s: bad magic %r" % (path, magic))
    if magic[:3] == b"ni1":
        raise ValueError("%s: two-file NIfTI (.hdr/.img) is not supported" % path)
    dim = struct.unpack(end + "8h", raw[40:56])
    datatype, bitpix = struct.unpack(end + "2h", raw[70:74])
    pixdim = struct.unpack(end + "8f", raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + "3f", raw[108:120])
    qform_code, sform_code = struct.unpack(end + "2h", raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + "6f", raw[256:280])
    srow = np.array(struct.unpack(end + "12f", raw[280:328]), dtype=np.float64).reshape(3, 4)
    if datatype not in _DTYPES:
        raise ValueError("%s: unsupported NIfTI datatype code %d" % (path, datatype))
    nd = dim[0]
    if not 1 <= nd <= 7:
        raise ValueError("%s: dim[0] = %d" % (path, nd))
    shape = tuple(int(v) for v in dim[1:1 + nd])
    dt = np.dtype(_DTYPES[datatype]).newbyteorder(end)
    off = int(vox_offset) if vox_offset >= 352 else 352
    n = int(np.prod(shape))
    # the voxels are read straight into the array that is returned (one buffer, writable like nibabel's get_data());
    # only a foreign byte order costs a second pass
    skip = off - len(raw)
    while skip > 0:
        got = f.read(min(skip, 1 << 20))
        if not got:
            break
        skip -= len(got)
    data = np.empty(n, dtype=dt)
    view = memoryview(data).cast("B")
    filled = 0
    while filled < len(view):
        got = f.readinto(view[filled:])
        if not got:
            raise ValueError("%s: %d bytes of voxel data, header promises %d" % (path, filled, len(view)))
        filled += got
    data = data.reshape(shape, order="F")
    if not dt.isnative:
        data = data.astype(dt.newbyteorder("="))
    if slope not in (0.0, 1.0) or (slope != 0.0 and inter != 0.0):
        if np.isfinite(slope) and np.isfinite(inter) and slope != 0.0:
            data = data.astype(np.float64) * slope + inter
    hdr = dict(dim=dim, datatype=datatype, bitpix=bitpix, pixdim=pixdim, vox_offset=vox_offset, scl_slope=slope,
               scl_inter=inter, qform_code=qform_code, sform_code=sform_code, quatern_b=qb, quatern_c=qc, quatern_d=qd,
               qoffset_x=qx, qoffset_y=qy, qoffset_z=qz, endianness=end)
    if sform_code > 0:
        aff = np.vstack([srow, [0.0, 0.0, 0.0, 1.0]])
    elif qform_code > 0:
        aff = _qform_affine(hdr)
    else:
        aff = np.diag([pixdim[1], pixdim[2], pixdim[3], 1.0]).astype(np.float64)
    return Nifti1Image(data, aff, hdr)


def _affine_to_quaternion(aff):
    """(quatern_b, c, d, qfac, zooms) of the rotation closest to the affine's 3 x 3 part -- what nibabel's ``set_qform`` stores:
    column norms as zooms, a negative determinant folded into qfac (third column flipped), the polar factor of the rest
    (SVD) as the rotation, and its unit quaternion with a >= 0 from the dominant eigenvector of the symmetric 4 x 4 matrix
    built from the rotation's entries (Bar-Itzhack's formulation, as nibabel.quaternions.mat2quat)."""
    rzs = aff[:3, :3]
    zooms = np.sqrt((rzs ** 2).sum(axis=0))
    zooms[zooms == 0] = 1.0
    r = rzs / zooms
    qfac = 1.0
    if np.linalg.det(r) < 0:
        qfac = -1.0
        r = r.copy()
        r[:, 2] *= -1.0
    u, _, vt = np.linalg.svd(r)
    m = u @ vt
    (xx, yx, zx), (xy, yy, zy), (xz, yz, zz) = m
    k = np.array([[xx - yy - zz, 0, 0, 0],
                  [yx + xy, yy - xx - zz, 0, 0],
                  [zx + xz, zy + yz, zz - xx - yy, 0],
                  [yz - zy, zx - xz, xy - yx, xx + yy + zz]]) / 3.0
    vals, vecs = np.linalg.eigh(k)
    x, y, z, w = vecs[:, np.argmax(vals)]
    if w < 0:
        x, y, z = -x, -y, -z
    return float(x), float(y), float(z), qfac, zooms


def save(img, path):
    """Write ``img`` (Nifti1Image, or any object with ``get_data()`` and ``affine``) as a single-file little-endian
    NIfTI-1 the way ``nib.Nifti1Image(array, affine)`` + ``nib.save`` does (heart_main.py:349-352): the affine goes into the
    sform with code 2 ('aligned'); the qform fields (quaternion of the closest rotation, offsets, qfac in pixdim[0]) are
    filled from the same affine and its code is left 0 ('unknown'), so a reader that honours the codes takes the sform and
    one that reads the quaternion regardless still finds the orientation; the pixdims are the affine's column norms."""
    data = np.asarray(img.get_data())
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype.name not in _CODES:
        raise ValueError("NIfTI-1 cannot store dtype %s" % data.dtype)
    if not 1 <= data.ndim <= 7:
        raise ValueError("NIfTI-1 stores 1 to 7 dimensions")
    aff = np.asarray(img.affine, dtype=np.float64).reshape(4, 4)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    qb, qc, qd, qfac, vox = _affine_to_quaternion(aff)
    pixdim = [qfac] + [float(v) for v in vox] + [1.0] * 4
    h = bytearray(348)
    struct.pack_into("<i", h, 0, 348)
    struct.pack_into("<8h", h, 40, *dim)
    struct.pack_into("<2h", h, 70, _CODES[data.dtype.name], data.dtype.itemsize * 8)
    struct.pack_into("<8f", h, 76, *pixdim)
    struct.pack_into("<3f", h, 108, 352.0, 1.0, 0.0)
    h[123] = 2                                  # xyzt_units: millimetres
    struct.pack_into("<2h", h, 252, 0, 2)       # qform 'unknown' (fields filled, as nibabel), sform 'aligned'
    struct.pack_into("<6f", h, 256, qb, qc, qd, *[float(v) for v in aff[:3, 3]])
    struct.pack_into("<12f", h, 280, *aff[:3].reshape(-1))
    h[344:348] = b"n+1\0"
    with _open(path, "wb") as f:
        f.write(bytes(h))
        f.write(b"\0\0\0\0")                    # no header extensions
        f.write(np.asfortranarray(data.astype(data.dtype.newbyteorder("<"))).tobytes(order="F"))
