"""CPU tier: the dependency-free NIfTI-1 reader / writer (cfun_amd/nifti.py) -- round trips for the data types and
layouts the reference's loaders meet (heart_main.py:211-352), gzip, big-endian files, scaled data, qform-only headers."""
import struct

import numpy as np
import pytest

from cfun_amd import nifti


@pytest.mark.parametrize("dtype", ["uint8", "int16", "int32", "float32", "float64", "uint16"])
@pytest.mark.parametrize("ext", [".nii", ".nii.gz"])
def test_round_trip(tmp_path, dtype, ext):
    rng = np.random.default_rng(0)
    a = (rng.standard_normal((5, 7, 3)) * 50).astype(dtype)
    aff = np.array([[0.0, -1.5, 0.0, 10.0], [2.0, 0.0, 0.0, -4.0], [0.0, 0.0, 3.0, 7.5], [0.0, 0.0, 0.0, 1.0]])
    p = str(tmp_path / ("v" + ext))
    nifti.save(nifti.Nifti1Image(a, aff), p)
    img = nifti.load(p)
    assert img.get_data().dtype == np.dtype(dtype) and img.shape == (5, 7, 3)
    np.testing.assert_array_equal(img.get_data(), a)
    np.testing.assert_allclose(img.affine, aff, rtol=0, atol=1e-6)
    np.testing.assert_allclose(img.header["pixdim"][1:4], [2.0, 1.5, 3.0], rtol=1e-6)


def test_file_order_is_x_fastest(tmp_path):
    """NIfTI stores x fastest: element (i, j, k) of the returned [X, Y, Z] array is voxel i + X*(j + Y*k) of the file."""
    a = np.arange(2 * 3 * 4, dtype=np.int16).reshape((2, 3, 4), order="F")
    p = str(tmp_path / "o.nii")
    nifti.save(nifti.Nifti1Image(a, np.eye(4)), p)
    raw = open(p, "rb").read()
    np.testing.assert_array_equal(np.frombuffer(raw, "<i2", offset=352), np.arange(24))
    np.testing.assert_array_equal(nifti.load(p).get_data(), a)


def test_big_endian_scaled_qform(tmp_path):
    """A big-endian file with scl_slope / scl_inter and only a qform (90 degree rotation about z, qfac -1)."""
    a = np.arange(24, dtype=np.int16).reshape((2, 3, 4), order="F")
    h = bytearray(348)
    struct.pack_into(">i", h, 0, 348)
    struct.pack_into(">8h", h, 40, 3, 2, 3, 4, 1, 1, 1, 1)
    struct.pack_into(">2h", h, 70, 4, 16)
    struct.pack_into(">8f", h, 76, -1.0, 2.0, 3.0, 4.0, 1.0, 1.0, 1.0, 1.0)
    struct.pack_into(">3f", h, 108, 352.0, 0.5, 10.0)
    struct.pack_into(">2h", h, 252, 1, 0)
    s = float(np.sqrt(0.5))
    struct.pack_into(">6f", h, 256, 0.0, 0.0, s, 1.0, 2.0, 3.0)     # quaternion (a = s, d = s): +90 degrees about z
    h[344:348] = b"n+1\0"
    p = str(tmp_path / "be.nii")
    with open(p, "wb") as f:
        f.write(bytes(h) + b"\0\0\0\0" + a.astype(">i2").tobytes(order="F"))
    img = nifti.load(p)
    np.testing.assert_allclose(img.get_data(), a * 0.5 + 10.0)
    want = np.array([[0.0, -3.0, 0.0, 1.0], [2.0, 0.0, 0.0, 2.0], [0.0, 0.0, -4.0, 3.0], [0.0, 0.0, 0.0, 1.0]])
    np.testing.assert_allclose(img.affine, want, atol=1e-6)


def test_rejects_what_it_does_not_read(tmp_path):
    p = str(tmp_path / "bad.nii")
    open(p, "wb").write(b"\0" * 400)
    with pytest.raises(ValueError):
        nifti.load(p)
    with pytest.raises(ValueError):
        nifti.save(nifti.Nifti1Image(np.zeros((2, 2), np.complex64)), p)


def test_reference_call_pattern(tmp_path):
    """heart_main.py:300-302,349-352: load image + label, take the label's affine, save an int32 mask with it."""
    rng = np.random.default_rng(1)
    img = rng.standard_normal((8, 8, 4)).astype(np.float32)
    lab = rng.integers(0, 8, (8, 8, 4)).astype(np.uint8)
    aff = np.diag([0.8, 0.8, 1.2, 1.0])
    nifti.save(nifti.Nifti1Image(img, aff), str(tmp_path / "i.nii.gz"))
    nifti.save(nifti.Nifti1Image(lab, aff), str(tmp_path / "l.nii.gz"))
    image = nifti.load(str(tmp_path / "i.nii.gz")).get_data().copy()
    label = nifti.load(str(tmp_path / "l.nii.gz"))
    mask = (label.get_data() > 3)
    vol = nifti.Nifti1Image(mask.astype(np.int32), label.affine)
    nifti.save(vol, str(tmp_path / "out.nii"))
    back = nifti.load(str(tmp_path / "out.nii"))
    assert image.shape == (8, 8, 4) and back.get_data().dtype == np.int32
    np.testing.assert_array_equal(back.get_data(), mask.astype(np.int32))
    np.testing.assert_allclose(back.affine, aff, atol=1e-6)


@pytest.mark.parametrize("flip", [False, True])
def test_saved_qform_fields_carry_the_affine(tmp_path, flip):
    """save() fills the quaternion / offsets / qfac from the affine as nibabel's Nifti1Image(array, affine) does (sform code 2,
    qform code 0): flipping the codes in the written header (qform 1, sform 0) must give the same affine back for a
    rotation + anisotropic scale, with and without a reflection (qfac = -1)."""
    ang = 0.3
    rz = np.array([[np.cos(ang), -np.sin(ang), 0.0], [np.sin(ang), np.cos(ang), 0.0], [0.0, 0.0, 1.0]])
    rx = np.array([[1.0, 0.0, 0.0], [0.0, np.cos(0.7), -np.sin(0.7)], [0.0, np.sin(0.7), np.cos(0.7)]])
    aff = np.eye(4)
    aff[:3, :3] = (rz @ rx) * np.array([0.8, 1.25, -2.5 if flip else 2.5])
    aff[:3, 3] = [-12.5, 40.0, 7.25]
    p = str(tmp_path / "q.nii")
    nifti.save(nifti.Nifti1Image(np.zeros((3, 4, 5), np.int16), aff), p)
    img = nifti.load(p)
    assert (img.header["qform_code"], img.header["sform_code"]) == (0, 2)
    np.testing.assert_allclose(img.affine, aff, atol=1e-5)
    assert img.header["pixdim"][0] == (-1.0 if flip else 1.0)
    raw = bytearray(open(p, "rb").read())
    struct.pack_into("<2h", raw, 252, 1, 0)
    open(p, "wb").write(bytes(raw))
    np.testing.assert_allclose(nifti.load(p).affine, aff, atol=1e-5)


def test_load_returns_a_writable_array_for_gz(tmp_path):
    a = np.arange(60, dtype=np.float32).reshape(3, 4, 5)
    p = str(tmp_path / "w.nii.gz")
    nifti.save(nifti.Nifti1Image(a, np.eye(4)), p)
    d = nifti.load(p).get_data()
    np.testing.assert_array_equal(d, a)
    d[0, 0, 0] = 5.0            # nibabel's get_data() hands out a writable array
    with open(str(tmp_path / "short.nii"), "wb") as f:
        f.write(open(p, "rb").read()[:100])


def test_spec_fixture_sform_scaled():
    """A file written byte by byte from the NIfTI-1 header layout (tests/golden/gen_nifti_fixture.py; not by cfun_amd.nifti):
    int16 [3,2,2] stored x + 3y + 6z, scl_slope 2 / scl_inter -5, sform and qform both coded -- the sform wins."""
    import os
    from conftest import GOLDEN
    img = nifti.load(os.path.join(GOLDEN, "nifti1_sform.nii"))
    x, y, z = np.meshgrid(np.arange(3), np.arange(2), np.arange(2), indexing="ij")
    expect = (x + 3 * y + 6 * z) * 2.0 - 5.0
    assert img.shape == (3, 2, 2)
    np.testing.assert_array_equal(img.get_data(), expect)
    np.testing.assert_array_equal(img.affine, np.array([[2.0, 0.0, 0.5, -10.0], [0.0, 1.5, 0.0, 20.0], [-0.25, 0.0, 3.0, 30.0],
                                                        [0.0, 0.0, 0.0, 1.0]]))
    assert img.header["sform_code"] == 1 and img.header["qform_code"] == 1 and img.header["datatype"] == 4


def test_spec_fixture_qform_big_endian_4d():
    """Big-endian float32 [2,3,1,2], unscaled, qform only: quaternion (0, 0, sqrt(1/2)) = 90 degrees about z, pixdim (2,3,4),
    qfac -1 -> affine columns (0,2,0), (-3,0,0), (0,0,-4), offsets (7,-8,9) -- the standard's METHOD 2, worked by hand."""
    import os
    from conftest import GOLDEN
    img = nifti.load(os.path.join(GOLDEN, "nifti1_qform_be.nii"))
    vals = (0.5 * np.arange(12) - 1.0).astype(np.float32)
    assert img.shape == (2, 3, 1, 2) and img.get_data().dtype == np.float32
    np.testing.assert_array_equal(img.get_data(), vals.reshape((2, 3, 1, 2), order="F"))
    np.testing.assert_allclose(img.affine, np.array([[0.0, -3.0, 0.0, 7.0], [2.0, 0.0, 0.0, -8.0], [0.0, 0.0, -4.0, 9.0],
                                                     [0.0, 0.0, 0.0, 1.0]]), rtol=0, atol=1e-6)


def test_writer_matches_the_spec_layout(tmp_path):
    """The writer's header read back with plain struct at the offsets of nifti1.h (not with the reader under test)."""
    import struct
    aff = np.array([[2.0, 0.0, 0.5, -10.0], [0.0, 1.5, 0.0, 20.0], [-0.25, 0.0, 3.0, 30.0], [0.0, 0.0, 0.0, 1.0]])
    data = (np.arange(24, dtype=np.int16).reshape(2, 3, 4) - 7)
    p = tmp_path / "w.nii"
    nifti.save(nifti.Nifti1Image(data, aff), str(p))
    raw = p.read_bytes()
    assert struct.unpack("<i", raw[:4])[0] == 348 and raw[344:348] == b"n+1\0" and raw[348:352] == b"\0\0\0\0"
    assert struct.unpack("<8h", raw[40:56])[:4] == (3, 2, 3, 4)
    assert struct.unpack("<2h", raw[70:74]) == (4, 16)
    assert struct.unpack("<3f", raw[108:120]) == (352.0, 1.0, 0.0)
    assert struct.unpack("<2h", raw[252:256])[1] > 0
    np.testing.assert_allclose(np.array(struct.unpack("<12f", raw[280:328])).reshape(3, 4), aff[:3], rtol=0, atol=1e-6)
    stored = np.frombuffer(raw[352:], dtype="<i2")
    np.testing.assert_array_equal(stored, data.reshape(-1, order="F"))       # first index fastest
    assert len(raw) == 352 + 2 * 24
