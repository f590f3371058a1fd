"""NIfTI-1 reader / writer (cinema_amd/nifti.py) against the published field layout (nifti1.h), with an independent
struct-based parser / assembler so that reader and writer are not only checked against each other.  nibabel is absent:
parity with nibabel's byte output is unpinned; what is pinned is the specification."""
import gzip
import os
import struct

import numpy as np
import pytest

from cinema_amd import nifti
from cinema_amd import utils as U


def _spec_parse(raw: bytes):
    """Field offsets straight from nifti1.h."""
    f = {}
    f["sizeof_hdr"], = struct.unpack_from("<i", raw, 0)
    f["dim"] = struct.unpack_from("<8h", raw, 40)
    f["datatype"], f["bitpix"] = struct.unpack_from("<hh", raw, 70)
    f["pixdim"] = struct.unpack_from("<8f", raw, 76)
    f["vox_offset"], f["scl_slope"], f["scl_inter"] = struct.unpack_from("<fff", raw, 108)
    f["qform_code"], f["sform_code"] = struct.unpack_from("<hh", raw, 252)
    f["quatern"] = struct.unpack_from("<3f", raw, 256)
    f["qoffset"] = struct.unpack_from("<3f", raw, 268)
    f["srow"] = np.array(struct.unpack_from("<12f", raw, 280)).reshape(3, 4)
    f["magic"] = raw[344:348]
    return f


def _spec_build(data, order="<", sform=None, quat=None, qoff=(0, 0, 0), pixdim=(1, 1, 1, 1), slope=0.0, inter=0.0, code=None,
                ext=b""):
    """Assemble a file by hand: independent of nifti.make_header."""
    codes = {"u1": 2, "i2": 4, "i4": 8, "f4": 16, "f8": 64}
    hdr = bytearray(348)
    struct.pack_into(order + "i", hdr, 0, 348)
    dim = [data.ndim] + list(data.shape) + [1] * (7 - data.ndim)
    struct.pack_into(order + "8h", hdr, 40, *dim)
    struct.pack_into(order + "hh", hdr, 70, code or codes[data.dtype.str[1:]], data.dtype.itemsize * 8)
    struct.pack_into(order + "8f", hdr, 76, pixdim[0], pixdim[1], pixdim[2], pixdim[3], 1, 1, 1, 1)
    struct.pack_into(order + "fff", hdr, 108, 352 + len(ext), slope, inter)
    struct.pack_into(order + "hh", hdr, 252, 1 if quat is not None else 0, 1 if sform is not None else 0)
    if quat is not None:
        struct.pack_into(order + "3f", hdr, 256, *quat)
        struct.pack_into(order + "3f", hdr, 268, *qoff)
    if sform is not None:
        struct.pack_into(order + "12f", hdr, 280, *np.asarray(sform, dtype=np.float64)[:3].reshape(-1))
    hdr[344:348] = b"n+1\0"
    flag = b"\1\0\0\0" if ext else b"\0\0\0\0"
    return bytes(hdr) + flag + ext + data.astype(data.dtype.newbyteorder(order)).tobytes(order="F")


AFF = np.array([[0.8, 0.05, 0.0, -6.3], [-0.04, 0.8, 0.02, -5.1], [0.0, -0.03, 0.8, -4.4], [0, 0, 0, 1.0]])


@pytest.mark.parametrize("dtype", ["uint8", "int16", "int32", "float32", "float64"])
@pytest.mark.parametrize("gz", [True, False])
def test_written_header_matches_specification(tmp_path, dtype, gz):
    rng = np.random.default_rng(1)
    data = (rng.random((5, 6, 7)) * 100).astype(dtype)
    path = str(tmp_path / ("a.nii.gz" if gz else "a.nii"))
    nifti.save(nifti.Nifti1Image(data, AFF), path)
    raw = open(path, "rb").read()
    assert (raw[:2] == b"\x1f\x8b") == gz
    if gz:
        raw = gzip.decompress(raw)
    f = _spec_parse(raw)
    assert f["sizeof_hdr"] == 348 and f["magic"] == b"n+1\0" and f["vox_offset"] == 352.0
    assert f["dim"][:4] == (3, 5, 6, 7) and all(d == 1 for d in f["dim"][4:])
    assert f["datatype"] == {"uint8": 2, "int16": 4, "int32": 8, "float32": 16, "float64": 64}[dtype]
    assert f["bitpix"] == data.dtype.itemsize * 8
    assert f["sform_code"] == 2 and f["qform_code"] == 0
    assert np.allclose(f["srow"], AFF[:3], atol=1e-6)
    assert np.allclose(f["pixdim"][1:4], np.sqrt((AFF[:3, :3] ** 2).sum(0)), atol=1e-6) and f["pixdim"][0] == 1.0
    assert np.allclose(f["qoffset"], AFF[:3, 3], atol=1e-6)
    assert len(raw) == 352 + data.size * data.dtype.itemsize
    back = np.frombuffer(raw, dtype=data.dtype, offset=352).reshape(data.shape, order="F")      # x fastest on disk
    assert np.array_equal(back, data)
    img = nifti.load(path)
    assert img.shape == data.shape and img.ndim == 3
    assert np.array_equal(img.get_fdata(), data.astype(np.float64)) and img.get_fdata().dtype == np.float64
    assert np.allclose(img.affine, AFF, atol=1e-6)


def test_quaternion_written_is_the_rotation_of_the_affine(tmp_path):
    th = 0.3
    R = np.array([[np.cos(th), -np.sin(th), 0], [np.sin(th), np.cos(th), 0], [0, 0, 1]])
    A = np.eye(4)
    A[:3, :3] = R @ np.diag([0.5, 0.6, -0.7])           # improper: qfac must come out -1
    A[:3, 3] = [1, 2, 3]
    path = str(tmp_path / "q.nii")
    nifti.save(nifti.Nifti1Image(np.zeros((2, 3, 4), np.float32), A), path)
    raw = bytearray(open(path, "rb").read())
    f = _spec_parse(bytes(raw))
    assert f["pixdim"][0] == -1.0 and np.allclose(f["pixdim"][1:4], [0.5, 0.6, 0.7], atol=1e-6)
    struct.pack_into("<hh", raw, 252, 1, 0)             # make readers take the qform
    open(path, "wb").write(bytes(raw))
    assert np.allclose(nifti.load(path).affine, A, atol=1e-5)


@pytest.mark.parametrize("order", ["<", ">"])
def test_reader_on_hand_assembled_files(tmp_path, order):
    rng = np.random.default_rng(2)
    data = rng.integers(-500, 500, size=(4, 5, 6)).astype(np.int16)
    p1 = str(tmp_path / "s.nii.gz")
    open(p1, "wb").write(gzip.compress(_spec_build(data, order, sform=AFF, slope=0.5, inter=10.0, ext=b"\0" * 16)))
    img = nifti.load(p1)
    assert np.array_equal(img.get_fdata(), data.astype(np.float64) * 0.5 + 10.0)        # scaling applied, extension skipped
    assert np.allclose(img.affine, AFF, atol=1e-6)
    # qform only: 90 degrees about z (b, c, d) = (0, 0, sin 45), zooms (2, 3, 4), qfac -1
    p2 = str(tmp_path / "q.nii")
    s = np.sqrt(0.5)
    open(p2, "wb").write(_spec_build(data, order, quat=(0, 0, s), qoff=(7, 8, 9), pixdim=(-1, 2, 3, 4)))
    exp = np.array([[0, -3, 0, 7], [2, 0, 0, 8], [0, 0, -4, 9], [0, 0, 0, 1.0]])
    assert np.allclose(nifti.load(p2).affine, exp, atol=1e-5)
    assert np.array_equal(nifti.load(p2).get_fdata(), data.astype(np.float64))          # slope 0 = unscaled
    # 4-d (x, y, z, 1) volume as in dHCP: read as 3-d, the file is left alone
    from cinema_amd.dataset import read_subject_volumes
    p3 = str(tmp_path / "d4.nii.gz")
    open(p3, "wb").write(gzip.compress(_spec_build(data[..., None].astype(np.float32), order, sform=AFF)))
    vols, aff = read_subject_volumes({"dataset": {"modalities": ["T2w"]}}, {"T2w": p3})
    assert vols[0].shape == data.shape and nifti.load(p3).shape == data.shape + (1,) and np.allclose(aff, AFF, atol=1e-6)


def test_reader_rejects_garbage(tmp_path):
    p = str(tmp_path / "bad.nii")
    open(p, "wb").write(b"\0" * 400)
    with pytest.raises(nifti.NiftiError):
        nifti.load(p)
    open(p, "wb").write(b"\0" * 10)
    with pytest.raises(nifti.NiftiError):
        nifti.load(p)
    raw = bytearray(_spec_build(np.zeros((2, 2, 2), np.float32), sform=AFF))
    open(p, "wb").write(bytes(raw[:-8]))                 # truncated data
    with pytest.raises(nifti.NiftiError):
        nifti.load(p)


def test_save_img_dtype_rules_and_save_atlas_names(tmp_path):
    """Writer behaviour of SURVEY.md section 5 "Output naming": int64/int32 -> int16, float64 -> float32; file naming, 4-d per channel."""
    out = str(tmp_path)
    U.save_img(np.arange(24, dtype=np.int64).reshape(2, 3, 4), np.eye(4), out, "sub/a.nii.gz")
    U.save_img(np.ones((2, 3, 4), dtype=np.float64), np.eye(4), out, "sub/b.nii.gz")
    assert nifti.load(os.path.join(out, "sub/a.nii.gz")).get_data_dtype() == np.int16
    assert nifti.load(os.path.join(out, "sub/b.nii.gz")).get_data_dtype() == np.float32
    import torch
    args = {"dataset": {"modalities": ["T2w", "Seg"], "label_names": ["BG", "CSF", "GM"]}, "save_certainty_maps": True,
            "output_dir": out}
    atlas = torch.rand(4, 5, 6, 2 + 3, 2, 3)             # [x, y, z, modalities(+certainty maps), conditions, t]
    U.save_atlas(args, atlas, torch.diag(torch.tensor([0.5, 0.5, 0.5, 1.0])), [37, 38, 39], [[0.1], [0.2]], epoch=7)
    names = ["T2w", "Seg", "CertaintyMaps/BG", "CertaintyMaps/CSF", "CertaintyMaps/GM"]
    for c in range(2):
        for i, nme in enumerate(names):
            img = nifti.load(os.path.join(out, f"{nme}_ga=37-39_cond={c}_ep=7.nii.gz"))
            assert img.shape == (4, 5, 6, 3)
            assert np.array_equal(img.get_fdata(), atlas[..., i, c, :].numpy().astype(np.float64))
            assert np.allclose(np.diag(img.affine), [0.5, 0.5, 0.5, 1.0])


def test_premask_matches_scipy_halo():
    """mask_reconstruction at load time (reference behaviour utils.py:385-397): intensities masked by seg > 0, background
    shell where scipy's Gaussian (sigma 1.5) of the foreground mask exceeds 0.001 -- scipy is the checker here."""
    import scipy.ndimage as ndi
    import torch
    from cinema_amd.dataset import premask_on_device
    rng = np.random.default_rng(3)
    seg = np.zeros((20, 17, 15))
    seg[5:13, 4:9, 6:11] = rng.integers(1, 4, size=(8, 5, 5))
    seg[0:2, 0:3, 12:15] = 2                                         # touches the border: mirrored edges matter
    t2 = rng.random(seg.shape) + 0.5
    got = premask_on_device([torch.from_numpy(t2), torch.from_numpy(seg)], ["none", "a", "b", "c", "BG"])
    want = ((ndi.gaussian_filter((seg > 0).astype(np.float32), sigma=1.5) > 0.001) * 4).astype(np.float64)
    want[seg > 0] = seg[seg > 0]
    assert np.array_equal(got[1].numpy(), want) and (want == 4).sum() > 100
    assert np.array_equal(got[0].numpy(), t2 * (seg > 0))


def test_subject_reading_checks(tmp_path):
    from cinema_amd import nifti
    from cinema_amd.dataset import read_subject_volumes, condition_row
    args = {"dataset": {"modalities": ["T2w", "Seg"], "label_names": ["none", "a", "b", "BG"],
                        "conditions": {"birth_age": True, "sex": False}, "constraints": {"birth_age": {"min": 30, "max": 40}}},
            "mask_reconstruction": True, "atlas_gen": {"cond_scale": 0.5}}
    p = lambda n: str(tmp_path / n)
    seg = np.zeros((10, 10, 10), np.uint8)
    seg[3:7, 3:7, 3:7] = 2
    nifti.save(nifti.Nifti1Image(np.full((10, 10, 10), 5.0, np.float32), np.eye(4)), p("t2.nii.gz"))
    nifti.save(nifti.Nifti1Image(seg, np.eye(4)), p("seg.nii.gz"))
    vols, aff = read_subject_volumes(args, {"T2w": p("t2.nii.gz"), "Seg": p("seg.nii.gz")})
    assert vols[0].dtype == np.float64 and vols[0][5, 5, 5] == 5.0 and vols[1][5, 5, 5] == 2 and np.array_equal(aff, np.eye(4))
    with pytest.raises(ValueError):
        read_subject_volumes(args, {"T2w": p("t2.nii.gz")})
    with pytest.raises(ValueError):
        read_subject_volumes(args, {"T2w": "", "Seg": p("seg.nii.gz")})
    nifti.save(nifti.Nifti1Image(seg[:9], np.eye(4)), p("seg9.nii.gz"))
    with pytest.raises(ValueError):
        read_subject_volumes(args, {"T2w": p("t2.nii.gz"), "Seg": p("seg9.nii.gz")})
    nifti.save(nifti.Nifti1Image(seg, np.diag([2.0, 1, 1, 1])), p("seg2.nii.gz"))
    with pytest.raises(ValueError):
        read_subject_volumes(args, {"T2w": p("t2.nii.gz"), "Seg": p("seg2.nii.gz")})
    row = {"birth_age": 37.5, "sex": 1}
    c = condition_row(args, row)
    assert c.shape == (1,) and abs(float(c[0]) - 0.25) < 1e-7 and row == {"birth_age": 37.5, "sex": 1}
