"""Dependency-free NIfTI-1 single-file (.nii / .nii.gz) reader and writer.

The reference does all volume I/O through nibabel 5.3.2 (requirements.txt:39), which exists neither in this
image nor on the GPU box: `nib.load(path)` + `.get_fdata()` + `.affine` + `.shape` / `.ndim`
(data_loading/dataset.py:56-75, utils.py:369-397) and `nib.Nifti1Image(data, affine)` + `nib.save` (utils.py:341-350).
This module restates exactly that surface from the published NIfTI-1 specification (nifti1.h):

  * `load(path)`            -> Nifti1Image        (little- or big-endian, gzip detected by magic bytes, header extensions skipped)
  * `Nifti1Image(data, affine)`, `.get_fdata()` (float64, scl_slope/scl_inter applied like nibabel: slope 0 or NaN = unscaled),
    `.affine` (sform if sform_code > 0, else qform, else the pixdim fallback -- nibabel's get_best_affine order),
    `.shape`, `.ndim`, `.to_filename(path)`
  * `save(img, path)`       header as nibabel fills it for an image built from (data, affine): sform_code 2 ("aligned") with the
    affine in srow_*, the quaternion of its rotation in quatern_*/qoffset_* with qform_code 0 ("unknown"), pixdim = column norms,
    vox_offset 352, data in Fortran order.  gzip level 1 like nibabel's default.

Parity: unpinned against nibabel itself (absent); pinned against the NIfTI-1 field layout by tests/test_nifti.py, which parses
the written bytes with an independent struct-based reader and feeds this reader hand-assembled files.
"""
from __future__ import annotations

import gzip
import os
from typing import Optional, Sequence, Tuple

import numpy as np

_HDR_FIELDS = [
    ("sizeof_hdr", "i4"), ("data_type", "S10"), ("db_name", "S18"), ("extents", "i4"), ("session_error", "i2"),
    ("regular", "S1"), ("dim_info", "u1"), ("dim", "i2", (8,)), ("intent_p1", "f4"), ("intent_p2", "f4"),
    ("intent_p3", "f4"), ("intent_code", "i2"), ("datatype", "i2"), ("bitpix", "i2"), ("slice_start", "i2"),
    ("pixdim", "f4", (8,)), ("vox_offset", "f4"), ("scl_slope", "f4"), ("scl_inter", "f4"), ("slice_end", "i2"),
    ("slice_code", "u1"), ("xyzt_units", "u1"), ("cal_max", "f4"), ("cal_min", "f4"), ("slice_duration", "f4"),
    ("toffset", "f4"), ("glmax", "i4"), ("glmin", "i4"), ("descrip", "S80"), ("aux_file", "S24"),
    ("qform_code", "i2"), ("sform_code", "i2"), ("quatern_b", "f4"), ("quatern_c", "f4"), ("quatern_d", "f4"),
    ("qoffset_x", "f4"), ("qoffset_y", "f4"), ("qoffset_z", "f4"), ("srow_x", "f4", (4,)), ("srow_y", "f4", (4,)),
    ("srow_z", "f4", (4,)), ("intent_name", "S16"), ("magic", "S4"),
]
HEADER_DTYPE = np.dtype(_HDR_FIELDS)
assert HEADER_DTYPE.itemsize == 348

# NIfTI-1 datatype codes (nifti1.h DT_*)
_CODE_TO_DTYPE = {2: "u1", 4: "i2", 8: "i4", 16: "f4", 64: "f8", 256: "i1", 512: "u2", 768: "u4", 1024: "i8", 1280: "u8"}
_DTYPE_TO_CODE = {np.dtype(v): k for k, v in _CODE_TO_DTYPE.items()}


class NiftiError(ValueError):
    pass


def _quat_to_rot(b: float, c: float, d: float) -> np.ndarray:
    a2 = 1.0 - (b * b + c * c + d * d)
    a = np.sqrt(a2) if a2 > 0 else 0.0
    return np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                     [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                     [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]], dtype=np.float64)


def _rot_to_quat(R: np.ndarray) -> Tuple[float, float, float, float]:
    """Unit quaternion (a >= 0) of a proper rotation, by the largest-component branch."""
    t = np.trace(R)
    if t > 0:
        a = 0.5 * np.sqrt(1.0 + t)
        b, c, d = (R[2, 1] - R[1, 2]) / (4 * a), (R[0, 2] - R[2, 0]) / (4 * a), (R[1, 0] - R[0, 1]) / (4 * a)
    else:
        i = int(np.argmax(np.diag(R)))
        j, k = (i + 1) % 3, (i + 2) % 3
        q = np.zeros(4)
        q[1 + i] = 0.5 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0))
        q[0] = (R[k, j] - R[j, k]) / (4 * q[1 + i])
        q[1 + j] = (R[j, i] + R[i, j]) / (4 * q[1 + i])
        q[1 + k] = (R[k, i] + R[i, k]) / (4 * q[1 + i])
        a, b, c, d = q
    if a < 0:
        a, b, c, d = -a, -b, -c, -d
    return float(a), float(b), float(c), float(d)


class Nifti1Image:
    """The part of nibabel's Nifti1Image the reference uses: built from (data, affine) or by `load`."""

    def __init__(self, dataobj, affine, header: Optional[np.ndarray] = None, slope: float = 1.0, inter: float = 0.0):
        self._data = np.asarray(dataobj)
        self.affine = None if affine is None else np.array(affine, dtype=np.float64).reshape(4, 4)
        self.header = header
        self._slope, self._inter = float(slope), float(inter)
        self._fdata = None

    @property
    def dataobj(self) -> np.ndarray:
        return self._data

    @property
    def shape(self) -> Tuple[int, ...]:
        return tuple(self._data.shape)

    @property
    def ndim(self) -> int:
        return self._data.ndim

    def get_data_dtype(self):
        return self._data.dtype

    def get_fdata(self) -> np.ndarray:
        """float64 array with the header's intensity scaling applied (cached, like nibabel's default caching)."""
        if self._fdata is None:
            arr = self._data.astype(np.float64)
            s, i = self._slope, self._inter
            if np.isfinite(s) and s != 0.0 and not (s == 1.0 and (i == 0.0 or not np.isfinite(i))):
                arr = arr * s + (i if np.isfinite(i) else 0.0)
            self._fdata = arr
        return self._fdata

    def to_filename(self, path: str) -> None:
        save(self, path)


def _read_all(path: str) -> bytes:
    with open(path, "rb") as f:
        raw = f.read()
    if raw[:2] == b"\x1f\x8b":
        raw = gzip.decompress(raw)
    return raw


def _header_affine(h: np.ndarray, shape: Sequence[int]) -> np.ndarray:
    if int(h["sform_code"]) != 0:
        A = np.eye(4)
        A[0], A[1], A[2] = h["srow_x"].astype(np.float64), h["srow_y"].astype(np.float64), h["srow_z"].astype(np.float64)
        return A
    zooms = np.abs(h["pixdim"][1:4].astype(np.float64))
    if int(h["qform_code"]) != 0:
        R = _quat_to_rot(float(h["quatern_b"]), float(h["quatern_c"]), float(h["quatern_d"]))
        qfac = -1.0 if float(h["pixdim"][0]) < 0 else 1.0
        A = np.eye(4)
        A[:3, :3] = R * (zooms * np.array([1.0, 1.0, qfac]))
        A[:3, 3] = [float(h["qoffset_x"]), float(h["qoffset_y"]), float(h["qoffset_z"])]
        return A
    # neither transform set: Analyze-style fallback, origin at the volume centre, x flipped (nibabel's base affine)
    z = zooms.copy()
    z[0] *= -1.0
    sh = np.array(list(shape[:3]) + [1] * (3 - len(shape[:3])), dtype=np.float64)
    A = np.diag(np.append(z, 1.0))
    A[:3, 3] = -((sh - 1.0) / 2.0) * z
    return A


def load(path: str) -> Nifti1Image:
    raw = _read_all(path)
    if len(raw) < 352:
        raise NiftiError(f"{path}: {len(raw)} bytes is too short for a NIfTI-1 header")
    h = np.frombuffer(raw, dtype=HEADER_DTYPE, count=1)[0]
    if int(h["sizeof_hdr"]) != 348:
        h = np.frombuffer(raw, dtype=HEADER_DTYPE.newbyteorder(">"), count=1)[0]
        if int(h["sizeof_hdr"]) != 348:
            raise NiftiError(f"{path}: sizeof_hdr is not 348 in either byte order (not NIfTI-1)")
        order = ">"
    else:
        order = "<"
    magic = bytes(h["magic"])
    if not magic.startswith(b"n+1"):
        raise NiftiError(f"{path}: magic {magic!r}: only single-file NIfTI-1 ('n+1') is supported")
    nd = int(h["dim"][0])
    if not 1 <= nd <= 7:
        raise NiftiError(f"{path}: dim[0] = {nd}")
    shape = tuple(int(x) for x in h["dim"][1:1 + nd])
    code = int(h["datatype"])
    if code not in _CODE_TO_DTYPE:
        raise NiftiError(f"{path}: datatype code {code} is not supported")
    dt = np.dtype(_CODE_TO_DTYPE[code]).newbyteorder(order)
    off = int(h["vox_offset"])
    if off < 352:
        off = 352
    count = int(np.prod(shape, dtype=np.int64))
    if len(raw) < off + count * dt.itemsize:
        raise NiftiError(f"{path}: file holds {len(raw) - off} data bytes, header promises {count * dt.itemsize}")
    data = np.frombuffer(raw, dtype=dt, count=count, offset=off).reshape(shape, order="F")
    if order == ">":
        data = data.astype(dt.newbyteorder("="))
    return Nifti1Image(data, _header_affine(h, shape), header=h.copy(), slope=float(h["scl_slope"]), inter=float(h["scl_inter"]))


def make_header(data: np.ndarray, affine: np.ndarray) -> np.ndarray:
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    if data.dtype not in _DTYPE_TO_CODE:
        raise NiftiError(f"dtype {data.dtype} has no NIfTI-1 datatype code")
    if not 1 <= data.ndim <= 7:
        raise NiftiError(f"{data.ndim}-d arrays do not fit NIfTI-1")
    if max(data.shape) > 32767:
        raise NiftiError("NIfTI-1 dims are int16")
    h = np.zeros((), dtype=HEADER_DTYPE)
    h["sizeof_hdr"] = 348
    h["regular"] = b"r"
    dim = np.ones(8, dtype=np.int16)
    dim[0] = data.ndim
    dim[1:1 + data.ndim] = data.shape
    h["dim"] = dim
    h["datatype"] = _DTYPE_TO_CODE[data.dtype]
    h["bitpix"] = data.dtype.itemsize * 8
    A = np.asarray(affine, dtype=np.float64).reshape(4, 4)
    RZS = A[:3, :3]
    zooms = np.sqrt((RZS * RZS).sum(axis=0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy()
        R[:, 2] *= -1.0
        qfac = -1.0
    U, _, Vt = np.linalg.svd(R)                # nearest orthogonal matrix, as nibabel's set_qform does
    _, b, c, d = _rot_to_quat(U @ Vt)
    pix = np.ones(8, dtype=np.float32)
    pix[0] = qfac
    pix[1:4] = zooms
    h["pixdim"] = pix
    h["vox_offset"] = 352.0
    h["scl_slope"], h["scl_inter"] = 1.0, 0.0
    h["qform_code"], h["sform_code"] = 0, 2
    h["quatern_b"], h["quatern_c"], h["quatern_d"] = b, c, d
    h["qoffset_x"], h["qoffset_y"], h["qoffset_z"] = A[0, 3], A[1, 3], A[2, 3]
    h["srow_x"], h["srow_y"], h["srow_z"] = A[0], A[1], A[2]
    h["magic"] = b"n+1\0"
    return h


def save(img: Nifti1Image, path: str, compresslevel: int = 1) -> None:
    data = img.dataobj
    if data.dtype == np.bool_:
        data = data.astype(np.uint8)
    data = data.astype(data.dtype.newbyteorder("<"), copy=False) if data.dtype.byteorder == ">" else data
    affine = img.affine if img.affine is not None else np.eye(4)
    h = make_header(data, affine)
    payload = h.tobytes() + b"\0\0\0\0" + np.asfortranarray(data).tobytes(order="F")
    d = os.path.dirname(path)
    if d:
        os.makedirs(d, exist_ok=True)
    if path.endswith(".gz"):
        with open(path, "wb") as f, gzip.GzipFile(filename="", mode="wb", compresslevel=compresslevel, fileobj=f, mtime=0) as g:
            g.write(payload)
    else:
        with open(path, "wb") as f:
            f.write(payload)
