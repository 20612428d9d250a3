"""Minimal NIfTI-1 single-file writer / reader (``.nii`` and ``.nii.gz``) for the label maps and volumes the reference
saves with ``nib.save(nib.Nifti1Image(array, affine), path)`` (``/root/reference/engine/test.py:158-170``); nibabel is not
available in this image.  Follows the published NIfTI-1.1 header layout (348 bytes + 4 extension bytes, data at offset
352, little endian): the affine goes to the sform rows with ``sform_code = 2`` (aligned), the quaternion fields are filled
from the same affine with ``qform_code = 0``.  The output is a valid NIfTI-1 file that readers place like the
reference's (same affine, dims, dtype); byte-for-byte equality with nibabel's header is NOT claimed (parity unpinned:
nibabel is absent and no reference fixture holds a header).  Host-side I/O: out of the GPU hot path."""
from __future__ import annotations

import gzip
import struct

import numpy as np

_DTYPES = {np.dtype("uint8"): (2, 8), np.dtype("int16"): (4, 16), np.dtype("int32"): (8, 32), np.dtype("float32"): (16, 32),
           np.dtype("float64"): (64, 64), np.dtype("int8"): (256, 8), np.dtype("uint16"): (512, 16)}
_CODES = {v[0]: k for k, v in _DTYPES.items()}


def _quaternion(R):
    """rotation matrix (det > 0) -> (b, c, d) of the unit quaternion with a >= 0 (NIfTI-1 nifti_mat44_to_quatern)"""
    r11, r12, r13, r21, r22, r23, r31, r32, r33 = R.reshape(-1)
    a = r11 + r22 + r33 + 1.0
    if a > 0.5:
        a = 0.5 * np.sqrt(a)
        b, c, d = 0.25 * (r32 - r23) / a, 0.25 * (r13 - r31) / a, 0.25 * (r21 - r12) / a
    else:
        xd, yd, zd = 1.0 + r11 - (r22 + r33), 1.0 + r22 - (r11 + r33), 1.0 + r33 - (r11 + r22)
        if xd > 1.0:
            b = 0.5 * np.sqrt(xd)
            c, d, a = 0.25 * (r12 + r21) / b, 0.25 * (r13 + r31) / b, 0.25 * (r32 - r23) / b
        elif yd > 1.0:
            c = 0.5 * np.sqrt(yd)
            b, d, a = 0.25 * (r12 + r21) / c, 0.25 * (r23 + r32) / c, 0.25 * (r13 - r31) / c
        else:
            d = 0.5 * np.sqrt(zd)
            b, c, a = 0.25 * (r13 + r31) / d, 0.25 * (r23 + r32) / d, 0.25 * (r21 - r12) / d
        if a < 0.0:
            b, c, d = -b, -c, -d
    return float(b), float(c), float(d)


def save_nifti(path: str, array, affine) -> None:
    a = np.ascontiguousarray(np.asarray(array))
    if a.dtype not in _DTYPES:
        raise ValueError(f"unsupported dtype {a.dtype}")
    if not 1 <= a.ndim <= 7:
        raise ValueError("NIfTI-1 holds 1 to 7 dimensions")
    aff = np.asarray(affine, dtype=np.float64).reshape(4, 4)
    code, bits = _DTYPES[a.dtype]
    dim = [a.ndim] + list(a.shape) + [1] * (7 - a.ndim)
    M = aff[:3, :3]
    vox = np.sqrt((M * M).sum(0))
    vox[vox == 0] = 1.0
    R = M / vox
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R = R.copy()
        R[:, 2] = -R[:, 2]
        qfac = -1.0
    # nearest orthogonal matrix (polar decomposition), as the reference implementation does before the quaternion
    U, _, Vt = np.linalg.svd(R)
    qb, qc, qd = _quaternion(U @ Vt)
    pixdim = [qfac] + [float(v) for v in vox] + [1.0] * 4
    hdr = struct.pack("<i10s18sihcB", 348, b"", b"", 0, 0, b"r", 0)
    hdr += struct.pack("<8h", *dim)
    hdr += struct.pack("<3f", 0.0, 0.0, 0.0)                          # intent_p1..3
    hdr += struct.pack("<4h", 0, code, bits, 0)                       # intent_code, datatype, bitpix, slice_start
    hdr += struct.pack("<8f", *pixdim)
    hdr += struct.pack("<f", 352.0)                                   # vox_offset
    hdr += struct.pack("<2f", float("nan"), float("nan"))             # scl_slope, scl_inter: "no scaling", as nibabel leaves them
    hdr += struct.pack("<hBB", 0, 0, 0)                               # slice_end, slice_code, xyzt_units (unknown: nothing here knows the units)
    hdr += struct.pack("<4f", 0.0, 0.0, 0.0, 0.0)                     # cal_max, cal_min, slice_duration, toffset
    hdr += struct.pack("<2i", 0, 0)                                   # glmax, glmin
    hdr += struct.pack("<80s24s", b"", b"")                           # descrip, aux_file
    hdr += struct.pack("<2h", 0, 2)                                   # qform_code (unknown), sform_code (aligned)
    hdr += struct.pack("<6f", qb, qc, qd, float(aff[0, 3]), float(aff[1, 3]), float(aff[2, 3]))
    hdr += struct.pack("<12f", *[float(v) for v in aff[:3].reshape(-1)])
    hdr += struct.pack("<16s4s", b"", b"n+1\0")
    assert len(hdr) == 348
    # NIfTI stores the first index fastest: Fortran order of the array as nibabel sees it
    payload = hdr + b"\0\0\0\0" + a.astype(a.dtype.newbyteorder("<"), copy=False).tobytes(order="F")
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "wb") as f:
        f.write(payload)


def load_nifti(path: str):
    """-> (array, affine from the sform rows); enough of a reader to round-trip what save_nifti writes"""
    opener = gzip.open if path.endswith(".gz") else open
    with opener(path, "rb") as f:
        raw = f.read()
    if struct.unpack("<i", raw[:4])[0] != 348 or raw[344:348] != b"n+1\0":
        raise ValueError("not a little-endian single-file NIfTI-1")
    dim = struct.unpack("<8h", raw[40:56])
    code, bits = struct.unpack("<2h", raw[70:74])
    off = int(struct.unpack("<f", raw[108:112])[0])
    shape = dim[1:1 + dim[0]]
    dt = _CODES[code].newbyteorder("<")
    data = np.frombuffer(raw, dtype=dt, count=int(np.prod(shape)), offset=off).reshape(shape, order="F")
    aff = np.eye(4)
    aff[:3] = np.array(struct.unpack("<12f", raw[280:328])).reshape(3, 4)
    return data, aff
