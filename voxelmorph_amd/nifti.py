"""NIfTI-1 (.nii / .nii.gz) and FreeSurfer .mgz volume I/O without nibabel.

The reference reads and writes its volumes through nibabel (`voxelmorph/py/utils.py:69-158`: `load_volfile` takes
nii / nii.gz / mgz / npz / npy and optionally returns the affine, `save_volfile` writes nii / nii.gz / npz); its scripts default to
`.nii.gz`.  nibabel is not part of this image, so the two formats are read and written here from their specifications with `struct`,
`gzip` and numpy: what `nib.load(f).dataobj` / `.affine` and `nib.save(nib.Nifti1Image(array, affine), f)` do for the cases the
reference's data path meets (single-file NIfTI-1, any endianness, the scalar datatypes, scl_slope / scl_inter scaling, sform / qform
/ fall-back affines; mgz types uchar / int / float / short).  Host-side only: nothing here touches the GPU.
"""
import gzip
import struct

import numpy as np

_NIFTI_DTYPES = {2: np.uint8, 4: np.int16, 8: np.int32, 16: np.float32, 64: np.float64, 256: np.int8, 512: np.uint16, 768: np.uint32,
                 1024: np.int64, 1280: np.uint64}
_NIFTI_CODES = {np.dtype(v).str[1:]: k for k, v in _NIFTI_DTYPES.items()}      # 'u1' -> 2, 'f4' -> 16, ...


def _open(filename, mode):
    return gzip.open(filename, mode) if str(filename).endswith(('.gz', '.mgz')) else open(filename, mode)


def _quaternion_affine(b, c, d, qx, qy, qz, dx, dy, dz, qfac):
    """NIfTI-1 'method 2': rotation from the quaternion (a recomputed from b, c, d), scaled by pixdim, the third column by qfac"""
    a = np.sqrt(max(1.0 - (b * b + c * c + d * d), 0.0))
    R = np.array([[a * a + b * b - c * c - d * d, 2 * (b * c - a * d), 2 * (b * d + a * c)],
                  [2 * (b * c + a * d), a * a + c * c - b * b - d * d, 2 * (c * d - a * b)],
                  [2 * (b * d - a * c), 2 * (c * d + a * b), a * a + d * d - b * b - c * c]])
    aff = np.eye(4)
    aff[:3, :3] = R * np.array([dx, dy, dz * (-1.0 if qfac < 0 else 1.0)])
    aff[:3, 3] = (qx, qy, qz)
    return aff


def read_nifti(filename):
    """-> (array, 4x4 affine).  array as nibabel's `np.asanyarray(img.dataobj)`: on-disk dtype, or float64 after scl_slope / scl_inter."""
    with _open(filename, 'rb') as f:
        raw = f.read()
    if len(raw) < 348:
        raise ValueError("'%s' is too short for a NIfTI-1 header" % filename)
    for end in ('<', '>'):
        if struct.unpack(end + 'i', raw[:4])[0] == 348:
            break
    else:
        raise ValueError("'%s' is not a NIfTI-1 file (sizeof_hdr != 348)" % filename)
    magic = raw[344:348]
    if magic not in (b'n+1\0', b'ni1\0'):
        raise ValueError("'%s': NIfTI-1 magic not found (%r)" % (filename, magic))
    if magic == b'ni1\0':
        raise ValueError("'%s' is the header of a two-file (.hdr/.img) NIfTI pair: only single-file .nii is read here" % filename)
    dim = struct.unpack(end + '8h', raw[40:56])
    datatype, bitpix = struct.unpack(end + '2h', raw[70:74])
    pixdim = struct.unpack(end + '8f', raw[76:108])
    vox_offset, slope, inter = struct.unpack(end + '3f', raw[108:120])
    qform_code, sform_code = struct.unpack(end + '2h', raw[252:256])
    qb, qc, qd, qx, qy, qz = struct.unpack(end + '6f', raw[256:280])
    srow = np.array(struct.unpack(end + '12f', raw[280:328]), dtype=np.float64).reshape(3, 4)
    ndim = dim[0]
    if not 1 <= ndim <= 7:
        raise ValueError("'%s': bad number of dimensions %d" % (filename, ndim))
    shape = tuple(int(v) for v in dim[1:1 + ndim])
    if datatype not in _NIFTI_DTYPES:
        raise ValueError("'%s': NIfTI datatype code %d is not a scalar type this reader handles" % (filename, datatype))
    dt = np.dtype(_NIFTI_DTYPES[datatype]).newbyteorder(end)
    off = int(vox_offset) if vox_offset >= 352 else 352
    n = int(np.prod(shape))
    if len(raw) < off + n * dt.itemsize:
        raise ValueError("'%s': %d data bytes expected after offset %d, %d present" % (filename, n * dt.itemsize, off, len(raw) - off))
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=off).reshape(shape, order='F')
    arr = arr.astype(dt.newbyteorder('='))                                    # native byte order, own memory
    if slope not in (0.0, 1.0) or (slope != 0.0 and inter != 0.0):
        if np.isfinite(slope) and np.isfinite(inter):
            arr = arr.astype(np.float64) * float(slope) + float(inter)
    # best affine: sform, else qform, else the header's base affine (nibabel: get_best_affine)
    if sform_code > 0:
        aff = np.vstack([srow, [0, 0, 0, 1]])
    elif qform_code > 0:
        aff = _quaternion_affine(qb, qc, qd, qx, qy, qz, pixdim[1], pixdim[2], pixdim[3], pixdim[0])
    else:
        zooms = np.array([pixdim[i + 1] if i < ndim and pixdim[i + 1] else 1.0 for i in range(3)], dtype=np.float64)
        zooms[0] *= -1.0                                                      # (nibabel's default_x_flip)
        sh3 = np.array([shape[i] if i < ndim else 1 for i in range(3)], dtype=np.float64)
        aff = np.diag(np.append(zooms, 1.0))
        aff[:3, 3] = -(sh3 - 1) / 2.0 * zooms
    return arr, aff


def _affine_quaternion(aff):
    """quaternion (b, c, d), pixdim and qfac of the rotation part of an affine (NIfTI-1 section on method 2; orthogonalised by polar
    decomposition as nibabel does)"""
    RZS = np.asarray(aff, dtype=np.float64)[:3, :3]
    zooms = np.sqrt((RZS * RZS).sum(axis=0))
    zooms[zooms == 0] = 1.0
    R = RZS / zooms
    qfac = 1.0
    if np.linalg.det(R) < 0:
        R[:, 2] *= -1.0
        qfac = -1.0
    U, _, Vt = np.linalg.svd(R)
    R = U @ Vt
    # rotation matrix -> quaternion (a >= 0)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    if tr > 0:
        a = 0.5 * np.sqrt(1.0 + tr)
        b, c, d = (R[2, 1] - R[1, 2]) / (4 * a), (R[0, 2] - R[2, 0]) / (4 * a), (R[1, 0] - R[0, 1]) / (4 * a)
    else:
        i = int(np.argmax([R[0, 0], R[1, 1], R[2, 2]]))
        j, k = (i + 1) % 3, (i + 2) % 3
        q = np.zeros(4)
        q[i + 1] = 0.5 * np.sqrt(max(1.0 + R[i, i] - R[j, j] - R[k, k], 0.0))
        q[0] = (R[k, j] - R[j, k]) / (4 * q[i + 1])
        q[j + 1] = (R[j, i] + R[i, j]) / (4 * q[i + 1])
        q[k + 1] = (R[k, i] + R[i, k]) / (4 * q[i + 1])
        if q[0] < 0:
            q = -q
        a, b, c, d = q
    return (b, c, d), zooms, qfac


def write_nifti(array, filename, affine=None):
    """`nib.save(nib.Nifti1Image(array, affine), filename)`: single-file little-endian NIfTI-1, data in Fortran order at offset 352,
    sform = affine (code 2, 'aligned'), qform filled from the same affine with code 0 ('unknown'), pixdim = its column norms."""
    array = np.asanyarray(array)
    if array.dtype == np.bool_:
        array = array.astype(np.uint8)
    key = array.dtype.newbyteorder('=').str[1:]
    if key not in _NIFTI_CODES:
        raise ValueError('array dtype %s has no NIfTI-1 scalar datatype' % array.dtype)
    if not 1 <= array.ndim <= 7:
        raise ValueError('NIfTI-1 stores 1 to 7 dimensions, got %d' % array.ndim)
    if affine is None:
        affine = np.eye(4)
    affine = np.asarray(affine, dtype=np.float64)
    (qb, qc, qd), zooms, qfac = _affine_quaternion(affine)
    hdr = bytearray(348)
    struct.pack_into('<i', hdr, 0, 348)
    dim = [array.ndim] + list(array.shape) + [1] * (7 - array.ndim)
    struct.pack_into('<8h', hdr, 40, *dim)
    struct.pack_into('<2h', hdr, 70, _NIFTI_CODES[key], array.dtype.itemsize * 8)
    pixdim = [qfac] + [float(zooms[i]) if i < 3 else 1.0 for i in range(array.ndim)] + [1.0] * (7 - array.ndim)
    struct.pack_into('<8f', hdr, 76, *pixdim[:8])
    struct.pack_into('<3f', hdr, 108, 352.0, 0.0, 0.0)                         # vox_offset, scl_slope (0: no scaling), scl_inter
    struct.pack_into('<2h', hdr, 252, 0, 2)                                    # qform_code unknown, sform_code aligned
    struct.pack_into('<6f', hdr, 256, qb, qc, qd, *[float(v) for v in affine[:3, 3]])
    struct.pack_into('<12f', hdr, 280, *[float(v) for v in affine[:3, :].reshape(-1)])
    hdr[344:348] = b'n+1\0'
    data = np.asfortranarray(array.astype(array.dtype.newbyteorder('<'), copy=False))
    with _open(filename, 'wb') as f:
        f.write(bytes(hdr))
        f.write(b'\0\0\0\0')                                                  # no header extensions
        f.write(data.tobytes(order='F'))


_MGZ_DTYPES = {0: '>u1', 1: '>i4', 3: '>f4', 4: '>i2'}


def read_mgz(filename):
    """FreeSurfer .mgh / .mgz (big-endian): -> (array [w, h, d(, frames)], vox2ras affine) as nibabel's MGHImage"""
    with _open(filename, 'rb') as f:
        raw = f.read()
    version, w, h, d, frames, typ, _dof, good = struct.unpack('>7ih', raw[:30])
    if version != 1 or typ not in _MGZ_DTYPES:
        raise ValueError("'%s': not an MGH version-1 volume of type uchar / int / float / short" % filename)
    delta = np.array(struct.unpack('>3f', raw[30:42]), dtype=np.float64)
    Mdc = np.array(struct.unpack('>9f', raw[42:78]), dtype=np.float64).reshape(3, 3).T       # columns x_ras, y_ras, z_ras
    c_ras = np.array(struct.unpack('>3f', raw[78:90]), dtype=np.float64)
    if not good:                                                              # FreeSurfer's defaults when the RAS fields are not set
        delta = np.ones(3)
        Mdc = np.array([[-1, 0, 0], [0, 0, -1], [0, 1, 0]], dtype=np.float64).T
        c_ras = np.zeros(3)
    dt = np.dtype(_MGZ_DTYPES[typ])
    n = w * h * d * frames
    arr = np.frombuffer(raw, dtype=dt, count=n, offset=284).reshape((w, h, d, frames) if frames > 1 else (w, h, d), order='F')
    arr = arr.astype(dt.newbyteorder('='))
    MdcD = Mdc * delta
    aff = np.eye(4)
    aff[:3, :3] = MdcD
    aff[:3, 3] = c_ras - MdcD @ (np.array([w, h, d], dtype=np.float64) / 2.0)
    return arr, aff
