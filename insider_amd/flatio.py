""".RData-free exchange formats for the INSIDER inputs and factors.

Two interchangeable carriers, both column-major like R's matrices (so R can produce / consume them with writeBin /
readBin, r/insider_hip.R:insider_write_flat):
  * ``.npy`` files (numpy; Fortran or C order, any real dtype — converted on load);
  * a "flat" directory: raw little-endian arrays X.f64 (n x p), levels.i32 (n x c, 1-based), train.u8, test.u8 (n x p),
    optional ctns.f64 (n x m), plus manifest.json {"n", "p", "c", "m", "format": "insider-flat-1"}; results are written
    back as A<i>.f64 (L_i x K), C.f64 (K x p), result.json.
"""
import json
import os

import numpy as np

FORMAT = "insider-flat-1"
_DT = {"f64": np.dtype("<f8"), "i32": np.dtype("<i4"), "u8": np.dtype("u1")}


def write_raw(path, a):
    """Column-major raw dump; the extension picks the element type (.f64 / .i32 / .u8)."""
    dt = _DT[path.rsplit(".", 1)[1]]
    np.asfortranarray(a, dtype=dt).ravel(order="F").tofile(path)


def read_raw(path, shape):
    dt = _DT[path.rsplit(".", 1)[1]]
    count = int(np.prod(shape))
    a = np.fromfile(path, dtype=dt)
    if a.size != count:
        raise ValueError(f"{path}: expected {count} elements for shape {tuple(shape)}, found {a.size}")
    return a.reshape(shape, order="F")


def write_flat(dirname, X, levels, train, test, ctns=None):
    os.makedirs(dirname, exist_ok=True)
    X = np.asarray(X)
    levels = np.asarray(levels).reshape(X.shape[0], -1)
    write_raw(os.path.join(dirname, "X.f64"), X)
    write_raw(os.path.join(dirname, "levels.i32"), levels)
    write_raw(os.path.join(dirname, "train.u8"), np.asarray(train) != 0)
    write_raw(os.path.join(dirname, "test.u8"), np.asarray(test) != 0)
    m = 0
    if ctns is not None:
        ctns = np.asarray(ctns, dtype=np.float64).reshape(X.shape[0], -1)
        m = ctns.shape[1]
        write_raw(os.path.join(dirname, "ctns.f64"), ctns)
    with open(os.path.join(dirname, "manifest.json"), "w") as f:
        json.dump({"n": int(X.shape[0]), "p": int(X.shape[1]), "c": int(levels.shape[1]), "m": m, "format": FORMAT}, f)
    return dirname


def read_flat(dirname):
    """-> dict(X, levels, train, test, ctns or None)"""
    with open(os.path.join(dirname, "manifest.json")) as f:
        mf = json.load(f)
    if mf.get("format") != FORMAT:
        raise ValueError(f"{dirname}: not an {FORMAT} directory")
    n, p, c, m = int(mf["n"]), int(mf["p"]), int(mf["c"]), int(mf.get("m", 0))
    out = dict(X=read_raw(os.path.join(dirname, "X.f64"), (n, p)), levels=read_raw(os.path.join(dirname, "levels.i32"), (n, c)),
               train=read_raw(os.path.join(dirname, "train.u8"), (n, p)), test=read_raw(os.path.join(dirname, "test.u8"), (n, p)),
               ctns=None)
    if m > 0:
        out["ctns"] = read_raw(os.path.join(dirname, "ctns.f64"), (n, m))
    return out


def load_matrix(path, dtype=None):
    """A matrix from .npy (or .csv / .txt with comma or whitespace separators)."""
    if path.endswith(".npy"):
        a = np.load(path, allow_pickle=False)
    else:
        with open(path) as f:
            delim = "," if "," in f.readline() else None
        a = np.loadtxt(path, delimiter=delim, ndmin=2)
    return np.asarray(a, dtype=dtype) if dtype is not None else a


def write_result(outdir, fmt, row_matrices, column_factor, summary):
    """Factors + result.json; fmt = "npy" or "flat"."""
    os.makedirs(outdir, exist_ok=True)
    for i, a in enumerate(row_matrices):
        if fmt == "npy":
            np.save(os.path.join(outdir, f"A{i}.npy"), np.asfortranarray(a))
        else:
            write_raw(os.path.join(outdir, f"A{i}.f64"), a)
    if fmt == "npy":
        np.save(os.path.join(outdir, "C.npy"), np.asfortranarray(column_factor))
    else:
        write_raw(os.path.join(outdir, "C.f64"), column_factor)
    with open(os.path.join(outdir, "result.json"), "w") as f:
        json.dump(summary, f, indent=1)
