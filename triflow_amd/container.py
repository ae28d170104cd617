"""Persistence of simulation states.

Counterpart of the reference's ``TriflowContainer``
(``triflow/plugins/container.py:44-253``) without xarray: it subscribes to the
simulation's stream, keeps ``(t, fields)`` snapshots (``save="all"``) or only the last
one (``save="last"``), and -- when a ``path`` is given -- writes them every ``nbuffer``
snapshots in the reference's on-disk layout::

    <path>/metadata.yml          the simulation parameters (container.py:68-70)
    <path>/data_<uuid1>.nc       one netCDF file per flushed buffer (container.py:129-131)
    <path>/data.nc               all of them merged at the end of the run (container.py:230-253)

The files are netCDF-3 (64-bit offset) written by ``scipy.io.netcdf_file``: dimensions
``t`` and ``x``, coordinate variables of the same names, one ``(t, x)`` variable per
field and the metadata as global attributes -- what ``xarray.Dataset.to_netcdf`` writes
with its SciPy engine, so the reference's ``retrieve_container`` / ``open_mfdataset`` read
them.  (HDF5-based netCDF-4 files, which a reference installation with the netCDF4
package writes, need that package to be read back here: INTEGRATION.md.)

Reading a snapshot is what brings a device-resident container to the host: a run that
saves every step pays one D2H per step, a run with ``save="last"`` pays one in total.
With a ``path`` the snapshots leave host memory when their buffer is written.

``container.data`` is a dict ``{"t": [nt], "x": [N], var: [nt, N], ...}`` sorted by ``t``.
"""

import glob
import os
import shutil
from uuid import uuid1

import numpy as np
import yaml
from scipy.io import netcdf_file


def _write_nc(filename, data, attrs):
    """``data``: {"t": [nt], "x": [N], field: [nt, N]} -> one netCDF-3 file."""
    tmp = filename + ".part"
    with netcdf_file(tmp, "w", version=2, mmap=False) as nc:
        nt, nx = len(data["t"]), len(data["x"])
        nc.createDimension("t", nt)
        nc.createDimension("x", nx)
        for key, value in attrs.items():
            value = _plain(value)
            setattr(nc, str(key), int(value) if isinstance(value, bool) else value)
        for key, dims in (("t", ("t",)), ("x", ("x",))):
            var = nc.createVariable(key, "d", dims)
            var[:] = np.asarray(data[key], dtype=float)
        for key, value in data.items():
            if key in ("t", "x"):
                continue
            var = nc.createVariable(key, "d", ("t", "x"))
            var[:] = np.asarray(value, dtype=float).reshape(nt, nx)
    os.replace(tmp, filename)


def _read_nc(filename):
    with netcdf_file(filename, "r", mmap=False) as nc:
        return {key: np.array(var[:], dtype=float) for key, var in nc.variables.items()}


def _concat_sorted(parts):
    """Snapshots of several files as one table, sorted by ``t`` (the reference's
    ``open_mfdataset(..., concat_dim="t").sortby("t")``, container.py:181-182)."""
    parts = [p for p in parts if p and len(p.get("t", ())) > 0]
    if not parts:
        return {}
    t = np.concatenate([np.atleast_1d(p["t"]) for p in parts])
    order = np.argsort(t, kind="stable")
    out = {"t": t[order], "x": parts[0]["x"]}
    for key in parts[0]:
        if key not in ("t", "x"):
            out[key] = np.concatenate([p[key] for p in parts])[order]
    return out


class TriflowContainer:
    def __init__(self, path=None, mode="a", *, save="all", metadata=None, force=False,
                 nbuffer=50):
        if save not in ("all", "last", -1):
            raise ValueError('save argument accept only "all", "last" or -1 as value, '
                             'not %s' % save)
        self.save = "all" if save == "all" else "last"
        self._nbuffer = nbuffer
        self._mode = mode
        self._metadata = dict(metadata or {})
        self._cached = []            # snapshots not yet on disk (all of them without a path)
        self.path = os.path.abspath(path) if path else None
        if not self.path:
            return
        if mode == "w" and force and os.path.exists(self.path):
            shutil.rmtree(self.path)
        if mode == "w" and not force and os.path.exists(self.path):
            raise FileExistsError("Directory %s exists, set force=True to override it"
                                  % self.path)
        if mode == "r" and not os.path.exists(self.path):
            raise FileNotFoundError("Container not found.")
        os.makedirs(self.path, exist_ok=True)
        if mode != "r":
            with open(os.path.join(self.path, "metadata.yml"), "w") as f:
                yaml.safe_dump({k: _plain(v) for k, v in self._metadata.items()}, f,
                               default_flow_style=False)

    # ---- collecting ---------------------------------------------------------------
    def connect(self, stream):
        stream.sink(lambda simul: self._collect(simul.t, simul.fields))
        return self

    def _collect(self, t, fields):
        snap = {"t": float(t)}
        for key in fields.keys():
            snap[key] = np.array(fields[key], dtype=float)      # host copy (D2H if resident)
        if self.save == "last":
            self._cached = [snap]
            return
        self._cached.append(snap)
        if self.path and len(self._cached) >= self._nbuffer:
            self.flush()

    @staticmethod
    def _stack(snaps):
        if not snaps:
            return {}
        out = {"t": np.array([s["t"] for s in snaps])}
        for key in snaps[0]:
            if key == "t":
                continue
            out[key] = snaps[0][key] if key == "x" else np.stack([s[key] for s in snaps])
        return out

    def flush(self):
        """Write the buffered snapshots as one ``data_<uuid1>.nc`` (unique names: a container
        re-opened with ``mode="a"`` adds files, it never overwrites) and drop them from
        memory; ``save="last"`` keeps only the newest file (container.py:127-137)."""
        if not self.path or not self._cached or self._mode == "r":
            return
        target = os.path.join(self.path, "data_%i.nc" % uuid1().int)
        _write_nc(target, self._stack(self._cached), self._metadata)
        self._cached = []
        if self.save == "last":
            # only the per-flush files, like the reference (container.py:132-135): the merged
            # data.nc of an earlier run of a container re-opened with mode="a" stays
            for other in glob.glob(os.path.join(self.path, "data_*.nc")):
                if other != target:
                    os.remove(other)

    def merge(self, override=True):
        """All ``data*.nc`` of the directory (an earlier ``data.nc`` included) into one
        ``data.nc`` sorted by ``t`` (container.py:230-253)."""
        if not self.path:
            return None
        return TriflowContainer.merge_datafiles(self.path, override=override)

    @staticmethod
    def merge_datafiles(path, override=False):
        path = os.path.abspath(path)
        merged_file = os.path.join(path, "data.nc")
        chunks = sorted(glob.glob(os.path.join(path, "data_*.nc")))
        if not chunks:
            return merged_file if os.path.exists(merged_file) else None
        if os.path.exists(merged_file) and not override:
            raise FileExistsError(merged_file)
        files = ([merged_file] if os.path.exists(merged_file) else []) + chunks
        merged = _concat_sorted([_read_nc(f) for f in files])
        attrs = {}
        meta_file = os.path.join(path, "metadata.yml")
        if os.path.exists(meta_file):
            with open(meta_file) as f:
                attrs = yaml.safe_load(f) or {}
        _write_nc(merged_file, merged, attrs)
        check = _read_nc(merged_file)
        if any(not np.array_equal(check[k], merged[k]) for k in merged):
            os.remove(merged_file)
            raise IOError("Unable to merge data ")
        for c in chunks:
            os.remove(c)
        return merged_file

    # ---- reading ------------------------------------------------------------------
    @property
    def data(self):
        loaded = self.__dict__.get("_loaded")
        if loaded is not None:
            return loaded
        on_disk = []
        if self.path:
            on_disk = [_read_nc(f) for f in sorted(glob.glob(os.path.join(self.path, "data*.nc")))]
        return _concat_sorted(on_disk + [self._stack(self._cached)])

    @property
    def metadata(self):
        return dict(self._metadata)

    @staticmethod
    def retrieve(path, isel="all", lazy=True):
        """``FieldsData``-like object (``.data``, ``.metadata``) of a container on disk;
        ``isel="last"`` or an index / slice along ``t`` selects snapshots
        (container.py:174-208)."""
        path = os.path.abspath(path)
        if not os.path.exists(path):
            raise FileNotFoundError("Container not found.")
        merged_file = os.path.join(path, "data.nc")
        files = [merged_file] if os.path.exists(merged_file) \
            else sorted(glob.glob(os.path.join(path, "data*.nc")))
        data = _concat_sorted([_read_nc(f) for f in files])
        if data and isel != "all":
            sel = -1 if isel == "last" else (isel.get("t", slice(None)) if isinstance(isel, dict) else isel)
            for key in data:
                if key != "x":
                    data[key] = data[key][sel]
        meta = {}
        if os.path.exists(os.path.join(path, "metadata.yml")):
            with open(os.path.join(path, "metadata.yml")) as f:
                meta = yaml.safe_load(f) or {}
        out = TriflowContainer.__new__(TriflowContainer)
        out.__dict__.update(path=path, save="all", _cached=[], _metadata=meta,
                            _nbuffer=0, _mode="r", _loaded=data)
        return out

    def __repr__(self):
        d = self.data
        nt = len(d["t"]) if "t" in d else 0
        return "path:   %s\n%i snapshots of %s" % (self.path, nt,
                                                  [k for k in d if k not in ("t", "x")])


def _plain(value):
    """YAML / netCDF-attribute friendly scalar (the reference coerces attributes the same
    way, container.py:27-41)."""
    if isinstance(value, (bool, int, float, str)):
        return value
    if np.ndim(value) == 0:
        for cast in (int, float):
            try:
                if cast(value) == value:
                    return cast(value)
            except (TypeError, ValueError):
                pass
    return str(value)


def retrieve_container(path, isel="all", lazy=False):
    """Reference ``triflow.retrieve_container`` (``triflow/__init__.py:9``)."""
    return TriflowContainer.retrieve(path, isel=isel)
