"""Persistence of simulation states.

A small, xarray-free counterpart of the reference's ``TriflowContainer``
(``triflow/plugins/container.py:44-253``): it subscribes to the simulation's stream,
keeps ``(t, fields)`` snapshots (``save="all"``) or only the last one
(``save="last"``), and -- when a ``path`` is given -- writes them every ``nbuffer``
snapshots as ``data_<k>.npz`` chunks next to a ``metadata.yml``, merged into one
``data.npz`` at the end of the run (the reference writes netCDF through xarray,
which this stack does not have).  Reading a snapshot is what brings a
device-resident container to the host: a run that saves every step pays one D2H
per step, a run with ``save="last"`` pays one in total.

``container.data`` is a dict ``{"t": [nt], "x": [N], var: [nt, N], ...}``.
"""

import glob
import os
import shutil

import numpy as np
import yaml


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
        self._pending = []           # snapshots not yet written
        self._kept = []              # snapshots kept in memory
        self._nchunks = 0
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
            self._kept = [snap]
            self._pending = [snap]
            return
        self._kept.append(snap)
        self._pending.append(snap)
        if self.path and len(self._pending) >= self._nbuffer:
            self.flush()

    @staticmethod
    def _stack(snaps):
        if not snaps:
            return {}
        out = {"t": np.array([s["t"] for s in snaps])}
        for key in snaps[0]:
            if key == "t":
                continue
            if key == "x":
                out[key] = snaps[0][key]
            else:
                out[key] = np.stack([s[key] for s in snaps])
        return out

    def flush(self):
        if not self.path or not self._pending:
            return
        name = "data.npz" if self.save == "last" else "data_%06d.npz" % self._nchunks
        np.savez(os.path.join(self.path, name), **self._stack(self._pending))
        self._nchunks += 1
        self._pending = []

    def merge(self, override=True):
        """Concatenate the chunks into ``data.npz`` (reference container.py:230-253)."""
        if not self.path:
            return
        chunks = sorted(glob.glob(os.path.join(self.path, "data_*.npz")))
        if not chunks:
            return
        parts = [dict(np.load(c)) for c in chunks]
        merged = {"t": np.concatenate([p["t"] for p in parts]), "x": parts[0]["x"]}
        for key in parts[0]:
            if key not in ("t", "x"):
                merged[key] = np.concatenate([p[key] for p in parts])
        np.savez(os.path.join(self.path, "data.npz"), **merged)
        if override:
            for c in chunks:
                os.remove(c)

    # ---- reading ------------------------------------------------------------------
    @property
    def data(self):
        if self._kept:
            return self._stack(self._kept)
        if self.path:
            return self.retrieve(self.path).data
        return {}

    @property
    def metadata(self):
        return dict(self._metadata)

    @staticmethod
    def retrieve(path, isel="all"):
        """``FieldsData``-like object (``.data``, ``.metadata``) of a container on disk;
        ``isel="last"`` or an index / slice along ``t`` selects snapshots."""
        path = os.path.abspath(path)
        if not os.path.exists(path):
            raise FileNotFoundError("Container not found.")
        files = [os.path.join(path, "data.npz")] if os.path.exists(os.path.join(path, "data.npz")) \
            else sorted(glob.glob(os.path.join(path, "data_*.npz")))
        parts = [dict(np.load(f)) for f in files]
        data = {}
        if parts:
            data = {"t": np.concatenate([p["t"] for p in parts]), "x": parts[0]["x"]}
            for key in parts[0]:
                if key not in ("t", "x"):
                    data[key] = np.concatenate([p[key] for p in parts])
            if isel != "all":
                sel = -1 if isel == "last" else isel
                for key in data:
                    if key != "x":
                        data[key] = data[key][sel]
        meta = {}
        if os.path.exists(os.path.join(path, "metadata.yml")):
            with open(os.path.join(path, "metadata.yml")) as f:
                meta = yaml.safe_load(f) or {}
        out = TriflowContainer.__new__(TriflowContainer)
        out.__dict__.update(path=path, save="all", _kept=[], _pending=[], _metadata=meta,
                            _nchunks=0, _nbuffer=0, _mode="r", _loaded=data)
        return out

    def __getattribute__(self, name):
        if name == "data":
            loaded = object.__getattribute__(self, "__dict__").get("_loaded")
            if loaded is not None:
                return loaded
        return object.__getattribute__(self, name)

    def __repr__(self):
        d = self.data
        nt = len(d["t"]) if "t" in d else 0
        return "path:   %s\n%i snapshots of %s" % (self.path, nt,
                                                  [k for k in d if k not in ("t", "x")])


def _plain(value):
    """YAML-friendly scalar (the reference coerces attributes the same way,
    container.py:27-41)."""
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
