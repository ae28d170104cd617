"""Persistence container (row f3), with the CPU oracle as the scheme: behaviours of the
reference's tests/test_containers.py that do not depend on xarray, and the on-disk layout
of triflow/plugins/container.py (netCDF chunks + data.nc + metadata.yml)."""
import glob
import os

import numpy as np
import pytest

from oracle import numpy_path as ora
from triflow_amd import Model, Simulation, retrieve_container


def make_sim(**kw):
    m = Model("k * dxxT", "T", "k", compiler=ora.numpy_compiler)
    x = np.linspace(0, 10, 50, endpoint=False)
    fields = m.fields_template(x=x, T=np.cos(x * 2 * np.pi / 10))
    return Simulation(m, fields, dict(periodic=True, k=1), dt=.5, tmax=2., scheme=ora.Theta,
                      time_stepping=False, **kw)


def test_in_memory_all_and_last():
    sim = make_sim()
    c = sim.attach_container()
    sim.run(progress=False)
    assert c.data["T"].shape == (5, 50)            # initial state + 4 steps
    assert np.allclose(c.data["t"], [0, .5, 1., 1.5, 2.])
    assert np.array_equal(c.data["T"][-1], np.asarray(sim.fields["T"]))
    sim = make_sim()
    c = sim.attach_container(save="last")
    sim.run(progress=False)
    assert c.data["T"].shape == (1, 50) and c.data["t"][0] == 2.


def test_on_disk_chunks_merge_and_retrieve(tmp_path):
    sim = make_sim(id="run")
    c = sim.attach_container(str(tmp_path), nbuffer=2)
    sim.run(progress=False)
    back = retrieve_container(str(tmp_path / "run"))
    assert np.array_equal(back.data["T"], c.data["T"])
    assert back.metadata["k"] == 1 and back.metadata["periodic"] is True
    last = retrieve_container(str(tmp_path / "run"), isel="last")
    assert last.data["T"].shape == (50,)
    with pytest.raises(FileExistsError):
        make_sim(id="run").attach_container(str(tmp_path))
    make_sim(id="run").attach_container(str(tmp_path), force=True)
    with pytest.raises(FileNotFoundError):
        retrieve_container(str(tmp_path / "nothing"))
    with pytest.raises(ValueError):
        make_sim().attach_container(save="some")


def test_on_disk_layout_is_the_references(tmp_path):
    """data_<uuid>.nc chunks while running, one data.nc + metadata.yml at the end; netCDF-3
    with dimensions (t, x), coordinate variables and the metadata as global attributes."""
    from scipy.io import netcdf_file
    sim = make_sim(id="run")
    c = sim.attach_container(str(tmp_path), nbuffer=2)
    it = iter(sim)
    next(it); next(it)                                  # initial state + 2 steps = 3 snapshots
    chunks = glob.glob(str(tmp_path / "run" / "data_*.nc"))
    assert len(chunks) == 1 and len(c._cached) == 1     # a written buffer leaves host memory
    assert c.data["T"].shape == (3, 50)                 # ... and is served from disk
    for _ in it:
        pass
    assert sorted(os.listdir(tmp_path / "run")) == ["data.nc", "metadata.yml"]
    with netcdf_file(str(tmp_path / "run" / "data.nc"), "r", mmap=False) as nc:
        assert nc.version_byte == 2
        assert dict(nc.dimensions) == {"t": 5, "x": 50}
        assert nc.variables["T"].dimensions == ("t", "x")
        assert np.allclose(nc.variables["t"][:], [0, .5, 1., 1.5, 2.])
        assert nc.k == 1 and nc.periodic == 1
    assert np.array_equal(c.data["T"][-1], np.asarray(sim.fields["T"]))


def test_append_mode_keeps_earlier_data(tmp_path):
    """mode="a" on an existing container adds snapshots: nothing is overwritten, merge()
    folds the earlier data.nc in (the reference opens data*.nc)."""
    from triflow_amd.container import TriflowContainer
    sim = make_sim(id="run")
    sim.attach_container(str(tmp_path))
    sim.run(progress=False)
    first = retrieve_container(str(tmp_path / "run")).data
    again = TriflowContainer(str(tmp_path / "run"), mode="a", metadata=dict(k=1, periodic=True),
                             nbuffer=2)
    m = Model("k * dxxT", "T", "k", compiler=ora.numpy_compiler)
    x = np.linspace(0, 10, 50, endpoint=False)
    for t in (2.5, 3.0, 3.5):
        again._collect(t, m.fields_template(x=x, T=np.full(50, t)))
    again.flush()
    assert len(glob.glob(str(tmp_path / "run" / "data_*.nc"))) == 2
    both = again.data
    assert np.allclose(both["t"], [0, .5, 1., 1.5, 2., 2.5, 3., 3.5])
    assert np.array_equal(both["T"][:5], first["T"])
    again.merge()
    back = retrieve_container(str(tmp_path / "run"))
    assert back.data["T"].shape == (8, 50) and np.array_equal(back.data["T"][:5], first["T"])
    assert np.array_equal(back.data["T"][7], np.full(50, 3.5))
    assert retrieve_container(str(tmp_path / "run"), isel=slice(1, 3)).data["t"].tolist() == [.5, 1.]
